/*
 * ohevc_annexb.h — the byte-stream side of the path (C ABI, host only, part of libohevc_host.so): what sits
 * between a raw .bit / .bin file and the host decoder that fills the work lists, and what the picture-hash check needs from the stream.
 *
 *   access-unit splitter     libavcodec/hevc_parser.c:40-87   hevc_find_frame_end (the parser the reference's harness reads raw files through)
 *   NAL unit scan of an AU   libavcodec/hevc.c:3854-3893      decode_nal_units' start-code search + hls_nal_unit's header (hevc.c:3672-3697)
 *   NAL unescape             libavcodec/hevc.c:3724-3829      ff_hevc_extract_rbsp (emulation-prevention bytes out, their positions kept)
 *   picture-hash SEI         libavcodec/hevc_sei.c:28-50,134-181   decode_nal_sei_message / decode_nal_sei_decoded_picture_hash
 *
 * Entropy decoding (CABAC, hevc_cabac.c) is NOT here: it stays on the host decoder (out of scope).
 */
#ifndef OHEVC_ANNEXB_H
#define OHEVC_ANNEXB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OH_AU_END_NOT_FOUND (-100)          /* parser.h END_NOT_FOUND */

/* streaming state of the splitter (hevc_parser.c's ParseContext.state64 / frame_start_found) */
typedef struct OhAuScanner { uint64_t state64; int32_t frame_start_found; int32_t reserved; } OhAuScanner;
void oh_au_scanner_init(OhAuScanner *sc);
/* Feed the next `size` bytes.  Returns the offset in `buf` of the first byte of the NEXT access unit — the first zero of the
 * three-byte start code that opens it; negative down to -5 when that start code began in the bytes of an earlier call — or
 * OH_AU_END_NOT_FOUND.  A new access unit begins at the first VPS / SPS / PPS / AUD / prefix-SEI / reserved 41-44, 48-55 NAL unit of
 * layer 0 that follows a picture's first slice segment, or at the next slice segment with first_slice_segment_in_pic_flag = 1. */
long oh_au_find_frame_end(OhAuScanner *sc, const uint8_t *buf, size_t size);
/* Whole buffer: offsets[0] = 0, offsets[k] = start of access unit k, offsets[n] = size; returns n (needs cap >= n + 1), or -(n + 1)
 * when cap is too small (nothing written past cap). */
long oh_annexb_split(const uint8_t *data, size_t size, size_t *offsets, size_t cap);

typedef struct OhNal {
    size_t  offset;                         /* of the first header byte (behind the start code) in the buffer scanned */
    size_t  size;                           /* escaped bytes up to the next start code / trailing zeros (header included) */
    int32_t type, layer_id, temporal_id;    /* nal_unit_type, nuh_layer_id, nuh_temporal_id_plus1 - 1 */
    int32_t first_slice_segment_in_pic;     /* VCL NAL units: the flag; others 0 */
} OhNal;
/* NAL units of one access unit (or any Annex-B buffer).  Returns their number (at most cap written), or -1 when bytes other than
 * zeros precede a start code ("No start code is found", hevc.c:3876). */
long oh_annexb_nal_units(const uint8_t *buf, size_t size, OhNal *out, size_t cap);

/* RBSP of one NAL unit: src[0 .. length) starts at the NAL header; dst needs `length` bytes.  Emulation-prevention bytes (00 00 03)
 * are dropped and the position in dst of the byte before each one — as ff_hevc_extract_rbsp keeps them for the entry-point
 * arithmetic of hls_slice_data_wpp (hevc.c:2829-2842) — goes to skipped_pos (up to cap; *n_skipped counts all).  Stops at the next
 * start code (00 00 00 / 00 00 01 / 00 00 02).  Returns the bytes of src consumed. */
long oh_nal_unescape(const uint8_t *src, size_t length, uint8_t *dst, size_t *dst_size, int32_t *skipped_pos, size_t cap, int32_t *n_skipped);

typedef struct OhPictureHash {
    int32_t  present;                       /* a decoded-picture-hash message was found */
    int32_t  hash_type;                     /* 0 MD5, 1 CRC, 2 checksum */
    uint8_t  md5[3][16];
    uint32_t crc[3];
    uint32_t checksum[3];
} OhPictureHash;
/* nal[0 .. size): one (escaped) SEI NAL unit, header first.  Walks its messages; a decoded picture hash (payload type 132 in a suffix
 * SEI, 256 in a prefix SEI as the reference also accepts) fills *out.  Returns 1 found, 0 none, -1 malformed / not an SEI NAL unit. */
int oh_sei_picture_hash(const uint8_t *nal, size_t size, OhPictureHash *out);

#ifdef __cplusplus
}
#endif
#endif
