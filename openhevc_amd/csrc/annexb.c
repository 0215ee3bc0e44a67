/*
 * annexb.c — Annex-B byte stream: access-unit splitter, NAL scan, NAL unescape, picture-hash SEI.  See include/ohevc_annexb.h for the
 * reference lines each entry point follows.  Plain C, no dependencies.
 */
#include "../../include/ohevc_annexb.h"
#include <stdlib.h>
#include <string.h>

enum { NAL_RASL_R = 9, NAL_BLA_W_LP = 16, NAL_CRA_NUT = 21, NAL_VPS = 32, NAL_AUD = 35, NAL_SEI_PREFIX = 39, NAL_SEI_SUFFIX = 40 };

void oh_au_scanner_init(OhAuScanner *sc)
{
    sc->state64 = ~(uint64_t)0;                               /* no start code in the history */
    sc->frame_start_found = 0;
    sc->reserved = 0;
}

static int is_vcl(int nut) { return nut <= NAL_RASL_R || (nut >= NAL_BLA_W_LP && nut <= NAL_CRA_NUT); }

/* what a NAL unit type means for the access-unit boundary (7.4.2.4.4: the first of VPS .. AUD, prefix SEI, the reserved / unspecified ranges
 * 41..44 and 48..55 after the last VCL NAL unit of a picture starts a new access unit; a VCL NAL unit does when it is a first slice segment) */
enum { CLS_NEUTRAL = 0, CLS_OPENS_AU = 1, CLS_VCL = 2 };
static int nal_class(int nut)
{
    static uint8_t cls[64];
    static int built;
    if (!built) {                                             /* (idempotent: concurrent first calls write the same values) */
        for (int t = 0; t < 64; t++)
            cls[t] = is_vcl(t) ? CLS_VCL : ((t >= NAL_VPS && t <= NAL_AUD) || t == NAL_SEI_PREFIX || (t >= 41 && t <= 44) || (t >= 48 && t <= 55)) ? CLS_OPENS_AU : CLS_NEUTRAL;
        built = 1;
    }
    return cls[nut & 63];
}

/* one NAL unit header seen by the scanner: returns 1 when the access unit that was open ENDS in front of this unit.  Only units of the base
 * layer move the state (hevc_parser.c:62, 71): the pictures of higher layers belong to the access unit of their base-layer picture. */
static int au_step(OhAuScanner *sc, int cls, int layer_id, int first_slice_segment)
{
    if (layer_id != 0 || cls == CLS_NEUTRAL || (cls == CLS_VCL && !first_slice_segment))
        return 0;
    if (!sc->frame_start_found) {                             /* no picture open: a first slice segment opens one, anything else waits */
        sc->frame_start_found = cls == CLS_VCL;
        return 0;
    }
    sc->frame_start_found = 0;                                /* a picture was open: this unit belongs to the next access unit */
    return 1;
}

long oh_au_find_frame_end(OhAuScanner *sc, const uint8_t *buf, size_t size)
{
    /* the history holds the last eight bytes across calls (the parser's state64): a header is complete when the byte BEHIND its two bytes
     * arrives — that byte carries first_slice_segment_in_pic_flag — i.e. when bytes -5 .. -3 of the history are the start code 00 00 01 */
    uint64_t hist = sc->state64;
    for (size_t i = 0; i < size; i++) {
        hist = (hist << 8) | buf[i];
        if ((hist & 0xFFFFFF000000ull) != 0x000001000000ull)
            continue;
        const unsigned h0 = (unsigned)(hist >> 16) & 0xFF, h1 = (unsigned)(hist >> 8) & 0xFF;      /* the two header bytes */
        if (au_step(sc, nal_class((int)(h0 >> 1) & 0x3F), (int)((h0 & 1) << 5 | h1 >> 3), buf[i] >> 7)) {
            sc->state64 = hist;
            return (long)i - 5;                               /* the cut lies in front of the start code (three bytes) + header (two) */
        }
    }
    sc->state64 = hist;
    return OH_AU_END_NOT_FOUND;
}

long oh_annexb_split(const uint8_t *data, size_t size, size_t *offsets, size_t cap)
{
    OhAuScanner sc;
    oh_au_scanner_init(&sc);
    size_t n = 0, pos = 0;
    if (cap > 0) offsets[0] = 0;
    while (pos < size) {
        const long r = oh_au_find_frame_end(&sc, data + pos, size - pos);
        if (r == OH_AU_END_NOT_FOUND)
            break;
        /* the bytes from the cut on are fed again with a fresh history, as the parser's caller re-feeds what ff_combine_frame
         * (parser.c) held back: the start code that closed the access unit is seen once more with frame_start_found = 0, where a
         * parameter set does nothing and a first slice segment opens the next picture */
        pos += (size_t)r;                                     /* r >= 0: a fresh history cannot end inside earlier bytes */
        n++;
        if (n < cap) offsets[n] = pos;
        oh_au_scanner_init(&sc);
    }
    n++;
    if (n < cap) offsets[n] = size;
    return n + 1 <= cap ? (long)n : -(long)(n + 1);
}

long oh_annexb_nal_units(const uint8_t *buf, size_t size, OhNal *out, size_t cap)
{
    size_t pos = 0, n = 0;
    while (size - pos >= 4) {
        if (buf[pos + 2] == 0) {                              /* zero_byte / leading zeros: slide */
            if (buf[pos] != 0) return -1;
            pos++;
            continue;
        }
        if (buf[pos] != 0 || buf[pos + 1] != 0 || buf[pos + 2] != 1)
            return -1;
        pos += 3;
        /* the unit runs to the next 00 00 0x with x < 3 (x == 3 is an escape), or to the end */
        size_t end = size;
        for (size_t i = pos; i + 2 < size; i++)
            if (buf[i] == 0 && buf[i + 1] == 0 && buf[i + 2] < 3) { end = i; break; }
        while (end == size && end > pos + 2 && buf[end - 1] == 0)
            end--, size--;                                    /* trailing_zero_8bits at the end of the buffer (a NAL unit never ends in 00) */
        if (end - pos >= 2 && n < cap) {
            OhNal *u = &out[n];
            u->offset = pos; u->size = end - pos;
            u->type = (buf[pos] >> 1) & 0x3F;
            u->layer_id = ((buf[pos] & 1) << 5) | (buf[pos + 1] >> 3);
            u->temporal_id = (buf[pos + 1] & 7) - 1;
            u->first_slice_segment_in_pic = is_vcl(u->type) && end - pos > 2 ? buf[pos + 2] >> 7 : 0;
        }
        if (end - pos >= 2) n++;
        pos = end;
    }
    return (long)n;
}

long oh_nal_unescape(const uint8_t *src, size_t length, uint8_t *dst, size_t *dst_size, int32_t *skipped_pos, size_t cap, int32_t *n_skipped)
{
    size_t si = 0, di = 0;
    int32_t ns = 0;
    while (si < length) {
        if (si + 2 < length && src[si] == 0 && src[si + 1] == 0 && src[si + 2] <= 3) {
            if (src[si + 2] != 3)
                break;                                        /* next start code: past the end of this unit */
            dst[di++] = 0; dst[di++] = 0;
            si += 3;
            if ((size_t)ns < cap && skipped_pos) skipped_pos[ns] = (int32_t)di - 1;
            ns++;
            continue;
        }
        dst[di++] = src[si++];
    }
    if (dst_size) *dst_size = di;
    if (n_skipped) *n_skipped = ns;
    return (long)si;
}

int oh_sei_picture_hash(const uint8_t *nal, size_t size, OhPictureHash *out)
{
    memset(out, 0, sizeof(*out));
    if (size < 3)
        return -1;
    const int nut = (nal[0] >> 1) & 0x3F;
    if (nut != NAL_SEI_PREFIX && nut != NAL_SEI_SUFFIX)
        return -1;
    uint8_t *rbsp = (uint8_t *)malloc(size);
    if (!rbsp)
        return -1;
    size_t n = 0;
    oh_nal_unescape(nal, size, rbsp, &n, NULL, 0, NULL);
    size_t p = 2;                                             /* behind the NAL header */
    int found = 0, bad = 0;
    /* sei_message()s until only rbsp_trailing_bits are left (hevc_sei.c:183-200 more_rbsp_data): the LAST byte being 0x80.  A 0x80
     * anywhere else is a payloadType (128, structure_of_pictures_info), not the end: a hash message may follow it in the same NAL */
    while (n > 0 && rbsp[n - 1] == 0) n--;                    /* cabac_zero_words / trailing zero bytes behind the trailing bits */
    while (p < n && !(p == n - 1 && rbsp[p] == 0x80)) {
        unsigned type = 0, len = 0;
        while (p < n && rbsp[p] == 0xFF) { type += 255; p++; }
        if (p >= n) { bad = 1; break; }
        type += rbsp[p++];
        while (p < n && rbsp[p] == 0xFF) { len += 255; p++; }
        if (p >= n) { bad = 1; break; }
        len += rbsp[p++];
        if (p + len > n) { bad = 1; break; }
        if ((nut == NAL_SEI_SUFFIX && type == 132) || (nut == NAL_SEI_PREFIX && type == 256)) {
            const uint8_t *q = rbsp + p;
            if (len < 1) { bad = 1; break; }
            const int ht = q[0];
            const unsigned per = ht == 0 ? 16 : ht == 1 ? 2 : ht == 2 ? 4 : 0;
            if (!per || len < 1 + per) { bad = 1; break; }
            const unsigned planes = (len - 1) / per < 3 ? (len - 1) / per : 3;      /* monochrome streams carry one */
            out->present = 1;
            out->hash_type = ht;
            for (unsigned c = 0; c < planes; c++) {
                const uint8_t *v = q + 1 + c * per;
                if (ht == 0) memcpy(out->md5[c], v, 16);
                else if (ht == 1) out->crc[c] = ((uint32_t)v[0] << 8) | v[1];
                else out->checksum[c] = ((uint32_t)v[0] << 24) | ((uint32_t)v[1] << 16) | ((uint32_t)v[2] << 8) | v[3];
            }
            found = 1;
        }
        p += len;
    }
    free(rbsp);
    return bad ? -1 : found;
}
