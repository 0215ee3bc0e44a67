/*
 * ohevc_stream.h — synthetic HEVC Annex-B stream writer (C ABI, host only; part of libohevc_host.so).
 *
 * No HEVC encoder or conformance stream exists in the build environment (SURVEY.md §7 "hard parts", §8c).  This writer
 * produces LEGAL bitstreams from scratch: parameter sets (VPS / SPS / PPS), slice headers, and CABAC-coded slice data
 * whose syntax elements are drawn at random inside what H.265 allows — coding quadtrees, skip / merge / AMVP prediction
 * units with random motion-vector differences, intra modes through the most-probable-mode syntax, transform trees,
 * residual blocks, PCM and transquant-bypass coding units, SAO parameters, several slices per picture, tiles, wavefront
 * entry points.  There is no rate control and no search: what the pictures look like is whatever the DECODER derives
 * from the syntax, which is the point — the reference decoder (oracle/_ref, built from /root/reference) decodes these
 * streams and is the checker for everything downstream of entropy decoding (tests/test_streams.py).
 *
 * Scope: Main / Main 10 (4:2:0) and the 4:4:4 range-extension profile incl. cross-component prediction, 8 or 10 bit; one layer, or two (SHVC
 * spatial scalability: shvc_el_width / shvc_el_height).  Written from the H.265 syntax (7.3) and CABAC (9.3) clauses.
 */
#ifndef OHEVC_STREAM_H
#define OHEVC_STREAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OhStreamParams {
    uint64_t seed;
    int32_t width, height;              /* luma samples, multiples of the minimum CB size (8) */
    int32_t bit_depth;                  /* 8 or 10 */
    int32_t log2_ctb_size;              /* 4..6 */
    int32_t log2_min_tb_size, log2_max_tb_size;   /* 2..5 */
    int32_t max_th_depth_intra, max_th_depth_inter;   /* max_transform_hierarchy_depth_* */
    int32_t n_pictures;
    int32_t gop;                        /* 0 all intra, 1 low-delay P, 2 low-delay B (both lists from past pictures), 3 hierarchical B: mini-GOPs of four
                                           in decode order +4, +2, +1, +3 (output order != decode order, two pictures of reordering; +1 / +3 are
                                           sub-layer non-reference pictures); n_refs >= 2 */
    int32_t n_refs;                     /* reference pictures kept (1..4) */
    int32_t idr_period;                 /* > 0: an IDR picture every so many pictures */
    int32_t qp;                         /* slice QP */
    /* tools (0 / 1) */
    int32_t amp, sao, pcm, transquant_bypass, transform_skip, cu_qp_delta, tmvp, strong_intra_smoothing, constrained_intra_pred,
            scaling_list /* 1 default lists, 2 random lists in the SPS */, weighted_pred, sign_data_hiding, cabac_init_present, deblocking_override;
    /* picture structure */
    int32_t n_slices;                   /* slices per picture (>= 1), at random CTB addresses (whole tiles when tiles are on) */
    int32_t tile_cols, tile_rows;       /* > 1: uniformly spaced tiles */
    int32_t wpp;                        /* entropy_coding_sync_enabled_flag */
    int32_t dependent_slices;           /* dependent_slice_segments_enabled_flag: about half of the slices after the first become dependent segments */
    int32_t lf_across_slices, lf_across_tiles;    /* the two loop_filter_across_* flags */
    /* content knobs, per cent */
    int32_t split_pct, intra_pct, skip_pct, merge_pct, bi_pct, cbf_pct, pcm_pct, bypass_pct, tskip_pct, sao_pct;
    int32_t mvd_range;                  /* |mvd| bound in quarter samples; occasionally far larger */
    int32_t coeff_density;              /* 1..100: how many coefficients a coded block gets */
    int32_t trace;                      /* 1: keep the list of syntax elements written (oh_stream_trace) */
    int32_t levels;                     /* 1: keep the quantised levels of every residual block written (oh_stream_levels); needs cu_qp_delta = 0 */
    int32_t conf_win_left, conf_win_right, conf_win_top, conf_win_bottom;   /* conformance window in luma samples (even), 0 = none */
    int32_t chroma_format_idc;          /* 1 (4:2:0, Main / Main 10), 2 (4:2:2: two square chroma blocks per transform unit one above the other, two
                                           chroma cbf flags, mode mapping of table 8-3) or 3 (4:4:4: chroma blocks of luma size incl. 4x4, one
                                           intra_chroma_pred_mode per partition); 2 and 3: format range extensions profile, chroma QP = min(qPi, 51) */
    int32_t cross_component_pred;       /* 4:4:4 only: cross_component_prediction_enabled_flag, random log2_res_scale_abs_plus1 / sign per chroma block */
    /* range-extension coding tools (sps_range_extension 7.3.2.2.2; any of them selects the format range extensions profile, also for 4:2:0): */
    int32_t tskip_rotation;             /* transform_skip_rotation_enabled_flag: 4x4 intra transform-skip blocks rotated by 180 degrees (hevc_cabac.c:1877-1884) */
    int32_t tskip_context;              /* transform_skip_context_enabled_flag: one sig_coeff_flag context for skip / bypass blocks (hevc_cabac.c:1633-1680) */
    int32_t implicit_rdpcm;             /* implicit_rdpcm_enabled_flag: intra bypass blocks predicted along mode 10 / 26 (hevc_cabac.c:1868-1874) */
    int32_t explicit_rdpcm;             /* explicit_rdpcm_enabled_flag: explicit_rdpcm_flag / _dir_flag on inter skip / bypass blocks (hevc_cabac.c:1502-1508) */
    int32_t intra_smoothing_disabled;   /* intra_smoothing_disabled_flag (hevcpred_template.c:289) */
    int32_t persistent_rice;            /* persistent_rice_adaptation_enabled_flag (hevc_cabac.c:1719-1725, 1779-1807); not with wpp: the reference does not
                                           synchronise StatCoeff with the contexts, its threaded and serial decodes would differ */
    int32_t log2_max_tskip_size;        /* 0 or 2..5: log2_max_transform_skip_block_size (pps_range_extension); > 2 needs one of the tools above or 4:4:4 */
    int32_t pcm_loop_filter;            /* 1: pcm_loop_filter_disabled_flag = 0 (PCM blocks are deblocked like any other) */
    int32_t chroma_qp_offsets;          /* 1: pps_cb_qp_offset / pps_cr_qp_offset from the next two fields (-12..12) instead of +1 / -2 */
    int32_t cb_qp_offset, cr_qp_offset;
    int32_t sao_offset_scale_luma, sao_offset_scale_chroma;   /* log2_sao_offset_scale_* of the pps_range_extension: 0 .. bit_depth - 10 (so: 12 bit only) */
    int32_t log2_min_cb_size;           /* 0 (= 3) or 3..5: smallest coding block (width and height are multiples of it); above 8x8 its inter
                                           partitions include NxN, its min PU / QP / PCM map granularity follows */
    int32_t shvc_el_width, shvc_el_height;   /* > 0: a TWO-LAYER stream (SHVC spatial scalability, the SHM 4.1 syntax the reference parses): this stream is
                                           the base layer (8 bit 4:2:0, no window, not gop 3, no range extensions), every access unit also carries an enhancement-layer
                                           picture of this size (the base layer's to twice the base layer's, and more than one CTB + 16 samples each way: x1 = SNR, x1.5 — up to 2048 columns and rows —, x2 or any ratio between) whose P slices predict from the
                                           up-sampled base-layer picture only (zero motion vectors), plus intra blocks and residuals */
} OhStreamParams;

/* syntax elements of the slice data as (id, value) pairs in coding order — the writer's side of tests/test_streams.py; the ids are
 * shared with oracle/ref_trace_unit.c, which logs what the reference decoder parsed */
enum { OH_SE_SAO_MERGE = 1, OH_SE_SAO_TYPE, OH_SE_SAO_OFFSET_ABS, OH_SE_SAO_OFFSET_SIGN, OH_SE_SAO_BAND_POS, OH_SE_SAO_EO_CLASS, OH_SE_END_OF_SLICE,
       OH_SE_SPLIT_CU, OH_SE_BYPASS_FLAG, OH_SE_SKIP, OH_SE_PRED_MODE, OH_SE_PART_MODE, OH_SE_PCM_FLAG, OH_SE_PREV_INTRA, OH_SE_MPM_IDX,
       OH_SE_REM_INTRA, OH_SE_CHROMA_MODE, OH_SE_MERGE_FLAG, OH_SE_MERGE_IDX, OH_SE_INTER_DIR, OH_SE_REF_IDX, OH_SE_MVD_X, OH_SE_MVD_Y, OH_SE_MVP,
       OH_SE_ROOT_CBF, OH_SE_SPLIT_TU, OH_SE_CBF_LUMA, OH_SE_CBF_CHROMA, OH_SE_QP_DELTA_ABS, OH_SE_QP_DELTA_SIGN, OH_SE_RESIDUAL,
       OH_SE_RES_SCALE_ABS, OH_SE_RES_SCALE_SIGN };
size_t oh_stream_trace(const int32_t **recs);            /* of the last oh_stream_write with trace = 1; pairs (id, value) */

/* the residual blocks of the last oh_stream_write with levels = 1, in coding order, pictures one after the other, as uint32 words:
 *   log2_size | c_idx << 4 | transform_skip << 8 | cu_transquant_bypass << 9 | intra CU << 10 | cross-component prediction << 11 |
 *   no residual_coding() (a cross-component block with cbf = 0) << 12 | qp << 16                             qp: the block's QP with QpBdOffset
 *   n | (uint8_t)res_scale_val << 24                                                                        number of non-zero levels
 *   n x (pos | (uint16_t)level << 16)                                                                       pos = y * N + x
 * — what residual_coding hands to de-quantisation (hevc_cabac.c:1478-1494, 1818-1841): the sparse hand-over of include/ohevc_frame.h is
 * built from this in tests/test_sparse_pin.py and must reproduce the reference decoder's pictures. */
size_t oh_stream_levels(const uint32_t **words);

typedef struct OhStream {
    uint8_t *data;                      /* Annex-B byte stream (start codes included) */
    size_t   size;
    int32_t  n_pictures;
    size_t  *au_offset;                 /* n_pictures + 1 offsets: access unit i is data[au_offset[i] .. au_offset[i+1]) */
} OhStream;

void oh_stream_defaults(OhStreamParams *p, int width, int height, uint64_t seed);
/* returns 0, or a negative value when the parameters are outside what the writer covers */
int  oh_stream_write(const OhStreamParams *p, OhStream *out);
/* the same stream with a decoded-picture-hash SEI (MD5, payload type 132, suffix SEI NAL) behind every picture;
 * md5[i * 48 ..]: the three plane digests of picture i in decode order */
int  oh_stream_add_md5(const OhStream *in, const uint8_t *md5, OhStream *out);
void oh_stream_free(OhStream *s);

#ifdef __cplusplus
}
#endif
#endif
