/*
 * ref_trace_unit.c — TEST INFRASTRUCTURE ONLY.  The reference's hevc.c compiled as it lies in /root/reference (the #include
 * at the bottom; nothing of it is copied) with every CABAC syntax-element function it calls wrapped by a logging macro: the
 * resulting library (oracle/_ref/libopenhevc_trace.so = the whole reference decoder with this unit in place of hevc.o)
 * reports the sequence of syntax elements the reference PARSED from a stream.  tests/test_streams.py compares it with the
 * sequence the stream writer (openhevc_amd/synth/stream.c) says it WROTE: the proof that the writer's streams mean to the
 * reference what the writer intended, element by element.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "libavcodec/hevc.h"

typedef struct TraceRec { int32_t id, v; } TraceRec;
static TraceRec *g_tr;
static size_t g_n, g_cap;
static int g_on;
static FILE *g_file;
static int tr_(int id, int v)
{
    if (g_file) { int32_t rec[2] = { id, v }; fwrite(rec, 4, 2, g_file); fflush(g_file); }     /* OH_TRACE_FILE: survives a crash of the decoder */
    if (g_on) {
        if (g_n == g_cap) { g_cap = g_cap ? 2 * g_cap : 1 << 16; g_tr = realloc(g_tr, g_cap * sizeof(*g_tr)); }
        g_tr[g_n].id = id; g_tr[g_n].v = v; g_n++;
    }
    return v;
}
__attribute__((visibility("default"))) void ref_trace_start(void)
{
    g_n = 0; g_on = 1;
    if (!g_file && getenv("OH_TRACE_FILE")) g_file = fopen(getenv("OH_TRACE_FILE"), "wb");
}
__attribute__((visibility("default"))) size_t ref_trace_get(const void **recs) { *recs = g_tr; return g_n; }

/* ids shared with the writer's trace (include/ohevc_stream.h: OH_SE_*) */
enum { SE_SAO_MERGE = 1, SE_SAO_TYPE, SE_SAO_OFFSET_ABS, SE_SAO_OFFSET_SIGN, SE_SAO_BAND_POS, SE_SAO_EO_CLASS, SE_END_OF_SLICE, SE_SPLIT_CU,
       SE_BYPASS_FLAG, SE_SKIP, SE_PRED_MODE, SE_PART_MODE, SE_PCM_FLAG, SE_PREV_INTRA, SE_MPM_IDX, SE_REM_INTRA, SE_CHROMA_MODE, SE_MERGE_FLAG,
       SE_MERGE_IDX, SE_INTER_DIR, SE_REF_IDX, SE_MVD_X, SE_MVD_Y, SE_MVP, SE_ROOT_CBF, SE_SPLIT_TU, SE_CBF_LUMA, SE_CBF_CHROMA,
       SE_QP_DELTA_ABS, SE_QP_DELTA_SIGN, SE_RESIDUAL, SE_RES_SCALE_ABS, SE_RES_SCALE_SIGN };

#define ff_hevc_sao_merge_flag_decode(s)            tr_(SE_SAO_MERGE, ff_hevc_sao_merge_flag_decode(s))
#define ff_hevc_sao_type_idx_decode(s)              tr_(SE_SAO_TYPE, ff_hevc_sao_type_idx_decode(s))
#define ff_hevc_sao_offset_abs_decode(s)            tr_(SE_SAO_OFFSET_ABS, ff_hevc_sao_offset_abs_decode(s))
#define ff_hevc_sao_offset_sign_decode(s)           tr_(SE_SAO_OFFSET_SIGN, ff_hevc_sao_offset_sign_decode(s))
#define ff_hevc_sao_band_position_decode(s)         tr_(SE_SAO_BAND_POS, ff_hevc_sao_band_position_decode(s))
#define ff_hevc_sao_eo_class_decode(s)              tr_(SE_SAO_EO_CLASS, ff_hevc_sao_eo_class_decode(s))
#define ff_hevc_end_of_slice_flag_decode(s)         tr_(SE_END_OF_SLICE, ff_hevc_end_of_slice_flag_decode(s))
#define ff_hevc_split_coding_unit_flag_decode(...)  tr_(SE_SPLIT_CU, ff_hevc_split_coding_unit_flag_decode(__VA_ARGS__))
#define ff_hevc_cu_transquant_bypass_flag_decode(s) tr_(SE_BYPASS_FLAG, ff_hevc_cu_transquant_bypass_flag_decode(s))
#define ff_hevc_skip_flag_decode(...)               tr_(SE_SKIP, ff_hevc_skip_flag_decode(__VA_ARGS__))
#define ff_hevc_pred_mode_decode(s)                 tr_(SE_PRED_MODE, ff_hevc_pred_mode_decode(s))
#define ff_hevc_part_mode_decode(...)               tr_(SE_PART_MODE, ff_hevc_part_mode_decode(__VA_ARGS__))
#define ff_hevc_pcm_flag_decode(s)                  tr_(SE_PCM_FLAG, ff_hevc_pcm_flag_decode(s))
#define ff_hevc_prev_intra_luma_pred_flag_decode(s) tr_(SE_PREV_INTRA, ff_hevc_prev_intra_luma_pred_flag_decode(s))
#define ff_hevc_mpm_idx_decode(s)                   tr_(SE_MPM_IDX, ff_hevc_mpm_idx_decode(s))
#define ff_hevc_rem_intra_luma_pred_mode_decode(s)  tr_(SE_REM_INTRA, ff_hevc_rem_intra_luma_pred_mode_decode(s))
#define ff_hevc_intra_chroma_pred_mode_decode(s)    tr_(SE_CHROMA_MODE, ff_hevc_intra_chroma_pred_mode_decode(s))
#define ff_hevc_merge_flag_decode(s)                tr_(SE_MERGE_FLAG, ff_hevc_merge_flag_decode(s))
#define ff_hevc_merge_idx_decode(s)                 tr_(SE_MERGE_IDX, ff_hevc_merge_idx_decode(s))
#define ff_hevc_inter_pred_idc_decode(...)          tr_(SE_INTER_DIR, ff_hevc_inter_pred_idc_decode(__VA_ARGS__))
#define ff_hevc_ref_idx_lx_decode(...)              tr_(SE_REF_IDX, ff_hevc_ref_idx_lx_decode(__VA_ARGS__))
#define ff_hevc_mvp_lx_flag_decode(s)               tr_(SE_MVP, ff_hevc_mvp_lx_flag_decode(s))
#define ff_hevc_no_residual_syntax_flag_decode(s)   tr_(SE_ROOT_CBF, ff_hevc_no_residual_syntax_flag_decode(s))
#define ff_hevc_split_transform_flag_decode(...)    tr_(SE_SPLIT_TU, ff_hevc_split_transform_flag_decode(__VA_ARGS__))
#define ff_hevc_cbf_luma_decode(...)                tr_(SE_CBF_LUMA, ff_hevc_cbf_luma_decode(__VA_ARGS__))
#define ff_hevc_cbf_cb_cr_decode(...)               tr_(SE_CBF_CHROMA, ff_hevc_cbf_cb_cr_decode(__VA_ARGS__))
#define ff_hevc_cu_qp_delta_abs(s)                  tr_(SE_QP_DELTA_ABS, ff_hevc_cu_qp_delta_abs(s))
#define ff_hevc_cu_qp_delta_sign_flag(s)            tr_(SE_QP_DELTA_SIGN, ff_hevc_cu_qp_delta_sign_flag(s))
#define ff_hevc_log2_res_scale_abs(...)             tr_(SE_RES_SCALE_ABS, ff_hevc_log2_res_scale_abs(__VA_ARGS__))
#define ff_hevc_res_scale_sign_flag(...)            tr_(SE_RES_SCALE_SIGN, ff_hevc_res_scale_sign_flag(__VA_ARGS__))
#define ff_hevc_hls_mvd_coding(s, x0, y0, l)        do { ff_hevc_hls_mvd_coding(s, x0, y0, l); tr_(SE_MVD_X, (s)->HEVClc->pu.mvd.x); tr_(SE_MVD_Y, (s)->HEVClc->pu.mvd.y); } while (0)
#define ff_hevc_hls_residual_coding(s, x0, y0, l, sc, c) do { tr_(SE_RESIDUAL, (l) | ((c) << 4) | ((sc) << 8)); ff_hevc_hls_residual_coding(s, x0, y0, l, sc, c); } while (0)

#include "libavcodec/hevc.c"
