/*
 * ohevc_tables.h — the reference's own plugin point: arch init hooks that fill its
 * function-pointer tables.
 *
 *   void ff_hevcdsp_init_hip (HEVCDSPContext  *c, const int bit_depth);   // like ff_hevcdsp_init_x86,  hevcdsp.h:173, called hevcdsp.c:1326
 *   void ff_hevcpred_init_hip(HEVCPredContext *c, const int bit_depth);   // like ff_hevcpred_init_x86, hevcpred.h:44,  called hevcpred.c:84
 *   void ff_videodsp_init_hip(VideoDSPContext *c, int bit_depth);         // like ff_videodsp_init_x86, videodsp.c:57
 *
 * The table types below reproduce the reference layouts member for member (hevcdsp.h:44-124 with
 * COM16_C806_EMT 0, hevcpred.h:31-41, videodsp.h emulated_edge_mc) so that the hooks can be
 * linked into the reference unchanged; reference-owned structs are opaque here.
 *
 * The slots installed by the hooks do not touch pixels.  They RECORD: each call is translated
 * into work-list items of the OhRecorder bound to the calling thread (oh_tables_bind), by
 * resolving the raw pointers the reference passes (Appendix B of SURVEY.md):
 *
 *   idct*, idct_dc*, idct_4x4_luma, transform_skip,   remember the transform of `coeffs`
 *   transform_rdpcm                                   (hevc_cabac.c:1868-1934)
 *   transform_add[n](dst, coeffs, stride)             -> oh_rec_tu(); attached to the intra block recorded
 *                                                        just before for the same position, else added in pass 2
 *   put_hevc_qpel*[..] (luma)                         -> oh_rec_pu(); MV rebuilt from the source position and
 *                                                        mx/my; put + bi(_w) pairs fused (hevc.c:1761-1773)
 *   put_hevc_epel*[..] (chroma)                       carry only the chroma weights of the PU being assembled
 *   vdsp.emulated_edge_mc                             no copy: remembers which picture rectangle the buffer
 *                                                        stands for (the MC kernel clamps coordinates itself)
 *   hpc.intra_pred[n](s, x0, y0, c_idx)               -> oh_rec_intra(); mode and candidate flags come from the
 *                                                        accessor registered with oh_tables_set_intra_accessor()
 *   hevc_*_loop_filter_*, sao_*                       no-ops: passes 4-5 run from the BS/QP/SAO arrays
 *   put_pcm                                           reads the samples (get_bits) and records OH_TU_PCM blocks
 *   upsample_*                                        do nothing: the inter-layer reference picture is resampled in HBM by
 *                                                     oh_pic_upsample, issued by the decoder's hand-over
 */
#ifndef OHEVC_TABLES_H
#define OHEVC_TABLES_H

#include <stddef.h>
#include <stdint.h>
#include "ohevc_recorder.h"

#ifdef __cplusplus
extern "C" {
#endif

/* layout of libavcodec/get_bits.h:54-59; the put_pcm slot reads its samples through it */
struct GetBitContext { const uint8_t *buffer, *buffer_end; int index; int size_in_bits; int size_in_bits_plus8; };
struct SAOParams; struct AVFrame; struct HEVCWindow; struct UpsamplInf; struct HEVCContext;

/* hevcdsp.h:44-124 */
typedef struct HEVCDSPContext {
    void (*put_pcm)(uint8_t *_dst, ptrdiff_t _stride, int width, int height, struct GetBitContext *gb, int pcm_bit_depth);
    void (*transform_add[4])(uint8_t *_dst, int16_t *coeffs, ptrdiff_t _stride);
    void (*transform_skip)(int16_t *coeffs, int16_t log2_size);
    void (*transform_rdpcm)(int16_t *coeffs, int16_t log2_size, int mode);
    void (*idct_4x4_luma)(int16_t *coeffs);
    void (*idct[4])(int16_t *coeffs, int col_limit);
    void (*idct_dc[4])(int16_t *coeffs);
    void (*sao_band_filter)(uint8_t *_dst, uint8_t *_src, ptrdiff_t _stride_dst, ptrdiff_t _stride_src, struct SAOParams *sao,
                            int *borders, int width, int height, int c_idx);
    void (*sao_edge_filter[2])(uint8_t *_dst, uint8_t *_src, ptrdiff_t _stride_dst, ptrdiff_t _stride_src, struct SAOParams *sao,
                               int *borders, int _width, int _height, int c_idx, uint8_t *vert_edge, uint8_t *horiz_edge,
                               uint8_t *diag_edge);
    void (*put_hevc_qpel[10][2][2])(int16_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                    int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_qpel_uni[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                        int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_qpel_uni_w[10][2][2])(uint8_t *_dst, ptrdiff_t _dststride, uint8_t *_src, ptrdiff_t _srcstride,
                                          int height, int denom, int wx, int ox, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_qpel_bi[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *_src, ptrdiff_t _srcstride,
                                       int16_t *src2, ptrdiff_t src2stride, int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_qpel_bi_w[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *_src, ptrdiff_t _srcstride,
                                         int16_t *src2, ptrdiff_t src2stride, int height, int denom, int wx0, int wx1,
                                         int ox0, int ox1, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel[10][2][2])(int16_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                                    int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel_uni[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *_src, ptrdiff_t _srcstride,
                                        int height, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel_uni_w[10][2][2])(uint8_t *_dst, ptrdiff_t _dststride, uint8_t *_src, ptrdiff_t _srcstride,
                                          int height, int denom, int wx, int ox, intptr_t mx, intptr_t my, int width);
    void (*put_hevc_epel_bi[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *_src, ptrdiff_t _srcstride,
                                       int16_t *src2, ptrdiff_t src2stride, int height, intptr_t mx, intptr_t my, int width);
    /* argument POSITIONS are (denom, w_l0, w_l1, o_l0, o_l1) at the call sites hevc.c:1940-1948 */
    void (*put_hevc_epel_bi_w[10][2][2])(uint8_t *dst, ptrdiff_t dststride, uint8_t *_src, ptrdiff_t _srcstride,
                                         int16_t *src2, ptrdiff_t src2stride, int height, int denom, int wx0, int ox0, int wx1,
                                         int ox1, intptr_t mx, intptr_t my, int width);
    void (*hevc_h_loop_filter_luma)(uint8_t *_pix, ptrdiff_t _stride, int _beta, int *_tc, uint8_t *_no_p, uint8_t *_no_q);
    void (*hevc_v_loop_filter_luma)(uint8_t *_pix, ptrdiff_t _stride, int _beta, int *_tc, uint8_t *_no_p, uint8_t *_no_q);
    void (*hevc_h_loop_filter_chroma)(uint8_t *_pix, ptrdiff_t _stride, int *_tc, uint8_t *_no_p, uint8_t *_no_q);
    void (*hevc_v_loop_filter_chroma)(uint8_t *_pix, ptrdiff_t _stride, int *_tc, uint8_t *_no_p, uint8_t *_no_q);
    void (*hevc_h_loop_filter_luma_c)(uint8_t *_pix, ptrdiff_t _stride, int _beta, int *_tc, uint8_t *_no_p, uint8_t *_no_q);
    void (*hevc_v_loop_filter_luma_c)(uint8_t *_pix, ptrdiff_t _stride, int _beta, int *_tc, uint8_t *_no_p, uint8_t *_no_q);
    void (*hevc_h_loop_filter_chroma_c)(uint8_t *_pix, ptrdiff_t _stride, int *_tc, uint8_t *_no_p, uint8_t *_no_q);
    void (*hevc_v_loop_filter_chroma_c)(uint8_t *_pix, ptrdiff_t _stride, int *_tc, uint8_t *_no_p, uint8_t *_no_q);
    void (*upsample_base_layer_frame)(struct AVFrame *FrameEL, struct AVFrame *FrameBL, short *Buffer[3],
                                      const struct HEVCWindow *Enhscal, struct UpsamplInf *up_info, int channel);
    void (*upsample_filter_block_luma_h[3])(int16_t *dst, ptrdiff_t dststride, uint8_t *_src, ptrdiff_t _srcstride, int x_EL, int x_BL,
                                            int block_w, int block_h, int widthEL, const struct HEVCWindow *Enhscal, struct UpsamplInf *up_info);
    void (*upsample_filter_block_luma_v[3])(uint8_t *dst, ptrdiff_t dststride, int16_t *_src, ptrdiff_t _srcstride, int y_BL, int x_EL, int y_EL,
                                            int block_w, int block_h, int widthEL, int heightEL, const struct HEVCWindow *Enhscal, struct UpsamplInf *up_info);
    void (*upsample_filter_block_cr_h[3])(int16_t *dst, ptrdiff_t dststride, uint8_t *_src, ptrdiff_t _srcstride, int x_EL, int x_BL,
                                          int block_w, int block_h, int widthEL, const struct HEVCWindow *Enhscal, struct UpsamplInf *up_info);
    void (*upsample_filter_block_cr_v[3])(uint8_t *dst, ptrdiff_t dststride, int16_t *_src, ptrdiff_t _srcstride, int y_BL, int x_EL, int y_EL,
                                          int block_w, int block_h, int widthEL, int heightEL, const struct HEVCWindow *Enhscal, struct UpsamplInf *up_info);
} HEVCDSPContext;

/* hevcpred.h:31-41 */
typedef struct HEVCPredContext {
    void (*intra_pred[4])(struct HEVCContext *s, int x0, int y0, int c_idx);
    void (*pred_planar[4])(uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride);
    void (*pred_dc)(uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int log2_size, int c_idx);
    void (*pred_angular[4])(uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int c_idx, int mode);
} HEVCPredContext;

/* videodsp.h: the member the HEVC decoder uses */
typedef struct VideoDSPContext {
    void (*emulated_edge_mc)(uint8_t *buf, const uint8_t *src, ptrdiff_t buf_linesize, ptrdiff_t src_linesize,
                             int block_w, int block_h, int src_x, int src_y, int w, int h);
} VideoDSPContext;

void ff_hevcdsp_init_hip(HEVCDSPContext *c, const int bit_depth);
void ff_hevcpred_init_hip(HEVCPredContext *c, const int bit_depth);
void ff_videodsp_init_hip(VideoDSPContext *c, int bit_depth);

/* ---- binding the recording slots to the picture being decoded by THIS thread ---- */

/* what the intra_pred slot needs from the reference's HEVCContext (hevcpred_template.c:73-109):
 * the mode of the block and the RESOLVED candidate flags (lc->na.* combined with the z-scan order
 * tests, OH_AV_* bits).  Implemented on the reference side, where hevc.h is visible. */
typedef void (*oh_intra_accessor)(struct HEVCContext *s, int x0, int y0, int c_idx, int log2_size, int *mode, int *avail);

/* cur_data/cur_linesize: the AVFrame planes the reference decodes into (s->frame, never written
 * by the recording slots); ref pictures likewise, slot == index into the recorder's ref_pics. */
void oh_tables_bind(OhRecorder *rec, uint8_t *const cur_data[3], const int cur_linesize[3]);
void oh_tables_bind_ref(int slot, uint8_t *const data[3], const int linesize[3]);
void oh_tables_set_intra_accessor(oh_intra_accessor fn);
/* Cross-component prediction (4:4:4 range extension, hevc.c:1295-1365, hevc_cabac.c:1942-1947): the reference adds
 * (res_scale_val * luma residual) >> 3 to a chroma block's residual on the host, between the slot calls — with recording slots the
 * luma residual does not exist yet.  The host decoder instead skips that addition and calls this right after it parsed the
 * component's res_scale_val (hls_cross_component_pred): the NEXT chroma block that reaches transform_add is linked to the luma
 * block of the same transform unit (oh_rec_tu_cross) and the engine does the addition.  INTEGRATION.md §10. */
void oh_tables_cross(int res_scale_val);
/* flush the PU being assembled; returns the number of slot calls that could not be translated
 * (unknown pointers, unsupported slots) since oh_tables_bind() */
int  oh_tables_finish(void);
/* diagnostics: the untranslated calls since oh_tables_bind() by slot family — 0 transform_add, 1 put_pcm, 2 intra without accessor,
 * 3 intra refused by the recorder, 4 edge emulation, 5 luma interpolation, 6 list-0 half of a bi-predicted block, 7 PU refused */
void oh_tables_untranslated_by_family(int out[8]);

#ifdef __cplusplus
}
#endif
#endif
