/*
 * oracle.h — CPU restatement of openHEVC's block-reconstruction hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so, and only as
 * the checker.  The product (openhevc_amd/) never links or falls back to this code.
 *
 * Parity status: PINNED.  Every function here is checked bit-for-bit against the reference's
 * own C kernels compiled from /root/reference (oracle/_ref, recipe oracle/Makefile) by
 * tests/test_oracle_vs_ref.py, and against the committed fixtures in tests/golden/ (generated
 * by tests/golden/make_golden.py from oracle/_ref).
 *
 * Each function cites the reference file:line whose arithmetic it restates (paths relative to
 * /root/reference/libavcodec/).  The restatement is written from the arithmetic (direct matrix
 * products instead of partial butterflies, one generic separable-filter core instead of 40
 * macro-expanded variants, whole-picture filter passes instead of per-CTB drivers).
 */
#ifndef OHEVC_ORACLE_H
#define OHEVC_ORACLE_H

#include <stddef.h>
#include <stdint.h>
#include "../include/ohevc_frame.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------- slot level (one call == one HEVCDSPContext / HEVCPredContext slot) -------- */

/* hevcdsp_template.c:45-111  transform_add{4,8,16,32}; stride in bytes */
void oh_or_transform_add(int bd, uint8_t *dst, const int16_t *res, ptrdiff_t stride, int log2);
/* hevcdsp_template.c:139-163 */
void oh_or_transform_skip(int bd, int16_t *coeffs, int log2);
/* hevcdsp_template.c:114-136 */
void oh_or_transform_rdpcm(int16_t *coeffs, int log2, int mode);
/* hevcdsp_template.c:185-203 */
void oh_or_idct_4x4_luma(int bd, int16_t *coeffs);
/* hevcdsp_template.c:279-301 (col_limit only skips zero inputs, so it is not a parameter) */
void oh_or_idct(int bd, int16_t *coeffs, int log2);
/* hevcdsp_template.c:303-316 */
void oh_or_idct_dc(int bd, int16_t *coeffs, int log2);

/* Interpolation (hevcdsp_template.c:610-1609).  taps = 8 (qpel: fx,fy in 0..3) or 4 (epel:
 * fx,fy in 0..7).  src points at the block origin inside a plane that has the needed margin
 * (taps/2-1 before, taps/2 after); strides of pixel buffers are in BYTES, of int16 in elements. */
void oh_or_mc_put  (int bd, int taps, int16_t *dst, ptrdiff_t dststride,
                    const uint8_t *src, ptrdiff_t srcstride, int h, int fx, int fy, int w);
void oh_or_mc_uni  (int bd, int taps, uint8_t *dst, ptrdiff_t dststride,
                    const uint8_t *src, ptrdiff_t srcstride, int h, int fx, int fy, int w);
void oh_or_mc_bi   (int bd, int taps, uint8_t *dst, ptrdiff_t dststride,
                    const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2, ptrdiff_t src2stride,
                    int h, int fx, int fy, int w);
void oh_or_mc_uni_w(int bd, int taps, uint8_t *dst, ptrdiff_t dststride,
                    const uint8_t *src, ptrdiff_t srcstride, int h, int denom, int wx, int ox,
                    int fx, int fy, int w);
void oh_or_mc_bi_w (int bd, int taps, uint8_t *dst, ptrdiff_t dststride,
                    const uint8_t *src, ptrdiff_t srcstride, const int16_t *src2, ptrdiff_t src2stride,
                    int h, int denom, int wx0, int wx1, int ox0, int ox1, int fx, int fy, int w);

/* hevcpred_template.c:359-538; top/left point at element 0, [-1] must be valid; like the
 * reference's pred_* slots (called with linesize/sizeof(pixel), :88) stride is in PIXELS */
void oh_or_pred_planar (int bd, uint8_t *dst, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int log2);
void oh_or_pred_dc     (int bd, uint8_t *dst, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int log2, int c_idx);
void oh_or_pred_angular(int bd, uint8_t *dst, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int log2, int c_idx, int mode);

/* hevcpred_template.c:30-344 with constrained_intra_pred_flag == 0: neighbour gathering from the
 * plane using the RESOLVED candidate flags, substitution, smoothing, prediction.
 * plane/stride(bytes)/pw/ph describe plane c_idx; x,y are in samples of that plane. */
void oh_or_intra_pred(const OhPicParams *p, uint8_t *plane, ptrdiff_t stride, int pw, int ph,
                      int x, int y, int c_idx, int log2, int mode, int avail, const uint8_t *is_intra);   /* is_intra: OhFrame.is_intra, NULL unless constrained_intra_pred */

/* hevcdsp_template.c:1629-1757; xstride/ystride in bytes as in the reference's inner functions */
void oh_or_loop_filter_luma  (int bd, uint8_t *pix, ptrdiff_t xstride, ptrdiff_t ystride, int beta,
                              const int *tc, const uint8_t *no_p, const uint8_t *no_q);
void oh_or_loop_filter_chroma(int bd, uint8_t *pix, ptrdiff_t xstride, ptrdiff_t ystride,
                              const int *tc, const uint8_t *no_p, const uint8_t *no_q);

/* hevcdsp_template.c:340-365 / :372-567 (variant 1 when any edge flag is given) */
void oh_or_sao_band(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t dststride, ptrdiff_t srcstride,
                    const int16_t *offset_val, int band_position, int w, int h);
void oh_or_sao_edge(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t dststride, ptrdiff_t srcstride,
                    const int16_t *offset_val, int eo_class, const int *borders, int w, int h,
                    int restore, const uint8_t *vert_edge, const uint8_t *horiz_edge, const uint8_t *diag_edge);

/* ---------------- picture level (one call == one GPU pass over one picture) ---------------- */

typedef struct OhHostPic {
    uint8_t  *data[3];
    ptrdiff_t stride[3];          /* bytes */
    int32_t   width[3], height[3];
    int32_t   bit_depth;
} OhHostPic;

/* pics[] is indexed by picture id (OhFrame.cur_pic / ref_pics[]) */
int oh_or_pass_inter   (const OhFrame *f, OhHostPic *pics);   /* hevc.c:1641-1949, 2103-2153 */
int oh_or_pass_residual(const OhFrame *f, OhHostPic *pics, int16_t *coeffs_rw); /* hevc_cabac.c:1868-1949 */
int oh_or_pass_intra   (const OhFrame *f, OhHostPic *pics, const int16_t *residuals);
int oh_or_pass_deblock (const OhFrame *f, OhHostPic *pics);   /* hevc_filter.c:345-581 */
int oh_or_pass_sao     (const OhFrame *f, OhHostPic *pics);   /* hevc_filter.c:197-322 */
/* all five in order; coefficients are copied internally (f->coeffs stays const) */
int oh_or_frame        (const OhFrame *f, OhHostPic *pics);

/* ---- SHVC up-sampling slots, hevcdsp_template.c:1834-2438.  variant: OH_UP_DEFAULT / X2 / X1_5;
 * strides in ELEMENTS of the pointed type, exactly as the slots use them. */
void oh_or_up_luma_h(int variant, int bd, int16_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                     int x_el, int x_bl, int block_w, int block_h, int width_el, const OhUpsample *u);
void oh_or_up_cr_h  (int variant, int bd, int16_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                     int x_el, int x_bl, int block_w, int block_h, int width_el, const OhUpsample *u);
void oh_or_up_luma_v(int variant, int bd, uint8_t *dst, ptrdiff_t dststride, const int16_t *src, ptrdiff_t srcstride,
                     int y_bl, int x_el, int y_el, int block_w, int block_h, int width_el, int height_el, const OhUpsample *u);
void oh_or_up_cr_v  (int variant, int bd, uint8_t *dst, ptrdiff_t dststride, const int16_t *src, ptrdiff_t srcstride,
                     int y_bl, int x_el, int y_el, int block_w, int block_h, int width_el, int height_el, const OhUpsample *u);
/* whole picture, 8-bit 4:2:0 only like the reference's routine (its edge code and shift are written for bytes):
 * upsample_base_layer_frame, hevcdsp_template.c:2164-2438.  el/bl: coded sizes = width[0]/height[0]. */
int  oh_or_upsample_frame(const OhHostPic *bl, OhHostPic *el, const OhUpsample *u);
/* boundary strengths from the maps ff_hevc_deblocking_boundary_strengths reads (hevc_filter.c:584-941); vbs / hbs: oh_bs_size(p) bytes */
int  oh_or_bs_derive(const OhPicParams *p, const OhBsInputs *in, uint8_t *vbs, uint8_t *hbs);

#ifdef __cplusplus
}
#endif
#endif
