/*
 * oracle.c — CPU restatement of the block-reconstruction hot path (see oracle.h).
 * TEST INFRASTRUCTURE ONLY: the checker for the HIP engine, never a fallback for it.
 * Parity: pinned against the reference's own C kernels (oracle/_ref) and tests/golden/.
 */
#include <stdlib.h>
#include <string.h>
#include "oracle.h"

/* ------------------------------------------------------------------------------------------
 * constant tables (facts of ITU-T H.265; the reference holds the same numbers at the cited lines)
 * ---------------------------------------------------------------------------------------- */

/* luma 1/4-sample interpolation taps, H.265 table 8-11 (hevcdsp.c:1038-1042); [0] unused */
static const int8_t oh_qpel_taps[4][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 },
    { -1, 4, -10, 58, 17, -5, 1, 0 },
    { -1, 4, -11, 40, 40, -11, 4, -1 },
    { 0, 1, -5, 17, 58, -10, 4, -1 },
};
/* chroma 1/8-sample taps, H.265 table 8-12 (hevcdsp.c:1028-1036); [0] unused */
static const int8_t oh_epel_taps[8][8] = {
    { 0, 64, 0, 0 },  { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
    { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 },
};
/* intraPredAngle for modes 2..34 and invAngle for modes 11..25 (hevcpred_template.c:430-437) */
static const int8_t oh_intra_angle[33] = {
    32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
    -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32
};
static const int16_t oh_inv_angle[15] = {
    -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096
};
/* deblocking tables, H.265 table 8-12 (hevc_filter.c:50-60) */
static const uint8_t oh_tc_table[54] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1,
    1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
    5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24
};
static const uint8_t oh_beta_table[52] = {
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8,
    9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24, 26, 28, 30, 32, 34, 36,
    38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64
};

static inline int oh_clip3(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int oh_clip16(int v) { return oh_clip3(v, -32768, 32767); }

/* The 32x32 inverse-DCT matrix (hevcdsp.c:879-944) is the integer cosine table c[m] ~
 * 64*sqrt(2)*cos(m*pi/64) unfolded by symmetry: M[k][n] = +-c[fold(k*(2n+1) mod 128)]. */
static int8_t oh_dct[32][32];
static int oh_dct_ready;
static void oh_dct_init(void)
{
    static const int8_t c[32] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
                                  64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4 };
    if (oh_dct_ready)
        return;
    for (int k = 0; k < 32; k++)
        for (int n = 0; n < 32; n++) {
            int m = (k * (2 * n + 1)) & 127;
            if (m > 64) m = 128 - m;
            oh_dct[k][n] = (int8_t)(k == 0 ? 64 : (m == 32 ? 0 : (m < 32 ? c[m] : -c[64 - m])));
        }
    oh_dct_ready = 1;
}

/* ------------------------------------------------------------------------------------------
 * pixel-typed code
 * ---------------------------------------------------------------------------------------- */
#define PX uint8_t
#define FN(x) x##_8
#include "oracle_px.inc"
#undef PX
#undef FN
#define PX uint16_t
#define FN(x) x##_16
#include "oracle_px.inc"
#undef PX
#undef FN

#define DISPATCH(bd, call8, call16) do { if ((bd) == 8) { call8; } else { call16; } } while (0)

/* ------------------------------------------------------------------------------------------
 * residual slots
 * ---------------------------------------------------------------------------------------- */
void oh_or_transform_add(int bd, uint8_t *dst, const int16_t *res, ptrdiff_t stride, int log2)
{
    DISPATCH(bd, transform_add_8(bd, dst, res, stride, 1 << log2),
                 transform_add_16(bd, (uint16_t *)dst, res, stride / 2, 1 << log2));
}

void oh_or_transform_skip(int bd, int16_t *c, int log2)
{
    int shift = 15 - bd - log2, n2 = 1 << (2 * log2);
    for (int i = 0; i < n2; i++)
        c[i] = shift > 0 ? (int16_t)((c[i] + (1 << (shift - 1))) >> shift) : (int16_t)(c[i] << -shift);
}

void oh_or_transform_rdpcm(int16_t *c, int log2, int mode)
{
    int n = 1 << log2;
    for (int y = 0; y < n; y++)
        for (int x = 0; x < n; x++) {
            if (mode && y)       c[y * n + x] = (int16_t)(c[y * n + x] + c[(y - 1) * n + x]);
            else if (!mode && x) c[y * n + x] = (int16_t)(c[y * n + x] + c[y * n + x - 1]);
        }
}

/* one 1-D pass of an n-point inverse transform with basis row(k)[i], applied to all n lines;
 * pass 0 works down the columns, pass 1 along the rows (hevcdsp_template.c:283-300) */
static void inv_pass(int16_t *c, int n, int pass, int shift, const int8_t *basis, int bstride)
{
    int16_t out[32];
    int add = 1 << (shift - 1);
    for (int line = 0; line < n; line++) {
        int16_t *p = pass == 0 ? c + line : c + line * n;
        int step = pass == 0 ? n : 1;
        for (int i = 0; i < n; i++) {
            int acc = 0;
            for (int k = 0; k < n; k++)
                acc += basis[k * bstride + i] * p[k * step];
            out[i] = (int16_t)oh_clip16((acc + add) >> shift);
        }
        for (int i = 0; i < n; i++)
            p[i * step] = out[i];
    }
}

void oh_or_idct(int bd, int16_t *c, int log2)
{
    int n = 1 << log2;
    oh_dct_init();
    /* the n-point basis is every (32/n)-th row of the 32-point matrix */
    inv_pass(c, n, 0, 7, &oh_dct[0][0], 32 * (32 / n));
    inv_pass(c, n, 1, 20 - bd, &oh_dct[0][0], 32 * (32 / n));
}

void oh_or_idct_4x4_luma(int bd, int16_t *c)
{
    /* inverse DST-VII basis, H.265 eq. 8-315 (the reference factors it, hevcdsp_template.c:170-183) */
    static const int8_t dst7[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };
    inv_pass(c, 4, 0, 7, &dst7[0][0], 4);
    inv_pass(c, 4, 1, 20 - bd, &dst7[0][0], 4);
}

void oh_or_idct_dc(int bd, int16_t *c, int log2)
{
    int shift = 14 - bd, add = 1 << (shift - 1);
    int v = (((c[0] + 1) >> 1) + add) >> shift;
    for (int i = 0; i < (1 << (2 * log2)); i++)
        c[i] = (int16_t)v;
}

/* ------------------------------------------------------------------------------------------
 * interpolation slots
 * ---------------------------------------------------------------------------------------- */
#define MC_INTER(bd, taps, v, src, srcstride, w, h, fx, fy)                                        \
    DISPATCH(bd, mc_intermediate_8(bd, taps, v, src, srcstride, w, h, fx, fy),                      \
                 mc_intermediate_16(bd, taps, v, (const uint16_t *)(src), (srcstride) / 2, w, h, fx, fy))

void oh_or_mc_put(int bd, int taps, int16_t *dst, ptrdiff_t dststride, const uint8_t *src,
                  ptrdiff_t srcstride, int h, int fx, int fy, int w)
{
    int32_t v[64 * 64];
    MC_INTER(bd, taps, v, src, srcstride, w, h, fx, fy);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++)
            dst[y * dststride + x] = (int16_t)v[y * w + x];
}

void oh_or_mc_uni(int bd, int taps, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src,
                  ptrdiff_t srcstride, int h, int fx, int fy, int w)
{
    int32_t v[64 * 64];
    if (!fx && !fy) {                                     /* plain copy, hevcdsp_template.c:626-640 */
        int bpp = bd > 8 ? 2 : 1;
        for (int y = 0; y < h; y++)
            memcpy(dst + y * dststride, src + y * srcstride, (size_t)w * bpp);
        return;
    }
    MC_INTER(bd, taps, v, src, srcstride, w, h, fx, fy);
    DISPATCH(bd, mc_store_uni_8(bd, dst, dststride, v, w, h),
                 mc_store_uni_16(bd, (uint16_t *)dst, dststride / 2, v, w, h));
}

void oh_or_mc_bi(int bd, int taps, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src,
                 ptrdiff_t srcstride, const int16_t *src2, ptrdiff_t src2stride, int h, int fx, int fy, int w)
{
    int32_t v[64 * 64];
    MC_INTER(bd, taps, v, src, srcstride, w, h, fx, fy);
    DISPATCH(bd, mc_store_bi_8(bd, dst, dststride, v, src2, src2stride, w, h),
                 mc_store_bi_16(bd, (uint16_t *)dst, dststride / 2, v, src2, src2stride, w, h));
}

void oh_or_mc_uni_w(int bd, int taps, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src,
                    ptrdiff_t srcstride, int h, int denom, int wx, int ox, int fx, int fy, int w)
{
    int32_t v[64 * 64];
    MC_INTER(bd, taps, v, src, srcstride, w, h, fx, fy);
    DISPATCH(bd, mc_store_uni_w_8(bd, dst, dststride, v, w, h, denom, wx, ox),
                 mc_store_uni_w_16(bd, (uint16_t *)dst, dststride / 2, v, w, h, denom, wx, ox));
}

void oh_or_mc_bi_w(int bd, int taps, uint8_t *dst, ptrdiff_t dststride, const uint8_t *src,
                   ptrdiff_t srcstride, const int16_t *src2, ptrdiff_t src2stride, int h, int denom,
                   int wx0, int wx1, int ox0, int ox1, int fx, int fy, int w)
{
    int32_t v[64 * 64];
    MC_INTER(bd, taps, v, src, srcstride, w, h, fx, fy);
    DISPATCH(bd, mc_store_bi_w_8(bd, dst, dststride, v, src2, src2stride, w, h, denom, wx0, wx1, ox0, ox1),
                 mc_store_bi_w_16(bd, (uint16_t *)dst, dststride / 2, v, src2, src2stride, w, h, denom, wx0, wx1, ox0, ox1));
}

/* ------------------------------------------------------------------------------------------
 * intra / deblock / SAO slots
 * ---------------------------------------------------------------------------------------- */
void oh_or_pred_planar(int bd, uint8_t *dst, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int log2)
{
    DISPATCH(bd, pred_planar_8(dst, top, left, stride, log2),
                 pred_planar_16((uint16_t *)dst, (const uint16_t *)top, (const uint16_t *)left, stride, log2));
}
void oh_or_pred_dc(int bd, uint8_t *dst, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int log2, int c_idx)
{
    DISPATCH(bd, pred_dc_8(dst, top, left, stride, log2, c_idx),
                 pred_dc_16((uint16_t *)dst, (const uint16_t *)top, (const uint16_t *)left, stride, log2, c_idx));
}
void oh_or_pred_angular(int bd, uint8_t *dst, const uint8_t *top, const uint8_t *left, ptrdiff_t stride,
                        int log2, int c_idx, int mode)
{
    DISPATCH(bd, pred_angular_8(bd, dst, top, left, stride, log2, c_idx, mode),
                 pred_angular_16(bd, (uint16_t *)dst, (const uint16_t *)top, (const uint16_t *)left, stride, log2, c_idx, mode));
}
void oh_or_intra_pred(const OhPicParams *p, uint8_t *plane, ptrdiff_t stride, int pw, int ph,
                      int x, int y, int c_idx, int log2, int mode, int avail, const uint8_t *is_intra)
{
    DISPATCH(p->bit_depth, intra_pred_8(p, plane, stride, pw, ph, x, y, c_idx, log2, mode, avail, is_intra),
                           intra_pred_16(p, (uint16_t *)plane, stride / 2, pw, ph, x, y, c_idx, log2, mode, avail, is_intra));
}
void oh_or_loop_filter_luma(int bd, uint8_t *pix, ptrdiff_t xs, ptrdiff_t ys, int beta, const int *tc,
                            const uint8_t *no_p, const uint8_t *no_q)
{
    DISPATCH(bd, loop_filter_luma_8(bd, pix, xs, ys, beta, tc, no_p, no_q),
                 loop_filter_luma_16(bd, (uint16_t *)pix, xs / 2, ys / 2, beta, tc, no_p, no_q));
}
void oh_or_loop_filter_chroma(int bd, uint8_t *pix, ptrdiff_t xs, ptrdiff_t ys, const int *tc,
                              const uint8_t *no_p, const uint8_t *no_q)
{
    DISPATCH(bd, loop_filter_chroma_8(bd, pix, xs, ys, tc, no_p, no_q),
                 loop_filter_chroma_16(bd, (uint16_t *)pix, xs / 2, ys / 2, tc, no_p, no_q));
}
void oh_or_sao_band(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t ds, ptrdiff_t ss,
                    const int16_t *offset_val, int band_position, int w, int h)
{
    DISPATCH(bd, sao_band_8(bd, dst, src, ds, ss, offset_val, band_position, w, h),
                 sao_band_16(bd, (uint16_t *)dst, (const uint16_t *)src, ds / 2, ss / 2, offset_val, band_position, w, h));
}
void oh_or_sao_edge(int bd, uint8_t *dst, const uint8_t *src, ptrdiff_t ds, ptrdiff_t ss,
                    const int16_t *offset_val, int eo_class, const int *borders, int w, int h,
                    int restore, const uint8_t *ve, const uint8_t *he, const uint8_t *de)
{
    DISPATCH(bd, sao_edge_8(bd, dst, src, ds, ss, offset_val, eo_class, borders, w, h, restore, ve, he, de),
                 sao_edge_16(bd, (uint16_t *)dst, (const uint16_t *)src, ds / 2, ss / 2, offset_val, eo_class,
                             borders, w, h, restore, ve, he, de));
}

/* ==========================================================================================
 * picture level
 * ======================================================================================== */

static inline int px_get(const OhHostPic *pic, int c, int x, int y)
{
    const uint8_t *row = pic->data[c] + (ptrdiff_t)y * pic->stride[c];
    return pic->bit_depth > 8 ? ((const uint16_t *)row)[x] : row[x];
}

/* ---- pass 1: inter prediction (hevc.c:1641-1949 drivers; videodsp_template.c:26-101 edge
 * emulation == clamping the source coordinates to the picture) ---- */
static void fetch_window(const OhHostPic *ref, int c, int x0, int y0, int w, int h, uint16_t *win, int wstride)
{
    int pw = ref->width[c], ph = ref->height[c];
    for (int y = 0; y < h; y++) {
        int sy = oh_clip3(y0 + y, 0, ph - 1);
        for (int x = 0; x < w; x++)
            win[y * wstride + x] = (uint16_t)px_get(ref, c, oh_clip3(x0 + x, 0, pw - 1), sy);
    }
}

int oh_or_pass_inter(const OhFrame *f, OhHostPic *pics)
{
    const OhPicParams *p = &f->p;
    OhHostPic *cur = &pics[f->cur_pic];
    int bd = p->bit_depth;
    int nplanes = p->chroma_format_idc ? 3 : 1;

    for (uint32_t i = 0; i < f->n_pu; i++) {
        const OhPu *pu = &f->pu[i];
        const OhWeights *wp = pu->wp == OH_NO_WP ? NULL : &f->wp[pu->wp];
        for (int c = 0; c < nplanes; c++) {
            int hs = oh_hshift(p, c), vs = oh_vshift(p, c);
            int taps = c ? 4 : 8, before = taps / 2 - 1;
            int bx = pu->x >> hs, by = pu->y >> vs, bw = pu->w >> hs, bh = pu->h >> vs;
            int32_t v[2][64 * 64];
            int16_t v0_16[64 * 64];
            int used[2] = { pu->ref[0] != OH_NO_REF, pu->ref[1] != OH_NO_REF };
            for (int l = 0; l < 2; l++) {
                if (!used[l])
                    continue;
                const OhHostPic *ref = &pics[f->ref_pics[pu->ref[l]]];
                int mvx = pu->mv[l][0], mvy = pu->mv[l][1];
                int fx, fy, ix, iy;
                if (c == 0) {
                    fx = mvx & 3; fy = mvy & 3; ix = mvx >> 2; iy = mvy >> 2;
                } else {                                   /* hevc.c:1807-1813 */
                    fx = (mvx & ((1 << (2 + hs)) - 1)) << (1 - hs);
                    fy = (mvy & ((1 << (2 + vs)) - 1)) << (1 - vs);
                    ix = mvx >> (2 + hs); iy = mvy >> (2 + vs);
                }
                uint16_t win[(64 + 7) * (64 + 8)];
                int ws = 64 + 8;
                fetch_window(ref, c, bx + ix - before, by + iy - before, bw + taps - 1, bh + taps - 1, win, ws);
                mc_intermediate_16(bd, taps, v[l], win + before * ws + before, ws, bw, bh, fx, fy);
            }
            uint8_t *dst = cur->data[c] + (ptrdiff_t)by * cur->stride[c] + (ptrdiff_t)bx * (bd > 8 ? 2 : 1);
            ptrdiff_t ds = cur->stride[c];
            int denom = wp ? wp->log2_denom[c ? 1 : 0] : 0;
            if (used[0] && used[1]) {
                for (int k = 0; k < bw * bh; k++)
                    v0_16[k] = (int16_t)v[0][k];           /* put_hevc_*pel into int16 tmp, hevc.c:1761 */
                if (!wp)
                    DISPATCH(bd, mc_store_bi_8(bd, dst, ds, v[1], v0_16, bw, bw, bh),
                                 mc_store_bi_16(bd, (uint16_t *)dst, ds / 2, v[1], v0_16, bw, bw, bh));
                else
                    DISPATCH(bd, mc_store_bi_w_8(bd, dst, ds, v[1], v0_16, bw, bw, bh, denom,
                                                 wp->w[0][c], wp->w[1][c], wp->o[0][c], wp->o[1][c]),
                                 mc_store_bi_w_16(bd, (uint16_t *)dst, ds / 2, v[1], v0_16, bw, bw, bh, denom,
                                                  wp->w[0][c], wp->w[1][c], wp->o[0][c], wp->o[1][c]));
            } else if (used[0] || used[1]) {
                int l = used[0] ? 0 : 1;
                if (!wp)
                    DISPATCH(bd, mc_store_uni_8(bd, dst, ds, v[l], bw, bh),
                                 mc_store_uni_16(bd, (uint16_t *)dst, ds / 2, v[l], bw, bh));
                else
                    DISPATCH(bd, mc_store_uni_w_8(bd, dst, ds, v[l], bw, bh, denom, wp->w[l][c], wp->o[l][c]),
                                 mc_store_uni_w_16(bd, (uint16_t *)dst, ds / 2, v[l], bw, bh, denom, wp->w[l][c], wp->o[l][c]));
            } else {
                return -1;
            }
        }
    }
    return 0;
}

/* ---- pass 2: residual (hevc_cabac.c:1868-1949) ---- */
/* ---- sparse hand-off: de-quantisation of the parsed levels into the dense block, hevc_cabac.c:1478-1494 (scale, shift,
 * matrix choice) and :1818-1841 (per coefficient).  PARITY UNPINNED for this function: the reference statements sit
 * inside ff_hevc_hls_residual_coding between CABAC reads and cannot be called on their own; restated from the text. ---- */
static void tu_dequant(const OhFrame *f, const OhTu *tu, const uint32_t *rec, int16_t *c)
{
    static const uint8_t level_scale[6] = { 40, 45, 51, 57, 64, 72 };
    const int log2 = tu->log2_size, n = 1 << log2, cnt = (int)(rec[0] & 0xffff), qp = (int)((rec[0] >> 16) & 0xff);
    const unsigned matrix = rec[0] >> 24;
    const int shift = f->p.bit_depth + log2 - 5;            /* + 10 - log2_transform_range, the range being 15 (:1416) */
    /* the reference's rem6[] / div6[] tables (hevc_cabac.c:1428-1440) are declared with 76 entries and initialised with 74: QP 74 and 75
     * (12 bit: luma QP 50 / 51) read zeros, i.e. scale = level_scale[0] << 0 */
    const int qp6 = qp < 74 ? qp : 0;
    const int64_t add = (int64_t)1 << (shift - 1), scale = (int64_t)level_scale[qp6 % 6] << (qp6 / 6);
    const uint8_t *m = matrix != OH_FLAT_MATRIX && f->scaling ? f->scaling->sl[log2 - 2][matrix] : NULL;
    const int dc_scale = m && log2 >= 4 ? f->scaling->sl_dc[log2 - 4][matrix] : (m ? -1 : 16);
    memset(c, 0, sizeof(int16_t) * (size_t)(n * n));
    for (int i = 0; i < cnt; i++) {
        const int pos = (int)(rec[1 + i] & 0xffff), x = pos & (n - 1), y = pos >> log2;
        int64_t v = (int16_t)(rec[1 + i] >> 16);
        int scale_m = 16;
        if (m) {
            if (x || y || log2 < 4)
                scale_m = m[log2 == 3 ? (y << 3) + x : log2 == 4 ? ((y >> 1) << 3) + (x >> 1) : log2 == 5 ? ((y >> 2) << 3) + (x >> 2) : (y << 2) + x];
            else
                scale_m = dc_scale;
        }
        v = (v * scale * scale_m + add) >> shift;
        c[pos] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
    }
}

static void tu_inverse(int bd, const OhTu *tu, int16_t *c)
{
    int log2 = tu->log2_size;
    switch (tu->kind) {
    case OH_TU_IDCT:   oh_or_idct(bd, c, log2); break;
    case OH_TU_DST4:   oh_or_idct_4x4_luma(bd, c); break;
    case OH_TU_SKIP:
        if (tu->flags & OH_TUF_ROTATE)                    /* hevc_cabac.c:1879-1882 */
            for (int i = 0; i < 8; i++) { int16_t t = c[i]; c[i] = c[15 - i]; c[15 - i] = t; }
        oh_or_transform_skip(bd, c, log2);
        if (tu->flags & OH_TUF_RDPCM)
            oh_or_transform_rdpcm(c, log2, !!(tu->flags & OH_TUF_RDPCM_VER));
        break;
    case OH_TU_BYPASS:
        if (tu->flags & OH_TUF_RDPCM)
            oh_or_transform_rdpcm(c, log2, !!(tu->flags & OH_TUF_RDPCM_VER));
        break;
    default: break;
    }
}

static void tu_store(const OhFrame *f, OhHostPic *cur, const OhTu *tu, const int16_t *res)
{
    int bd = f->p.bit_depth, n = 1 << tu->log2_size;
    uint8_t *dst = cur->data[tu->c_idx] + (ptrdiff_t)tu->y * cur->stride[tu->c_idx] + (ptrdiff_t)tu->x * (bd > 8 ? 2 : 1);
    if (tu->kind == OH_TU_PCM) {                           /* put_pcm, hevcdsp_template.c:30-43 */
        for (int y = 0; y < n; y++)
            for (int x = 0; x < n; x++) {
                if (bd > 8) ((uint16_t *)(dst + y * cur->stride[tu->c_idx]))[x] = (uint16_t)res[y * n + x];
                else        dst[y * cur->stride[tu->c_idx] + x] = (uint8_t)res[y * n + x];
            }
    } else {
        oh_or_transform_add(bd, dst, res, cur->stride[tu->c_idx], tu->log2_size);
    }
}

int oh_or_pass_residual(const OhFrame *f, OhHostPic *pics, int16_t *coeffs)
{
    OhHostPic *cur = &pics[f->cur_pic];
    for (uint32_t i = 0; i < f->n_tu; i++) {
        const OhTu *tu = &f->tu[i];
        int16_t *c = coeffs + tu->coeff_off;
        if ((tu->flags & OH_TUF_SPARSE) && f->tu_sparse && f->tu_sparse[i] != OH_NO_COEFF)
            tu_dequant(f, tu, f->sparse + f->tu_sparse[i], c);
        tu_inverse(f->p.bit_depth, tu, c);
        /* cross-component prediction, hevc_cabac.c:1942-1947 (coded chroma block) and hevc.c:1319-1331, 1352-1364 (cbf 0):
         * the luma block of the transform unit precedes its chroma blocks in the list, its residual is already in the
         * pool.  PARITY UNPINNED for this statement: it is host code of hevc.c / hevc_cabac.c, outside the files that
         * can be compiled here; restated from the text (int16 storage, arithmetic shift). */
        if ((tu->flags & OH_TUF_CROSS) && f->tu_cross && f->tu_cross[i] != OH_NO_COEFF) {
            const OhTu *ty = &f->tu[f->tu_cross[i] & 0xffffff];
            const int scale = (int8_t)(f->tu_cross[i] >> 24), n2 = 1 << (2 * tu->log2_size);
            const int16_t *cy = coeffs + ty->coeff_off;
            for (int k = 0; k < n2; k++) c[k] = (int16_t)(c[k] + ((scale * cy[k]) >> 3));
        }
        if (tu->flags & OH_TUF_ADD_NOW)
            tu_store(f, cur, tu, c);
    }
    return 0;
}

/* ---- pass 3: intra prediction + deferred residual add, level by level ---- */
int oh_or_pass_intra(const OhFrame *f, OhHostPic *pics, const int16_t *residuals)
{
    OhHostPic *cur = &pics[f->cur_pic];
    for (uint32_t i = 0; i < f->n_intra; i++) {
        const OhIntra *it = &f->intra[i];
        int c = it->c_idx;
        oh_or_intra_pred(&f->p, cur->data[c], cur->stride[c], cur->width[c], cur->height[c],
                         it->x, it->y, c, it->log2_size, it->mode, it->avail, f->is_intra);
        if (it->tu != OH_NO_COEFF) {
            const OhTu *tu = &f->tu[it->tu];
            tu_store(f, cur, tu, residuals + tu->coeff_off);
        }
    }
    return 0;
}

/* ---- pass 4: deblocking (hevc_filter.c:345-581), restated as two whole-picture passes:
 * all vertical edges, then all horizontal edges.  The per-CTB driver's parameter quirks are kept
 * as functions of the edge position (see DESIGN.md "deblock parameter rules"). ---- */
static int get_qpy(const OhFrame *f, int x, int y)                       /* hevc_filter.c:143-149 */
{
    int l = f->p.log2_min_cb_size;
    return f->qp_y_tab[(x >> l) + (y >> l) * oh_min_cb_width(&f->p)];
}
static int get_pcm(const OhFrame *f, int x, int y)                       /* hevc_filter.c:324-338 */
{
    int l = f->p.log2_min_pu_size;
    if (x < 0 || y < 0)
        return 2;
    if ((x >> l) >= oh_min_pu_width(&f->p) || (y >> l) >= oh_min_pu_height(&f->p))
        return 2;
    return f->is_pcm ? f->is_pcm[(y >> l) * oh_min_pu_width(&f->p) + (x >> l)] : 0;
}
static int luma_tc(int qp, int bs, int tc_offset)                        /* TC_CALC, hevc_filter.c:340-343 */
{
    return oh_tc_table[oh_clip3(qp + 2 * (bs - 1) + (tc_offset >> 1 << 1), 0, 53)];
}
static int chroma_tc(const OhFrame *f, int qp_y, int c_idx, int tc_offset) /* hevc_filter.c:62-89 */
{
    static const uint8_t qp_c[14] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };
    int qp_i = oh_clip3(qp_y + (c_idx == 1 ? f->p.cb_qp_offset : f->p.cr_qp_offset), 0, 57);
    int qp;
    if (f->p.chroma_format_idc == 1)
        qp = qp_i < 30 ? qp_i : (qp_i > 43 ? qp_i - 6 : qp_c[qp_i - 30]);
    else
        qp = oh_clip3(qp_i, 0, 51);
    return oh_tc_table[oh_clip3(qp + 2 + tc_offset, 0, 53)];
}

/* The reference's CTB driver order leaks into the result in ONE configuration: 16x16 CTBs with horizontally subsampled chroma
 * (chroma CTBs 8 samples wide).  deblocking_filter_CTB filters the chroma horizontal edges over [x0 - 16, x_end - 16) luma samples
 * (hevc_filter.c:526-530), so the chroma columns of CTB X get their horizontal-edge filtering inside the call for CTB X + 1 — and
 * sao_filter_CTB(X - 1, Y - 1) runs right after deblocking_filter_CTB(X, Y) (ff_hevc_hls_filter, :1027-1052), i.e. BEFORE the
 * call for X + 1: the SAO of CTB (cx, cy) copies the first chroma column of its right neighbour (copy_CTB, :305-307) while the
 * horizontal edges of the CTB rows r >= min(cy + 1, ctb_height - 2) have not touched that column yet (SAO(cx, cy) is triggered by
 * decoding CTB (cx + 2, min(cy + 2, last row)), the deblocking call for CTB (cx + 2, r) by CTB (cx + 3, min(r + 1, last row)):
 * in the last two CTB rows the lag between the two shrinks to one CTB; and when cx + 2 is the LAST CTB column, the deblocking call
 * for it comes from the row-end branch of ff_hevc_hls_filters, a CTB earlier: then r >= min(cy + 1, ctb_height - 1)).  With wider chroma CTBs, and for luma, the own first columns of a CTB are filtered in its own call and
 * nothing is pending.  To reproduce it the deblock pass keeps the chroma planes as they are before its horizontal chroma
 * edges; the SAO pass of the SAME picture (next call on this thread) patches the neighbour column from that. */
static __thread struct { uint8_t *px[3]; size_t bytes[3]; const void *pic; int w, h, valid; } g_pre_h;
static int sao_sees_stale_column(const OhPicParams *p) { return p->log2_ctb_size == 4 && oh_hshift(p, 1) == 1; }

int oh_or_pass_deblock(const OhFrame *f, OhHostPic *pics)
{
    const OhPicParams *p = &f->p;
    OhHostPic *cur = &pics[f->cur_pic];
    g_pre_h.valid = 0;
    if (!p->deblock_enabled)
        return 0;
    int bd = p->bit_depth, bpp = bd > 8 ? 2 : 1;
    int W = p->width, H = p->height, bsw = W >> 2;
    int lc = p->log2_ctb_size, ctbw = oh_ctb_width(p);
    int pcmf = p->pcm_loop_filter_disable || p->transquant_bypass_enable;
    int hs = oh_hshift(p, 1), vs = oh_vshift(p, 1), hh = 1 << hs, vv = 1 << vs;
    uint8_t no_p[2] = { 0, 0 }, no_q[2] = { 0, 0 };
    int tc[2], tc2[2];

    /* vertical edges: every parameter comes from the CTB that contains the edge (the Q side) */
    for (int y = 0; y < H; y += 8)
        for (int x = 8; x < W; x += 8) {
            int bs0 = f->vertical_bs[(x + y * bsw) >> 2], bs1 = f->vertical_bs[(x + (y + 4) * bsw) >> 2];
            if (!bs0 && !bs1)
                continue;
            const OhDeblockCtb *db = &f->deblock[(y >> lc) * ctbw + (x >> lc)];
            int qp = (get_qpy(f, x - 1, y) + get_qpy(f, x, y) + 1) >> 1;
            int beta = oh_beta_table[oh_clip3(qp + db->beta_offset, 0, 51)];
            tc[0] = bs0 ? luma_tc(qp, bs0, db->tc_offset) : 0;
            tc[1] = bs1 ? luma_tc(qp, bs1, db->tc_offset) : 0;
            if (pcmf) {
                no_p[0] = (uint8_t)get_pcm(f, x - 1, y); no_p[1] = (uint8_t)get_pcm(f, x - 1, y + 4);
                no_q[0] = (uint8_t)get_pcm(f, x, y);     no_q[1] = (uint8_t)get_pcm(f, x, y + 4);
            }
            oh_or_loop_filter_luma(bd, cur->data[0] + (ptrdiff_t)y * cur->stride[0] + x * bpp,
                                   bpp, cur->stride[0], beta, tc, no_p, no_q);
        }
    if (p->chroma_format_idc)
        for (int y = 0; y < H; y += 8 * vv)
            for (int x = 8 * hh; x < W; x += 8 * hh) {
                int bs0 = f->vertical_bs[(x + y * bsw) >> 2], bs1 = f->vertical_bs[(x + (y + 4 * vv) * bsw) >> 2];
                if (bs0 != 2 && bs1 != 2)
                    continue;
                const OhDeblockCtb *db = &f->deblock[(y >> lc) * ctbw + (x >> lc)];
                int qp0 = (get_qpy(f, x - 1, y) + get_qpy(f, x, y) + 1) >> 1;
                int qp1 = (get_qpy(f, x - 1, y + 4 * vv) + get_qpy(f, x, y + 4 * vv) + 1) >> 1;
                tc[0]  = bs0 == 2 ? chroma_tc(f, qp0, 1, db->tc_offset) : 0;
                tc[1]  = bs1 == 2 ? chroma_tc(f, qp1, 1, db->tc_offset) : 0;
                tc2[0] = bs0 == 2 ? chroma_tc(f, qp0, 2, db->tc_offset) : 0;
                tc2[1] = bs1 == 2 ? chroma_tc(f, qp1, 2, db->tc_offset) : 0;
                if (pcmf) {
                    no_p[0] = (uint8_t)get_pcm(f, x - 1, y); no_p[1] = (uint8_t)get_pcm(f, x - 1, y + 4 * vv);
                    no_q[0] = (uint8_t)get_pcm(f, x, y);     no_q[1] = (uint8_t)get_pcm(f, x, y + 4 * vv);
                }
                oh_or_loop_filter_chroma(bd, cur->data[1] + (ptrdiff_t)(y >> vs) * cur->stride[1] + (x >> hs) * bpp,
                                         bpp, cur->stride[1], tc, no_p, no_q);
                oh_or_loop_filter_chroma(bd, cur->data[2] + (ptrdiff_t)(y >> vs) * cur->stride[2] + (x >> hs) * bpp,
                                         bpp, cur->stride[2], tc2, no_p, no_q);
            }

    /* horizontal edges.  The reference filters [x0-8, x_end-8) inside CTB (x0,..)'s call, so the
     * "processing CTB" of an edge at x is the one containing x+8 (x+8h for chroma), capped to the
     * last column; tc comes from the processing CTB, beta from the CTB that contains x
     * (hevc_filter.c:481-520), chroma's first segment takes tc from the CTB containing x and its
     * second segment from the processing CTB (:523-580). */
    for (int y = 8; y < H; y += 8)
        for (int x = 0; x < W; x += 8) {
            int bs0 = f->horizontal_bs[(x + y * bsw) >> 2], bs1 = f->horizontal_bs[((x + 4) + y * bsw) >> 2];
            if (!bs0 && !bs1)
                continue;
            int pcx = (x + 8) >> lc; if (pcx > ctbw - 1) pcx = ctbw - 1;
            const OhDeblockCtb *dbp = &f->deblock[(y >> lc) * ctbw + pcx];
            const OhDeblockCtb *dbx = &f->deblock[(y >> lc) * ctbw + (x >> lc)];
            int qp = (get_qpy(f, x, y - 1) + get_qpy(f, x, y) + 1) >> 1;
            int beta = oh_beta_table[oh_clip3(qp + dbx->beta_offset, 0, 51)];
            tc[0] = bs0 ? luma_tc(qp, bs0, dbp->tc_offset) : 0;
            tc[1] = bs1 ? luma_tc(qp, bs1, dbp->tc_offset) : 0;
            if (pcmf) {
                no_p[0] = (uint8_t)get_pcm(f, x, y - 1); no_p[1] = (uint8_t)get_pcm(f, x + 4, y - 1);
                no_q[0] = (uint8_t)get_pcm(f, x, y);     no_q[1] = (uint8_t)get_pcm(f, x + 4, y);
            }
            oh_or_loop_filter_luma(bd, cur->data[0] + (ptrdiff_t)y * cur->stride[0] + x * bpp,
                                   cur->stride[0], bpp, beta, tc, no_p, no_q);
        }
    if (p->chroma_format_idc && sao_sees_stale_column(p)) {
        for (int c = 1; c < 3; c++) {
            size_t n = (size_t)cur->stride[c] * cur->height[c];
            if (g_pre_h.bytes[c] < n) { free(g_pre_h.px[c]); g_pre_h.px[c] = (uint8_t *)malloc(n); g_pre_h.bytes[c] = g_pre_h.px[c] ? n : 0; }
            if (!g_pre_h.px[c]) return -1;
            memcpy(g_pre_h.px[c], cur->data[c], n);
        }
        g_pre_h.pic = cur->data[0]; g_pre_h.w = W; g_pre_h.h = H; g_pre_h.valid = 1;
    }
    if (p->chroma_format_idc)
        for (int y = 8 * vv; y < H; y += 8 * vv)
            for (int x = 0; x < W; x += 8 * hh) {
                int bs0 = f->horizontal_bs[(x + y * bsw) >> 2], bs1 = f->horizontal_bs[((x + 4 * hh) + y * bsw) >> 2];
                if (bs0 != 2 && bs1 != 2)
                    continue;
                int pcx = (x + 8 * hh) >> lc; if (pcx > ctbw - 1) pcx = ctbw - 1;
                int tco_p = f->deblock[(y >> lc) * ctbw + pcx].tc_offset;
                int tco_x = f->deblock[(y >> lc) * ctbw + (x >> lc)].tc_offset;
                int qp0 = bs0 == 2 ? (get_qpy(f, x, y - 1) + get_qpy(f, x, y) + 1) >> 1 : 0;
                int qp1 = bs1 == 2 ? (get_qpy(f, x + 4 * hh, y - 1) + get_qpy(f, x + 4 * hh, y) + 1) >> 1 : 0;
                tc[0]  = bs0 == 2 ? chroma_tc(f, qp0, 1, tco_x) : 0;
                tc[1]  = bs1 == 2 ? chroma_tc(f, qp1, 1, tco_p) : 0;
                tc2[0] = bs0 == 2 ? chroma_tc(f, qp0, 2, tco_x) : 0;
                tc2[1] = bs1 == 2 ? chroma_tc(f, qp1, 2, tco_p) : 0;
                if (pcmf) {
                    no_p[0] = (uint8_t)get_pcm(f, x, y - 1); no_p[1] = (uint8_t)get_pcm(f, x + 4 * hh, y - 1);
                    no_q[0] = (uint8_t)get_pcm(f, x, y);     no_q[1] = (uint8_t)get_pcm(f, x + 4 * hh, y);
                }
                oh_or_loop_filter_chroma(bd, cur->data[1] + (ptrdiff_t)(y >> vs) * cur->stride[1] + (x >> hs) * bpp,
                                         cur->stride[1], bpp, tc, no_p, no_q);
                oh_or_loop_filter_chroma(bd, cur->data[2] + (ptrdiff_t)(y >> vs) * cur->stride[2] + (x >> hs) * bpp,
                                         cur->stride[2], bpp, tc2, no_p, no_q);
            }
    return 0;
}

/* ---- pass 5: SAO (hevc_filter.c:197-322) restated as one whole-picture pass from a copy of the
 * deblocked picture (the reference keeps that copy CTB by CTB in s->sao_frame) ---- */
int oh_or_pass_sao(const OhFrame *f, OhHostPic *pics)
{
    const OhPicParams *p = &f->p;
    OhHostPic *cur = &pics[f->cur_pic];
    if (!p->sao_enabled || !f->sao)
        return 0;
    int bd = p->bit_depth, bpp = bd > 8 ? 2 : 1;
    int ctbw = oh_ctb_width(p), ctbh = oh_ctb_height(p), lc = p->log2_ctb_size;
    int nplanes = p->chroma_format_idc ? 3 : 1;
    int tqb = p->transquant_bypass_enable || p->pcm_loop_filter_disable;

    for (int c = 0; c < nplanes; c++) {
        int hs = oh_hshift(p, c), vs = oh_vshift(p, c);
        int pw = cur->width[c], ph = cur->height[c];
        ptrdiff_t st = cur->stride[c];
        uint8_t *copy = (uint8_t *)malloc((size_t)st * ph);
        if (!copy)
            return -1;
        memcpy(copy, cur->data[c], (size_t)st * ph);
        for (int cy = 0; cy < ctbh; cy++)
            for (int cx = 0; cx < ctbw; cx++) {
                const OhSaoCtb *s = &f->sao[cy * ctbw + cx];
                int x0 = (cx << lc) >> hs, y0 = (cy << lc) >> vs;
                int w = (1 << lc) >> hs, h = (1 << lc) >> vs;
                if (w > pw - x0) w = pw - x0;
                if (h > ph - y0) h = ph - y0;
                int borders[4] = { cx == 0, cy == 0, cx == ctbw - 1, cy == ctbh - 1 };
                uint8_t ve[2] = { (uint8_t)(s->edge_flags & 1), (uint8_t)((s->edge_flags >> 1) & 1) };
                uint8_t he[2] = { (uint8_t)((s->edge_flags >> 2) & 1), (uint8_t)((s->edge_flags >> 3) & 1) };
                uint8_t de[4] = { (uint8_t)((s->edge_flags >> 4) & 1), (uint8_t)((s->edge_flags >> 5) & 1),
                                  (uint8_t)((s->edge_flags >> 6) & 1), (uint8_t)((s->edge_flags >> 7) & 1) };
                uint8_t *dst = cur->data[c] + (ptrdiff_t)y0 * st + x0 * bpp;
                const uint8_t *src = copy + (ptrdiff_t)y0 * st + x0 * bpp;
                if (s->type_idx[c] == 1)
                    oh_or_sao_band(bd, dst, src, st, st, s->offset_val[c], s->band_position[c], w, h);
                else if (s->type_idx[c] == 2) {
                    /* 16x16 CTBs, subsampled chroma: the right neighbour's first column as the reference's driver order shows it
                     * (comment at g_pre_h): rows touched by the horizontal edges of CTB rows >= r_pending are not filtered yet */
                    const int stale = c && sao_sees_stale_column(p) && p->deblock_enabled && g_pre_h.valid && g_pre_h.pic == cur->data[0] &&
                                      g_pre_h.w == p->width && g_pre_h.h == p->height && cx + 2 < ctbw;
                    uint8_t saved[2 * (16 + 2)];
                    /* first CTB row whose horizontal edges the neighbour column has not seen yet.  When the neighbour's own right
                     * neighbour is the LAST CTB column, its deblocking call comes from the row-end branch of ff_hevc_hls_filters
                     * (hevc_filter.c:1058-1059), which in the last two CTB rows runs BEFORE the call that triggers this SAO: there only
                     * the rows from cy + 1 on are pending */
                    const int r_lag = cx + 2 == ctbw - 1 ? ctbh - 1 : (ctbh >= 2 ? ctbh - 2 : 0);
                    const int r_pending = cy + 1 < r_lag ? cy + 1 : r_lag, ya = y0 > 0 ? y0 - 1 : 0, yb = y0 + h < ph ? y0 + h : ph - 1;
                    /* bit 0: the edges of CTB row cy pending, bit 1: those of row cy + 1 — from the work list when the picture was not
                     * decoded in raster order (tiles: OhFrame.sao_pending), else the closed form above */
                    const int pend = f->sao_pending ? f->sao_pending[cy * ctbw + cx] : (cy >= r_pending ? 1 : 0) | (cy + 1 >= r_pending ? 2 : 0);
                    if (stale)
                        for (int y = ya; y <= yb; y++) {
                            uint8_t *px = copy + (ptrdiff_t)y * st + (x0 + w) * bpp;
                            memcpy(saved + (y - ya) * bpp, px, (size_t)bpp);
                            const int ye = (y & 7) == 7 ? y + 1 : y;                    /* the horizontal edge that touches row y: p0 row or q0 row */
                            if ((ye & 7) == 0 && ye > 0 && ((pend >> (((ye << vs) >> lc) > cy)) & 1))
                                memcpy(px, g_pre_h.px[c] + (ptrdiff_t)y * st + (x0 + w) * bpp, (size_t)bpp);
                        }
                    oh_or_sao_edge(bd, dst, src, st, st, s->offset_val[c], s->eo_class[c], borders, w, h,
                                   s->edge_flags != 0, ve, he, de);
                    if (stale)
                        for (int y = ya; y <= yb; y++)
                            memcpy(copy + (ptrdiff_t)y * st + (x0 + w) * bpp, saved + (y - ya) * bpp, (size_t)bpp);
                } else
                    continue;
                /* restore_tqb_pixels (hevc_filter.c:163-193): called with the CTB's LUMA origin but
                 * the COMPONENT's width/height, so for subsampled chroma only the min-PUs of the
                 * luma rectangle [X0, X0+w) x [Y0, Y0+h) are restored — kept as is. */
                if (tqb && f->is_pcm) {
                    int l = p->log2_min_pu_size, mpw = oh_min_pu_width(p);
                    int X0 = cx << lc, Y0 = cy << lc;
                    for (int py = Y0 >> l; py < (Y0 + h) >> l; py++)
                        for (int px = X0 >> l; px < (X0 + w) >> l; px++) {
                            if (!f->is_pcm[py * mpw + px])
                                continue;
                            int sx = (px << l) >> hs, sy = (py << l) >> vs;
                            int len = (1 << l) >> hs;   /* BYTES, not samples: the reference's memcpy
                                                           length ignores pixel_shift (:177,:185) */
                            for (int n = 0; n < (1 << l) >> vs; n++)
                                memcpy(cur->data[c] + (ptrdiff_t)(sy + n) * st + sx * bpp,
                                       copy + (ptrdiff_t)(sy + n) * st + sx * bpp, (size_t)len);
                        }
                }
            }
        free(copy);
    }
    return 0;
}

int oh_or_bs_derive(const OhPicParams *p, const OhBsInputs *in, uint8_t *vbs, uint8_t *hbs);

int oh_or_frame(const OhFrame *f, OhHostPic *pics)
{
    int16_t *c = (int16_t *)malloc(sizeof(int16_t) * (size_t)(f->n_coeff ? f->n_coeff : 1));
    int r = 0;
    if (!c)
        return -1;
    if (f->n_coeff)
        memcpy(c, f->coeffs, sizeof(int16_t) * (size_t)f->n_coeff);
    if (!r) r = oh_or_pass_inter(f, pics);
    if (!r) r = oh_or_pass_residual(f, pics, c);
    if (!r) r = oh_or_pass_intra(f, pics, c);
    if (!r && f->bs_in && f->p.deblock_enabled) {                 /* the grids come from the motion field (oh_or_bs_derive below) */
        OhFrame g = *f;
        uint8_t *v = (uint8_t *)malloc(oh_bs_size(&f->p)), *h = (uint8_t *)malloc(oh_bs_size(&f->p));
        r = (v && h) ? oh_or_bs_derive(&f->p, f->bs_in, v, h) : -1;
        g.vertical_bs = v; g.horizontal_bs = h; g.bs_size = oh_bs_size(&f->p);
        if (!r) r = oh_or_pass_deblock(&g, pics);
        free(v); free(h);
    } else if (!r && (f->flags & OH_FRAME_BS_PACKED) && f->p.deblock_enabled && f->vertical_bs && f->horizontal_bs) {
        /* the grids travel four strengths to the byte (OhFrame.flags): back to the reference's byte grids for the pass */
        OhFrame g = *f;
        const size_t n = f->bs_size;
        uint8_t *v = (uint8_t *)malloc(n ? n : 1), *h = (uint8_t *)malloc(n ? n : 1);
        if (v && h) {
            for (size_t i = 0; i < n; i++) {
                v[i] = (uint8_t)(f->vertical_bs[i >> 2] >> (2 * (i & 3)) & 3);
                h[i] = (uint8_t)(f->horizontal_bs[i >> 2] >> (2 * (i & 3)) & 3);
            }
            g.vertical_bs = v; g.horizontal_bs = h;
            r = oh_or_pass_deblock(&g, pics);
        } else
            r = -1;
        free(v); free(h);
    } else if (!r) r = oh_or_pass_deblock(f, pics);
    if (!r) r = oh_or_pass_sao(f, pics);
    free(c);
    return r;
}

/* =====================================================================================================
 * SHVC inter-layer up-sampling.  Restated from hevcdsp_template.c:1834-2438 and the
 * filter tables hevcdsp.c:948-1024: 16-phase 8-tap (luma) / 4-tap (chroma) separable resampling, the x2 and
 * x1.5 slot variants pick their phases by parity / modulo 3 instead of the 16.16 position.
 * ===================================================================================================== */
static const int8_t up_luma16[16][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 63, 4, -2, 1, 0 }, { -1, 2, -5, 62, 8, -3, 1, 0 }, { -1, 3, -8, 60, 13, -4, 1, 0 },
    { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 52, 26, -8, 3, -1 }, { -1, 3, -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 },
    { -1, 4, -11, 40, 40, -11, 4, -1 }, { -1, 4, -10, 34, 45, -11, 4, -1 }, { -1, 4, -10, 31, 47, -9, 3, -1 }, { -1, 3, -8, 26, 52, -11, 4, -1 },
    { 0, 1, -5, 17, 58, -10, 4, -1 }, { 0, 1, -4, 13, 60, -8, 3, -1 }, { 0, 1, -3, 8, 62, -5, 2, -1 }, { 0, 1, -2, 4, 63, -3, 1, 0 } };
static const int8_t up_chroma16[16][4] = {
    { 0, 64, 0, 0 }, { -2, 62, 4, 0 }, { -2, 58, 10, -2 }, { -4, 56, 14, -2 }, { -4, 54, 16, -2 }, { -6, 52, 20, -2 }, { -6, 46, 28, -4 }, { -4, 42, 30, -4 },
    { -4, 36, 36, -4 }, { -4, 30, 42, -4 }, { -4, 28, 46, -6 }, { -2, 20, 52, -6 }, { -2, 16, 54, -4 }, { -2, 14, 56, -4 }, { -2, 10, 58, -2 }, { 0, 4, 62, -2 } };
/* the variants' own small tables are rows of the 16-phase ones: phase of luma x2 {0,8}, x1.5 {0,11,5};
 * chroma h x2 {0,8}, x1.5 {0,11,5}; chroma v x2 {14,6}, x1.5 {15,9,4} (hevcdsp.c:988-1022) */
static const int up_lx2[2] = { 0, 8 }, up_lx15[3] = { 0, 11, 5 }, up_cvx2[2] = { 14, 6 }, up_cvx15[3] = { 15, 9, 4 };
#define UP_SHIFT 12                              /* N_SHIFT = 20 - 8 whatever the bit depth, hevcdsp.h:40 */

static inline int up_get(const uint8_t *p, int bd, ptrdiff_t i) { return bd > 8 ? ((const uint16_t *)p)[i] : p[i]; }
static inline void up_put(uint8_t *p, int bd, ptrdiff_t i, int v)
{
    v = v < 0 ? 0 : (v > (1 << bd) - 1 ? (1 << bd) - 1 : v);
    if (bd > 8) ((uint16_t *)p)[i] = (uint16_t)v; else p[i] = (uint8_t)v;
}

void oh_or_up_luma_h(int variant, int bd, int16_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                     int x_el, int x_bl, int block_w, int block_h, int width_el, const OhUpsample *u)
{
    const int left = u->win_left, right_end = width_el - u->win_right;
    for (int i = 0; i < block_w; i++) {
        int x = oh_clip3(i + x_el, left, right_end), phase, pos;                  /* upper bound inclusive: :1852 */
        if (variant == OH_UP_X2) { phase = up_lx2[x & 1]; pos = ((x - left) >> 1) - x_bl; }
        else if (variant == OH_UP_X1_5) { phase = up_lx15[(x - left) % 3]; pos = (((x - left) << 1) / 3) - x_bl; }
        else { int r16 = ((x - left) * u->scale_x_lum + u->add_x_lum) >> 12; phase = r16 & 15; pos = (r16 >> 4) - x_bl; }
        for (int j = 0; j < block_h; j++) {
            int s = 0;
            for (int k = 0; k < 8; k++) s += up_luma16[phase][k] * up_get(src, bd, j * srcstride + pos + k - 3);
            dst[j * dststride + i] = (int16_t)s;
        }
    }
}

void oh_or_up_cr_h(int variant, int bd, int16_t *dst, ptrdiff_t dststride, const uint8_t *src, ptrdiff_t srcstride,
                   int x_el, int x_bl, int block_w, int block_h, int width_el, const OhUpsample *u)
{
    const int left = u->win_left >> 1, right_end = width_el - (u->win_right >> 1);
    for (int i = 0; i < block_w; i++) {
        int x = oh_clip3(i + x_el, left, right_end), phase, pos;
        if (variant == OH_UP_X2) { phase = up_lx2[x & 1]; pos = (x >> 1) - x_bl; }                /* no window offset: :2014 */
        else if (variant == OH_UP_X1_5) { phase = up_lx15[(x - left) % 3]; pos = (((x - left) << 1) / 3) - x_bl; }
        else { int r16 = ((x - left) * u->scale_x_cr + u->add_x_cr) >> 12; phase = r16 & 15; pos = (r16 >> 4) - x_bl; }
        for (int j = 0; j < block_h; j++) {
            int s = 0;
            for (int k = 0; k < 4; k++) s += up_chroma16[phase][k] * up_get(src, bd, j * srcstride + pos + k - 1);
            dst[j * dststride + i] = (int16_t)s;
        }
    }
}

void oh_or_up_luma_v(int variant, int bd, uint8_t *dst, ptrdiff_t dststride, const int16_t *src, ptrdiff_t srcstride,
                     int y_bl, int x_el, int y_el, int block_w, int block_h, int width_el, int height_el, const OhUpsample *u)
{
    const int top = u->win_top, bottom_end = height_el - u->win_bottom, right_end = width_el - u->win_right, left = u->win_left;
    for (int j = 0; j < block_h; j++) {
        int y = oh_clip3(y_el + j, top, bottom_end - 1), phase, row;
        if (variant == OH_UP_X2) { phase = up_lx2[(y - top) & 1]; row = ((y - top) >> 1) - y_bl; }
        else if (variant == OH_UP_X1_5) { phase = up_lx15[(y - top) % 3]; row = (((y - top) << 1) / 3) - y_bl; }
        else { int r16 = ((y - top) * u->scale_y_lum + u->add_y_lum) >> 12; phase = r16 & 15; row = (r16 >> 4) - y_bl; }
        int col = 0;                                 /* the source column only advances inside the window: :1925 */
        for (int i = 0; i < block_w; i++) {
            int s = 0;
            for (int k = 0; k < 8; k++) s += up_luma16[phase][k] * src[(row + k - 3) * srcstride + col];
            up_put(dst, bd, (ptrdiff_t)(y_el + j) * dststride + x_el + i, (s + (1 << (UP_SHIFT - 1))) >> UP_SHIFT);
            if (x_el + i >= left && x_el + i <= right_end - 2) col++;
        }
    }
}

void oh_or_up_cr_v(int variant, int bd, uint8_t *dst, ptrdiff_t dststride, const int16_t *src, ptrdiff_t srcstride,
                   int y_bl, int x_el, int y_el, int block_w, int block_h, int width_el, int height_el, const OhUpsample *u)
{
    const int left = u->win_left >> 1, right_end = width_el - (u->win_right >> 1), top = u->win_top >> 1, bottom_end = height_el - (u->win_bottom >> 1);
    for (int j = 0; j < block_h; j++) {
        int y = oh_clip3(y_el + j, top, bottom_end - 1);
        int r16 = (((y - top) * u->scale_y_cr + u->add_y_cr) >> 12) - 4;
        int phase = variant == OH_UP_X2 ? up_cvx2[y & 1] : (variant == OH_UP_X1_5 ? up_cvx15[y % 3] : (r16 & 15));
        int row = (r16 >> 4) - y_bl, col = 0;
        for (int i = 0; i < block_w; i++) {
            int s = 0;
            for (int k = 0; k < 4; k++) s += up_chroma16[phase][k] * src[(row + k - 1) * srcstride + col];
            up_put(dst, bd, (ptrdiff_t)y * dststride + x_el + i, (s + (1 << (UP_SHIFT - 1))) >> UP_SHIFT);      /* row y, the CLIPPED one: :1952 */
            if (x_el + i >= left && x_el + i <= right_end - 2) col++;
        }
    }
}

/* one plane of upsample_base_layer_frame: horizontal pass into tmp[h_bl][w_el] (int16), vertical pass to the picture */
static void up_frame_plane(const uint8_t *src, ptrdiff_t sstride, int w_bl, int h_bl, uint8_t *dst, ptrdiff_t dstride, int w_el, int h_el,
                           int taps, int left, int right_end_h, int right_end_v, int top, int bottom_end,
                           int scale_x, int add_x, int scale_y, int add_y, int y_bias, int16_t *tmp)
{
    const int before = taps / 2 - 1;
    for (int i = 0; i < w_el; i++) {
        int x = oh_clip3(i, left, right_end_h);
        int r16 = ((x - left) * scale_x + add_x) >> 12, phase = r16 & 15, pos = (r16 >> 4) - before;
        for (int j = 0; j < h_bl; j++) {
            int s = 0;
            for (int k = 0; k < taps; k++) {
                int c = taps == 8 ? up_luma16[phase][k] : up_chroma16[phase][k];
                s += c * src[j * sstride + oh_clip3(pos + k, 0, w_bl - 1)];           /* memset/memcpy edge buffers = clamping */
            }
            tmp[j * w_el + i] = (int16_t)s;
        }
    }
    for (int j = 0; j < h_el; j++) {
        int y = oh_clip3(j, top, bottom_end - 1);
        int r16 = (((y - top) * scale_y + add_y) >> 12) - y_bias, phase = r16 & 15, pos = (r16 >> 4) - before;
        for (int i = 0; i < w_el; i++) {
            int col = oh_clip3(i, left, right_end_v - 1) - left;                      /* srcY1++ only inside [left, right_end - 2] */
            int s = 0;
            for (int k = 0; k < taps; k++) {
                int c = taps == 8 ? up_luma16[phase][k] : up_chroma16[phase][k];
                s += c * tmp[oh_clip3(pos + k, 0, h_bl - 1) * w_el + col];
            }
            s = (s + (1 << (UP_SHIFT - 1))) >> UP_SHIFT;
            dst[j * dstride + i] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
        }
    }
}

int oh_or_upsample_frame(const OhHostPic *bl, OhHostPic *el, const OhUpsample *u)
{
    if (bl->bit_depth != 8 || el->bit_depth != 8 || !bl->data[1] || !el->data[1])
        return -1;
    const int w_el = el->width[0], h_el = el->height[0], w_bl = bl->width[0];
    int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)w_el * (size_t)(bl->height[0] > h_el ? bl->height[0] : h_el));
    if (!tmp)
        return -1;
    /* luma: heightBL = min(BL height, EL height) (:2220); x clipped to [left, right_end] inclusive (:2223) */
    up_frame_plane(bl->data[0], bl->stride[0], w_bl, bl->height[0] <= h_el ? bl->height[0] : h_el, el->data[0], el->stride[0], w_el, h_el, 8,
                   u->win_left, w_el - u->win_right, w_el - u->win_right, u->win_top, h_el - u->win_bottom,
                   u->scale_x_lum, u->add_x_lum, u->scale_y_lum, u->add_y_lum, 0, tmp);
    /* chroma: heightBL = max(BL height, EL chroma height) >> 1 (:2317-2320); x clipped to [left, right_end - 1] (:2324);
     * the vertical position carries the -4 of :2384 */
    const int wc_el = w_el >> 1, hc_el = h_el >> 1, wc_bl = w_bl >> 1;
    const int hc_bl = (bl->height[0] > hc_el ? bl->height[0] : hc_el) >> 1;
    const int left_c = u->win_left >> 1, right_end_c = wc_el - (u->win_right >> 1), top_c = u->win_top >> 1, bottom_end_c = hc_el - (u->win_bottom >> 1);
    for (int c = 1; c <= 2; c++)
        up_frame_plane(bl->data[c], bl->stride[c], wc_bl, hc_bl, el->data[c], el->stride[c], wc_el, hc_el, 4,
                       left_c, right_end_c - 1, right_end_c, top_c, bottom_end_c,
                       u->scale_x_cr, u->add_x_cr, u->scale_y_cr, u->add_y_cr, 4, tmp);
    free(tmp);
    return 0;
}

/* =========================================================================================
 * boundary strengths: ff_hevc_deblocking_boundary_strengths, hevc_filter.c:805-941, over the maps
 * it reads, one call per block in raster order of the block origins (every edge segment is written by exactly one call,
 * so the order of the calls does not matter).  boundary_strength() is the TEST_MV_POC / memcmp build (:584-700).
 * ======================================================================================= */
static int bs_abs4(int a, int b) { int d = a - b; return (d < 0 ? -d : d) >= 4; }

static int bs_motion(const OhMvField *curr, const OhMvField *neigh)
{
    if (memcmp(curr, neigh, sizeof(OhMvField)) == 0)                                            /* :600 */
        return 0;
    if (curr->pred_flag == 3 && neigh->pred_flag == 3) {                                        /* both PF_BI, :603 */
        if (curr->poc[0] == neigh->poc[0] && curr->poc[0] == curr->poc[1] && neigh->poc[0] == neigh->poc[1]) {       /* :605-607 */
            const int straight = bs_abs4(neigh->mv[0][0], curr->mv[0][0]) || bs_abs4(neigh->mv[0][1], curr->mv[0][1]) ||
                                 bs_abs4(neigh->mv[1][0], curr->mv[1][0]) || bs_abs4(neigh->mv[1][1], curr->mv[1][1]);
            const int crossed  = bs_abs4(neigh->mv[1][0], curr->mv[0][0]) || bs_abs4(neigh->mv[1][1], curr->mv[0][1]) ||
                                 bs_abs4(neigh->mv[0][0], curr->mv[1][0]) || bs_abs4(neigh->mv[0][1], curr->mv[1][1]);
            return straight && crossed;                                                         /* :623-630 */
        } else if (neigh->poc[0] == curr->poc[0] && neigh->poc[1] == curr->poc[1]) {            /* :632-650 */
            return bs_abs4(neigh->mv[0][0], curr->mv[0][0]) || bs_abs4(neigh->mv[0][1], curr->mv[0][1]) ||
                   bs_abs4(neigh->mv[1][0], curr->mv[1][0]) || bs_abs4(neigh->mv[1][1], curr->mv[1][1]);
        } else if (neigh->poc[1] == curr->poc[0] && neigh->poc[0] == curr->poc[1]) {            /* :651-671 */
            return bs_abs4(neigh->mv[1][0], curr->mv[0][0]) || bs_abs4(neigh->mv[1][1], curr->mv[0][1]) ||
                   bs_abs4(neigh->mv[0][0], curr->mv[1][0]) || bs_abs4(neigh->mv[0][1], curr->mv[1][1]);
        }
        return 1;                                                                               /* :672-674 */
    } else if (curr->pred_flag != 3 && neigh->pred_flag != 3) {                                 /* one vector each, :675-698 */
        const int la = (curr->pred_flag & 1) ? 0 : 1, lb = (neigh->pred_flag & 1) ? 0 : 1;
        if (curr->poc[la] != neigh->poc[lb])
            return 1;
        return bs_abs4(curr->mv[la][0], neigh->mv[lb][0]) || bs_abs4(curr->mv[la][1], neigh->mv[lb][1]);
    }
    return 1;                                                                                   /* :700 */
}

int oh_or_bs_derive(const OhPicParams *p, const OhBsInputs *in, uint8_t *vbs, uint8_t *hbs)
{
    const int lpu = p->log2_min_pu_size, ltu = p->log2_min_tb_size, lc = p->log2_ctb_size;
    const int mpw = p->width >> lpu, mtw = p->width >> ltu, mth = p->height >> ltu, bsw = p->width >> 2, ctbw = oh_ctb_width(p);
    memset(vbs, 0, oh_bs_size(p));
    memset(hbs, 0, oh_bs_size(p));
    for (int ty = 0; ty < mth; ty++)
        for (int tx = 0; tx < mtw; tx++) {
            const int log2 = in->call_log2[ty * mtw + tx], x0 = tx << ltu, y0 = ty << ltu;
            if (!log2 || (x0 & ((1 << log2) - 1)) || (y0 & ((1 << log2) - 1)))
                continue;                                       /* not the origin of a call */
            const int size = 1 << log2;
            const int flags = in->ctb_flags[(y0 >> lc) * ctbw + (x0 >> lc)];
            const int is_intra = in->mvf[(y0 >> lpu) * mpw + (x0 >> lpu)].pred_flag == 0;      /* :814-815 */
            if (y0 > 0 && (y0 & 7) == 0) {                                                      /* :818-855 */
                const int bd_ctby = y0 & ((1 << lc) - 1);
                const int bd_slice = (flags & OH_BSF_ACROSS_SLICES) || !(flags & OH_BSF_UP_SLICE);
                const int bd_tiles = in->loop_filter_across_tiles || !(flags & OH_BSF_UP_TILE);
                if ((bd_slice && bd_tiles) || bd_ctby)
                    for (int i = 0; i < size && x0 + i < p->width; i += 4) {
                        const OhMvField *top = &in->mvf[((y0 - 1) >> lpu) * mpw + ((x0 + i) >> lpu)];
                        const OhMvField *curr = &in->mvf[(y0 >> lpu) * mpw + ((x0 + i) >> lpu)];
                        const int cbf = in->cbf_luma[((y0 - 1) >> ltu) * mtw + ((x0 + i) >> ltu)] || in->cbf_luma[(y0 >> ltu) * mtw + ((x0 + i) >> ltu)];
                        hbs[((x0 + i) + y0 * bsw) >> 2] = (uint8_t)((curr->pred_flag == 0 || top->pred_flag == 0) ? 2 : cbf ? 1 : bs_motion(curr, top));
                    }
            }
            if (x0 > 0 && (x0 & 7) == 0) {                                                      /* :858-895 */
                const int bd_ctbx = x0 & ((1 << lc) - 1);
                const int bd_slice = (flags & OH_BSF_ACROSS_SLICES) || !(flags & OH_BSF_LEFT_SLICE);
                const int bd_tiles = in->loop_filter_across_tiles || !(flags & OH_BSF_LEFT_TILE);
                if ((bd_slice && bd_tiles) || bd_ctbx)
                    for (int i = 0; i < size && y0 + i < p->height; i += 4) {
                        const OhMvField *left = &in->mvf[((y0 + i) >> lpu) * mpw + ((x0 - 1) >> lpu)];
                        const OhMvField *curr = &in->mvf[((y0 + i) >> lpu) * mpw + (x0 >> lpu)];
                        const int cbf = in->cbf_luma[((y0 + i) >> ltu) * mtw + ((x0 - 1) >> ltu)] || in->cbf_luma[((y0 + i) >> ltu) * mtw + (x0 >> ltu)];
                        vbs[(x0 + (y0 + i) * bsw) >> 2] = (uint8_t)((curr->pred_flag == 0 || left->pred_flag == 0) ? 2 : cbf ? 1 : bs_motion(curr, left));
                    }
            }
            if (log2 > lpu && !is_intra) {                                                      /* :897-940: prediction-unit edges inside the block */
                for (int i = 0; i < size && x0 + i < p->width; i += 4) {
                    const OhMvField *top = &in->mvf[((y0 + 8 - 1) >> lpu) * mpw + ((x0 + i) >> lpu)];
                    for (int j = 8; j < size && y0 + j < p->height; j += 8) {
                        const OhMvField *curr = &in->mvf[((y0 + j) >> lpu) * mpw + ((x0 + i) >> lpu)];
                        hbs[((x0 + i) + (y0 + j) * bsw) >> 2] = (uint8_t)bs_motion(curr, top);
                        top = curr;
                    }
                }
                for (int j = 0; j < size && y0 + j < p->height; j += 4) {
                    const OhMvField *left = &in->mvf[((y0 + j) >> lpu) * mpw + ((x0 + 8 - 1) >> lpu)];
                    for (int i = 8; i < size && x0 + i < p->width; i += 8) {
                        const OhMvField *curr = &in->mvf[((y0 + j) >> lpu) * mpw + ((x0 + i) >> lpu)];
                        vbs[((x0 + i) + (y0 + j) * bsw) >> 2] = (uint8_t)bs_motion(curr, left);
                        left = curr;
                    }
                }
            }
        }
    return 0;
}
