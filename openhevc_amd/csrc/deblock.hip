/*
 * deblock.hip — pass 4: deblocking, vertical edges then horizontal edges, in place
 * (gfx950; overview of the passes: kernels.hip; bit-exactness: tests/test_gpu_parity.py)
 */
#include "kernels_common.h"

/* tc / beta / chroma QP tables (H.265 facts; same numbers as hevc_filter.c:50-60) */
__constant__ uint8_t c_tc[54] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                  5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
__constant__ uint8_t c_beta[52] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24,
                                    26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };
__constant__ uint8_t c_qpc[14] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };

/* =========================================================================================
 * pass 4: deblocking — hevcdsp_template.c:1629-1757 with the per-edge parameter rules of
 * deblocking_filter_CTB (hevc_filter.c:345-581).  One lane per 4-line segment.  All vertical
 * edges of the picture, then (second launch) all horizontal edges; both in place: segments of one
 * direction never touch each other's samples.
 * ======================================================================================= */
static __device__ __forceinline__ int get_qpy(const DevFrame *f, int x, int y)
{
    int l = f->pp.log2_min_cb_size;
    return f->qp[(x >> l) + (y >> l) * (f->pp.width >> l)];
}
static __device__ __forceinline__ int get_pcm(const DevFrame *f, int x, int y)
{
    int l = f->pp.log2_min_pu_size;
    int mpw = f->pp.width >> l, mph = f->pp.height >> l;
    if (x < 0 || y < 0 || (x >> l) >= mpw || (y >> l) >= mph)
        return 2;
    return f->is_pcm ? f->is_pcm[(y >> l) * mpw + (x >> l)] : 0;
}

template <typename PX, int HORIZ>       /* HORIZ = 1: horizontal edges (filter across y) */
__global__ __launch_bounds__(256) void deblock_luma_kernel(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.z];
    const OhPicParams &pp = f->pp;
    const int W = pp.width, H = pp.height, bd = pp.bit_depth;
    /* x index runs fastest in both directions so that a wave touches neighbouring addresses */
    const int gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y;
    int x, y;
    if (!HORIZ) { x = 8 * (gx + 1); y = 4 * gy; } else { x = 4 * gx; y = 8 * (gy + 1); }
    if (x >= W || y >= H)
        return;
    const int bsw = W >> 2;
    const int bsi = (x + y * bsw) >> 2, bs = ((HORIZ ? f->hbs : f->vbs)[bsi >> 2] >> ((bsi & 3) * 2)) & 3;      /* four strengths to the byte */
    if (!bs)
        return;
    const int lc = pp.log2_ctb_size, ctbw = (W + (1 << lc) - 1) >> lc;
    int qp, beta_off, tc_off;
    if (!HORIZ) {
        int y8 = y & ~7;
        qp = (get_qpy(f, x - 1, y8) + get_qpy(f, x, y8) + 1) >> 1;
        OhDeblockCtb d = f->db[(y >> lc) * ctbw + (x >> lc)];
        beta_off = d.beta_offset; tc_off = d.tc_offset;
    } else {
        int x8 = x & ~7;
        qp = (get_qpy(f, x8, y - 1) + get_qpy(f, x8, y) + 1) >> 1;
        int pcx = min((x8 + 8) >> lc, ctbw - 1);          /* hevc_filter.c:481-520 */
        tc_off = f->db[(y >> lc) * ctbw + pcx].tc_offset;
        beta_off = f->db[(y >> lc) * ctbw + (x8 >> lc)].beta_offset;
    }
    const int beta = c_beta[clip3(qp + beta_off, 0, 51)] << (bd - 8);
    const int tc = c_tc[clip3(qp + 2 * (bs - 1) + (tc_off >> 1 << 1), 0, 53)] << (bd - 8);
    int no_p = 0, no_q = 0;
    if (pp.pcm_loop_filter_disable || pp.transquant_bypass_enable) {
        no_p = HORIZ ? get_pcm(f, x, y - 1) : get_pcm(f, x - 1, y);
        no_q = get_pcm(f, x, y);
    }
    const int stride = f->cur.stride[0];
    GLOBAL PX *pix = G_MUT(PX, f->cur.p[0]) + (size_t)y * stride + x;

    /* [line][distance from the edge]; vertical edges: a line is one row (8 contiguous samples),
     * horizontal edges: a line is one column, the lane's 4 columns are contiguous in every row */
    int P[4][4], Q[4][4];
    if (!HORIZ) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            int t[4];
            load4<PX>(pix + (size_t)d * stride - 4, t);
            P[d][3] = t[0]; P[d][2] = t[1]; P[d][1] = t[2]; P[d][0] = t[3];
            load4<PX>(pix + (size_t)d * stride, Q[d]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int t[4];
            load4<PX>(pix - (size_t)(k + 1) * stride, t);
            P[0][k] = t[0]; P[1][k] = t[1]; P[2][k] = t[2]; P[3][k] = t[3];
            load4<PX>(pix + (size_t)k * stride, t);
            Q[0][k] = t[0]; Q[1][k] = t[1]; Q[2][k] = t[2]; Q[3][k] = t[3];
        }
    }
    int NP[4][3], NQ[4][3];                                /* filtered samples, distance 0..2 */
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
        for (int k = 0; k < 3; k++) { NP[d][k] = P[d][k]; NQ[d][k] = Q[d][k]; }
    const int dp0 = abs(P[0][2] - 2 * P[0][1] + P[0][0]), dq0 = abs(Q[0][2] - 2 * Q[0][1] + Q[0][0]);
    const int dp3 = abs(P[3][2] - 2 * P[3][1] + P[3][0]), dq3 = abs(Q[3][2] - 2 * Q[3][1] + Q[3][0]);
    const int d0 = dp0 + dq0, d3 = dp3 + dq3;
    if (d0 + d3 >= beta)
        return;
    const int beta3 = beta >> 3, beta2 = beta >> 2, tc25 = (tc * 5 + 1) >> 1;
    const bool strong =
        abs(P[0][3] - P[0][0]) + abs(Q[0][3] - Q[0][0]) < beta3 && abs(P[0][0] - Q[0][0]) < tc25 &&
        abs(P[3][3] - P[3][0]) + abs(Q[3][3] - Q[3][0]) < beta3 && abs(P[3][0] - Q[3][0]) < tc25 &&
        (d0 << 1) < beta2 && (d3 << 1) < beta2;
    if (strong) {
        const int tc2 = tc << 1;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            int p3 = P[d][3], p2 = P[d][2], p1 = P[d][1], p0 = P[d][0];
            int q0 = Q[d][0], q1 = Q[d][1], q2 = Q[d][2], q3 = Q[d][3];
            if (!no_p) {
                NP[d][0] = p0 + clip3(((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0, -tc2, tc2);
                NP[d][1] = p1 + clip3(((p2 + p1 + p0 + q0 + 2) >> 2) - p1, -tc2, tc2);
                NP[d][2] = p2 + clip3(((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2, -tc2, tc2);
            }
            if (!no_q) {
                NQ[d][0] = q0 + clip3(((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0, -tc2, tc2);
                NQ[d][1] = q1 + clip3(((p0 + q0 + q1 + q2 + 2) >> 2) - q1, -tc2, tc2);
                NQ[d][2] = q2 + clip3(((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2, -tc2, tc2);
            }
        }
    } else {
        const int side = (beta + (beta >> 1)) >> 3, tc_2 = tc >> 1;
        const bool nd_p = dp0 + dp3 < side, nd_q = dq0 + dq3 < side;
#pragma unroll
        for (int d = 0; d < 4; d++) {
            int p2 = P[d][2], p1 = P[d][1], p0 = P[d][0], q0 = Q[d][0], q1 = Q[d][1], q2 = Q[d][2];
            int delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4;
            if (abs(delta) >= 10 * tc)
                continue;
            delta = clip3(delta, -tc, tc);
            if (!no_p) NP[d][0] = clip_px(p0 + delta, bd);
            if (!no_q) NQ[d][0] = clip_px(q0 - delta, bd);
            if (!no_p && nd_p) NP[d][1] = clip_px(p1 + clip3((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1, -tc_2, tc_2), bd);
            if (!no_q && nd_q) NQ[d][1] = clip_px(q1 + clip3((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1, -tc_2, tc_2), bd);
        }
    }
    /* write back whole 4-sample groups; the untouched outer samples (distance 3) are rewritten with
     * their own values, which no other segment of this pass modifies */
    if (!HORIZ) {
#pragma unroll
        for (int d = 0; d < 4; d++) {
            store4<PX>(pix + (size_t)d * stride - 4, P[d][3], NP[d][2], NP[d][1], NP[d][0]);
            store4<PX>(pix + (size_t)d * stride, NQ[d][0], NQ[d][1], NQ[d][2], Q[d][3]);
        }
    } else {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            store4<PX>(pix - (size_t)(k + 1) * stride, NP[0][k], NP[1][k], NP[2][k], NP[3][k]);
            store4<PX>(pix + (size_t)k * stride, NQ[0][k], NQ[1][k], NQ[2][k], NQ[3][k]);
        }
    }
}

static __device__ __forceinline__ int chroma_tc(const DevFrame *f, int qp_y, int c, int tc_off)   /* hevc_filter.c:62-89 */
{
    int qp_i = clip3(qp_y + (c == 1 ? f->pp.cb_qp_offset : f->pp.cr_qp_offset), 0, 57);
    int qp;
    if (f->pp.chroma_format_idc == 1) qp = qp_i < 30 ? qp_i : (qp_i > 43 ? qp_i - 6 : c_qpc[qp_i - 30]);
    else                              qp = min(qp_i, 51);
    return c_tc[clip3(qp + 2 + tc_off, 0, 53)];
}

template <typename PX, int HORIZ>
__global__ __launch_bounds__(256) void deblock_chroma_kernel(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.z >> 1];
    const OhPicParams &pp = f->pp;
    const int W = pp.width, H = pp.height, bd = pp.bit_depth;
    const int hs = hsh(pp, 1), vs = vsh(pp, 1), hh = 1 << hs, vv = 1 << vs;
    const int gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y, c = 1 + (blockIdx.z & 1);
    int x, y;                                               /* luma coordinates of the segment */
    if (!HORIZ) { x = 8 * hh * (gx + 1); y = 4 * vv * gy; } else { x = 4 * hh * gx; y = 8 * vv * (gy + 1); }
    if (x >= W || y >= H)
        return;
    const int bsw = W >> 2;
    const int bsi = (x + y * bsw) >> 2, bs = ((HORIZ ? f->hbs : f->vbs)[bsi >> 2] >> ((bsi & 3) * 2)) & 3;      /* four strengths to the byte */
    const int lc = pp.log2_ctb_size, ctbw = (W + (1 << lc) - 1) >> lc;
    if (HORIZ && f->sao_stale && !(gx & 1)) {
        /* 16x16 CTBs, subsampled chroma: deblocking_filter_CTB handles the horizontal chroma edges of a CTB's columns in the call
         * for the NEXT CTB (hevc_filter.c:526-530), after sao_filter_CTB of the previous one has copied this column
         * (ff_hevc_hls_filter, :1027-1052): keep what that SAO call saw — the unfiltered p0 / q0 of the CTB's first column */
        const int st = f->cur.stride[c], xc = x >> hs, yc = y >> vs;
        const GLOBAL PX *q = G_CONST(PX, f->cur.p[c]) + (size_t)yc * st + xc;
        GLOBAL uint16_t *o = G_MUT(uint16_t, f->sao_stale) + oh_sao_stale_index(&pp, c, yc >> 3, xc >> 3);
        o[0] = q[-st]; o[1] = q[0];
    }
    if (bs != 2)
        return;
    int qp, tc_off;
    if (!HORIZ) {
        qp = (get_qpy(f, x - 1, y) + get_qpy(f, x, y) + 1) >> 1;
        tc_off = f->db[(y >> lc) * ctbw + (x >> lc)].tc_offset;
    } else {
        qp = (get_qpy(f, x, y - 1) + get_qpy(f, x, y) + 1) >> 1;
        int x16 = x & ~(8 * hh - 1);                        /* start of the 8-sample chroma edge */
        int pcx = min((x16 + 8 * hh) >> lc, ctbw - 1);      /* hevc_filter.c:523-580 */
        tc_off = x == x16 ? f->db[(y >> lc) * ctbw + (x16 >> lc)].tc_offset : f->db[(y >> lc) * ctbw + pcx].tc_offset;
    }
    const int tc = chroma_tc(f, qp, c, tc_off) << (bd - 8);
    if (tc <= 0)
        return;
    int no_p = 0, no_q = 0;
    if (pp.pcm_loop_filter_disable || pp.transquant_bypass_enable) {
        no_p = HORIZ ? get_pcm(f, x, y - 1) : get_pcm(f, x - 1, y);
        no_q = get_pcm(f, x, y);
    }
    const int stride = f->cur.stride[c];
    GLOBAL PX *pix = G_MUT(PX, f->cur.p[c]) + (size_t)(y >> vs) * stride + (x >> hs);
    if (!HORIZ) {                                          /* a line is one row: p3..p0 | q0..q3 contiguous */
#pragma unroll
        for (int d = 0; d < 4; d++) {
            int p[4], q[4];
            load4<PX>(pix + (size_t)d * stride - 4, p);
            load4<PX>(pix + (size_t)d * stride, q);
            int delta = clip3((((q[0] - p[3]) * 4) + p[2] - q[1] + 4) >> 3, -tc, tc);
            if (!no_p) p[3] = clip_px(p[3] + delta, bd);
            if (!no_q) q[0] = clip_px(q[0] - delta, bd);
            store4<PX>(pix + (size_t)d * stride - 4, p[0], p[1], p[2], p[3]);
            store4<PX>(pix + (size_t)d * stride, q[0], q[1], q[2], q[3]);
        }
    } else {                                               /* the lane's 4 columns are contiguous in every row */
        int p1[4], p0[4], q0[4], q1[4];
        load4<PX>(pix - 2 * (size_t)stride, p1);
        load4<PX>(pix - (size_t)stride, p0);
        load4<PX>(pix, q0);
        load4<PX>(pix + (size_t)stride, q1);
#pragma unroll
        for (int d = 0; d < 4; d++) {
            int delta = clip3((((q0[d] - p0[d]) * 4) + p1[d] - q1[d] + 4) >> 3, -tc, tc);
            if (!no_p) p0[d] = clip_px(p0[d] + delta, bd);
            if (!no_q) q0[d] = clip_px(q0[d] - delta, bd);
        }
        store4<PX>(pix - (size_t)stride, p0[0], p0[1], p0[2], p0[3]);
        store4<PX>(pix, q0[0], q0[1], q0[2], q0[3]);
    }
}

/* =========================================================================================
 * launcher
 * ======================================================================================= */
extern "C" void ohk_deblock(const OhBatch *B, int n, const OhPicParams *p, int horiz, hipStream_t st)
{
    const int W = p->width, H = p->height;
    const int hs = p->chroma_format_idc == 1 || p->chroma_format_idc == 2, vs = p->chroma_format_idc == 1;
    if (!horiz) {
        dim3 g(W / 8 / 256 + 1, H / 4, n), gc(W / (8 << hs) / 256 + 1, (H + (4 << vs) - 1) / (4 << vs), 2 * n);
        if (p->bit_depth == 8) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_luma_kernel<uint8_t, 0>), g, dim3(256), 0, st, *B);
            if (p->chroma_format_idc) hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_chroma_kernel<uint8_t, 0>), gc, dim3(256), 0, st, *B);
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_luma_kernel<uint16_t, 0>), g, dim3(256), 0, st, *B);
            if (p->chroma_format_idc) hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_chroma_kernel<uint16_t, 0>), gc, dim3(256), 0, st, *B);
        }
    } else {
        dim3 g(W / 4 / 256 + 1, H / 8, n), gc(W / (4 << hs) / 256 + 1, (H + (8 << vs) - 1) / (8 << vs), 2 * n);
        if (p->bit_depth == 8) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_luma_kernel<uint8_t, 1>), g, dim3(256), 0, st, *B);
            if (p->chroma_format_idc) hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_chroma_kernel<uint8_t, 1>), gc, dim3(256), 0, st, *B);
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_luma_kernel<uint16_t, 1>), g, dim3(256), 0, st, *B);
            if (p->chroma_format_idc) hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_chroma_kernel<uint16_t, 1>), gc, dim3(256), 0, st, *B);
        }
    }
}
