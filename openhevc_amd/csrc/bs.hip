/*
 * bs.hip — boundary strengths from the motion field, upstream of pass 4
 * (gfx950; overview of the passes: kernels.hip; bit-exactness: tests/test_bs_derive.py, tests/test_gpu_parity.py)
 */
#include "kernels_common.h"

/* =========================================================================================
 * ff_hevc_deblocking_boundary_strengths (hevc_filter.c:805-941) turned inside out: the reference walks the blocks it is
 * called for and writes the strengths of their outer edges and of the prediction-unit edges inside them; every 4-sample
 * edge segment is written by exactly one call, so one lane per 4x4 cell of the picture can ask "which call covers me, and
 * am I on its edge or inside it" and produce both grids in one HBM-bound sweep over the maps (OhBsInputs):
 *   cell (x, y), block = call_log2 at the cell, origin = (x, y) rounded down to the block size
 *   y % 8 == 0, y > 0:  on the block's top edge    -> the slice / tile / CTB conditions, then intra 2 / cbf 1 / motion
 *                       inside the block           -> motion only, if the block is larger than a min PU and not intra;
 *                                                     the neighbour is the row the reference's running `top` pointer
 *                                                     holds: y0 + 7 for the first inner edge, 8 rows up afterwards
 *   same for x and the vertical grid.  Cells no call covers keep 0 (the reference zeroes both grids per picture).
 * boundary_strength() is the TEST_MV_POC build: whole-struct equality first (:600), then the POC / vector rules (:603-700).
 * ======================================================================================= */
struct MvF { int mv01, mv23; int poc0, poc1; unsigned pred_flag; unsigned tail; };      /* OhMvField as six dwords */

static __device__ __forceinline__ MvF load_mvf(const GLOBAL OhMvField *p)
{
    const GLOBAL uint2v *q = (const GLOBAL uint2v *)p;                                   /* 24 bytes, 8-byte aligned */
    const uint2v a = q[0], b = q[1], c = q[2];
    return MvF{ (int)a[0], (int)a[1], (int)b[0], (int)b[1], c[0], c[1] };
}
static __device__ __forceinline__ bool far4(int a, int b) { return abs(a - b) >= 4; }
/* both components of vector la of `a` against vector lb of `b` (packed x | y << 16) */
static __device__ __forceinline__ bool mv_far(int a, int b)
{
    return far4((int)(int16_t)a, (int)(int16_t)b) || far4(a >> 16, b >> 16);
}

static __device__ __forceinline__ int bs_motion(const MvF &c, const MvF &n)
{
    if (c.mv01 == n.mv01 && c.mv23 == n.mv23 && c.poc0 == n.poc0 && c.poc1 == n.poc1 && c.pred_flag == n.pred_flag && c.tail == n.tail)
        return 0;
    if (c.pred_flag == 3 && n.pred_flag == 3) {
        if (c.poc0 == n.poc0 && c.poc0 == c.poc1 && n.poc0 == n.poc1)
            return (mv_far(n.mv01, c.mv01) || mv_far(n.mv23, c.mv23)) && (mv_far(n.mv23, c.mv01) || mv_far(n.mv01, c.mv23));
        if (n.poc0 == c.poc0 && n.poc1 == c.poc1)
            return mv_far(n.mv01, c.mv01) || mv_far(n.mv23, c.mv23);
        if (n.poc1 == c.poc0 && n.poc0 == c.poc1)
            return mv_far(n.mv23, c.mv01) || mv_far(n.mv01, c.mv23);
        return 1;
    }
    if (c.pred_flag != 3 && n.pred_flag != 3) {
        const bool c0 = c.pred_flag & 1, n0 = n.pred_flag & 1;
        if ((c0 ? c.poc0 : c.poc1) != (n0 ? n.poc0 : n.poc1))
            return 1;
        return mv_far(c0 ? c.mv01 : c.mv23, n0 ? n.mv01 : n.mv23);
    }
    return 1;
}

struct BsArgs {
    const OhMvField *mvf; const uint8_t *cbf, *call_log2, *ctb_flags;
    uint8_t *vbs, *hbs;
    int width, height, lpu, ltu, lc, across_tiles;
};

__global__ __launch_bounds__(256) void bs_kernel(const BsArgs a)
{
    const int cx = blockIdx.x * 64 + (threadIdx.x & 63), cy = blockIdx.y * 4 + (threadIdx.x >> 6);      /* 4x4 cell */
    const int x = cx << 2, y = cy << 2;
    if (x >= a.width || y >= a.height)
        return;
    const int mpw = a.width >> a.lpu, mtw = a.width >> a.ltu, bsw = a.width >> 2, ctbw = (a.width + (1 << a.lc) - 1) >> a.lc;
    const GLOBAL OhMvField *__restrict__ mvf = G_CONST(OhMvField, a.mvf);
    const GLOBAL uint8_t *__restrict__ cbf = G_CONST(uint8_t, a.cbf);
    const int log2 = G_CONST(uint8_t, a.call_log2)[(y >> a.ltu) * mtw + (x >> a.ltu)];
    int v = 0, h = 0;
    if (log2) {
        const int mask = (1 << log2) - 1, x0 = x & ~mask, y0 = y & ~mask;
        const int flags = G_CONST(uint8_t, a.ctb_flags)[(y0 >> a.lc) * ctbw + (x0 >> a.lc)];
        const MvF curr = load_mvf(mvf + (y >> a.lpu) * mpw + (x >> a.lpu));
        const bool inner = log2 > a.lpu && mvf[(y0 >> a.lpu) * mpw + (x0 >> a.lpu)].pred_flag != 0;
        const int my_cbf = cbf[(y >> a.ltu) * mtw + (x >> a.ltu)];
        if (y > 0 && (y & 7) == 0) {
            if (y == y0) {
                const bool bd_slice = (flags & OH_BSF_ACROSS_SLICES) || !(flags & OH_BSF_UP_SLICE);
                const bool bd_tiles = a.across_tiles || !(flags & OH_BSF_UP_TILE);
                if ((bd_slice && bd_tiles) || (y0 & ((1 << a.lc) - 1))) {
                    const MvF top = load_mvf(mvf + ((y - 1) >> a.lpu) * mpw + (x >> a.lpu));
                    h = (curr.pred_flag == 0 || top.pred_flag == 0) ? 2 : (my_cbf || cbf[((y - 1) >> a.ltu) * mtw + (x >> a.ltu)]) ? 1 : bs_motion(curr, top);
                }
            } else if (inner) {
                const int ty = y - y0 == 8 ? y0 + 7 : y - 8;
                h = bs_motion(curr, load_mvf(mvf + (ty >> a.lpu) * mpw + (x >> a.lpu)));
            }
        }
        if (x > 0 && (x & 7) == 0) {
            if (x == x0) {
                const bool bd_slice = (flags & OH_BSF_ACROSS_SLICES) || !(flags & OH_BSF_LEFT_SLICE);
                const bool bd_tiles = a.across_tiles || !(flags & OH_BSF_LEFT_TILE);
                if ((bd_slice && bd_tiles) || (x0 & ((1 << a.lc) - 1))) {
                    const MvF left = load_mvf(mvf + (y >> a.lpu) * mpw + ((x - 1) >> a.lpu));
                    v = (curr.pred_flag == 0 || left.pred_flag == 0) ? 2 : (my_cbf || cbf[(y >> a.ltu) * mtw + ((x - 1) >> a.ltu)]) ? 1 : bs_motion(curr, left);
                }
            } else if (inner) {
                const int tx = x - x0 == 8 ? x0 + 7 : x - 8;
                v = bs_motion(curr, load_mvf(mvf + (y >> a.lpu) * mpw + (tx >> a.lpu)));
            }
        }
    }
    /* four strengths to the byte (dev_frame.h: vbs / hbs), the grids were cleared before the launch: only non-zero ones are written */
    const uint32_t i = (uint32_t)(x + y * bsw) >> 2;
    if (h) atomicOr((uint32_t *)a.hbs + (i >> 4), (uint32_t)h << ((i & 15) * 2));
    if (v) atomicOr((uint32_t *)a.vbs + (i >> 4), (uint32_t)v << ((i & 15) * 2));
}

/* both grids of one picture; vbs / hbs: oh_bs_size() strengths each, four to the byte, cleared by the caller (word-aligned size) */
extern "C" void ohk_bs_derive(const OhPicParams *p, const void *mvf, const void *cbf, const void *call_log2, const void *ctb_flags,
                              int across_tiles, void *vbs, void *hbs, hipStream_t st)
{
    BsArgs a = { (const OhMvField *)mvf, (const uint8_t *)cbf, (const uint8_t *)call_log2, (const uint8_t *)ctb_flags, (uint8_t *)vbs, (uint8_t *)hbs,
                 p->width, p->height, p->log2_min_pu_size, p->log2_min_tb_size, p->log2_ctb_size, across_tiles };
    dim3 grid(((p->width >> 2) + 63) / 64, ((p->height >> 2) + 3) / 4);
    hipLaunchKernelGGL(bs_kernel, grid, dim3(256), 0, st, a);
}
