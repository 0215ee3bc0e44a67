/*
 * upsample.hip — SHVC inter-layer up-sampling of a whole base-layer picture
 * (gfx950; overview of the passes: kernels.hip; bit-exactness: tests/test_gpu_parity.py)
 */
#include "kernels_common.h"

/* =========================================================================================
 * SHVC inter-layer up-sampling: upsample_base_layer_frame, hevcdsp_template.c:2164-2438 —
 * 16-phase separable resampling of a base-layer plane into the enhancement layer's geometry, 8-tap luma /
 * 4-tap chroma (tables hevcdsp.c:948-986).  One launch per plane, one workgroup per TILE of the enhancement-layer
 * plane (a CTB: 64 x 64 luma, 32 x 32 chroma — the unit the reference's on-demand path up-samples, ff_upsample_block
 * hevc_filter.c:1370-1426 / is_upsampled[]): the base-layer window the tile's taps reach is staged in LDS once
 * (coordinate clamps = the reference's edge buffers), the horizontal pass writes the int16 rows (no rounding, as the
 * reference's short buffer) into LDS, the vertical pass reads them there, rounds (>> 12), clips and stores — the
 * intermediate plane of the reference's two loops never exists in HBM.  Tiles come from a list (the CTBs a picture's
 * inter-layer prediction touches: oh_pic_upsample_ctbs) or are all tiles of the plane (oh_pic_upsample).
 * ======================================================================================= */
__constant__ int8_t c_up_luma[16][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 63, 4, -2, 1, 0 }, { -1, 2, -5, 62, 8, -3, 1, 0 }, { -1, 3, -8, 60, 13, -4, 1, 0 },
    { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 52, 26, -8, 3, -1 }, { -1, 3, -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 },
    { -1, 4, -11, 40, 40, -11, 4, -1 }, { -1, 4, -10, 34, 45, -11, 4, -1 }, { -1, 4, -10, 31, 47, -9, 3, -1 }, { -1, 3, -8, 26, 52, -11, 4, -1 },
    { 0, 1, -5, 17, 58, -10, 4, -1 }, { 0, 1, -4, 13, 60, -8, 3, -1 }, { 0, 1, -3, 8, 62, -5, 2, -1 }, { 0, 1, -2, 4, 63, -3, 1, 0 } };
__constant__ int8_t c_up_chroma[16][4] = {
    { 0, 64, 0, 0 }, { -2, 62, 4, 0 }, { -2, 58, 10, -2 }, { -4, 56, 14, -2 }, { -4, 54, 16, -2 }, { -6, 52, 20, -2 }, { -6, 46, 28, -4 }, { -4, 42, 30, -4 },
    { -4, 36, 36, -4 }, { -4, 30, 42, -4 }, { -4, 28, 46, -6 }, { -2, 20, 52, -6 }, { -2, 16, 54, -4 }, { -2, 14, 56, -4 }, { -2, 10, 58, -2 }, { 0, 4, 62, -2 } };

#define UP_TILE 64                    /* tile edge (luma CTB); scale <= 1 (EL >= BL): the window is at most UP_TILE + 8 */
#define UP_WIN  (UP_TILE + 8)

template <int TAPS>
__global__ __launch_bounds__(256) void upsample_tile_kernel(const OhUpPlane a, const int tw, const int th, const int tiles_x, const uint32_t *list)
{
    __shared__ uint8_t srcL[UP_WIN][UP_WIN + 8];
    __shared__ int16_t tmpL[UP_WIN][UP_TILE];
    const int tile = list ? (int)list[blockIdx.x] : (int)blockIdx.x;
    const int x0 = (tile % tiles_x) * tw, y0 = (tile / tiles_x) * th;
    if (x0 >= a.w_el || y0 >= a.h_el)
        return;
    const int w = min(tw, a.w_el - x0), h = min(th, a.h_el - y0), tid = threadIdx.x;
    constexpr int B = TAPS / 2 - 1;
    /* the intermediate columns this tile's outputs read: the source column only advances inside the window (:1925) */
    const int c_first = clip3(x0, a.left, a.right_end_v - 1) - a.left, c_last = clip3(x0 + w - 1, a.left, a.right_end_v - 1) - a.left;
    const int nc = c_last - c_first + 1;
    auto xpos = [&](int c) { return ((clip3(c, a.left, a.right_end_h) - a.left) * a.scale_x + a.add_x) >> 12; };     /* 1/16 sample */
    auto ypos = [&](int j) { return (((clip3(j, a.top, a.bottom_end - 1) - a.top) * a.scale_y + a.add_y) >> 12) - a.y_bias; };
    const int p_first = (xpos(c_first) >> 4) - B, p_last = (xpos(c_last) >> 4) - B + TAPS - 1;
    const int r_first = (ypos(y0) >> 4) - B, r_last = (ypos(y0 + h - 1) >> 4) - B + TAPS - 1;
    const int sw = min(p_last - p_first + 1, UP_WIN + 8), sh = min(r_last - r_first + 1, UP_WIN);   /* the host checked scale <= 1 */
    const GLOBAL uint8_t *__restrict__ src = G_CONST(uint8_t, a.src);
    for (int e = tid; e < sw * sh; e += 256) {
        const int r = e / sw, c = e - r * sw;
        srcL[r][c] = src[(size_t)clip3(r_first + r, 0, a.h_bl - 1) * a.sstride + clip3(p_first + c, 0, a.w_bl - 1)];
    }
    __syncthreads();
    for (int e = tid; e < nc * sh; e += 256) {
        const int r = e / nc, c = e - r * nc;
        const int r16 = xpos(c_first + c), phase = r16 & 15, o = (r16 >> 4) - B - p_first;
        int s = 0;
#pragma unroll
        for (int k = 0; k < TAPS; k++)
            s += (TAPS == 8 ? c_up_luma[phase][k] : c_up_chroma[phase][k]) * srcL[r][o + k];
        tmpL[r][c] = (int16_t)s;
    }
    __syncthreads();
    GLOBAL uint8_t *__restrict__ dst = G_MUT(uint8_t, a.dst);
    for (int e = tid; e < w * h; e += 256) {
        const int j = e / w, i = e - j * w;
        const int r16 = ypos(y0 + j), phase = r16 & 15, o = (r16 >> 4) - B - r_first;
        const int col = clip3(x0 + i, a.left, a.right_end_v - 1) - a.left - c_first;
        int s = 0;
#pragma unroll
        for (int k = 0; k < TAPS; k++)
            s += (TAPS == 8 ? c_up_luma[phase][k] : c_up_chroma[phase][k]) * tmpL[o + k][col];
        dst[(size_t)(y0 + j) * a.dstride + x0 + i] = (uint8_t)clip3((s + 2048) >> 12, 0, 255);
    }
}

/* one plane: tiles tw x th, either those of `list` (n_list tile indices, row-major over ceil(w_el / tw) columns; device memory) or all */
extern "C" void ohk_upsample_plane(const OhUpPlane *a, int taps, int tw, int th, const uint32_t *list, int n_list, hipStream_t st)
{
    const int tiles_x = (a->w_el + tw - 1) / tw, tiles_y = (a->h_el + th - 1) / th;
    const int n = list ? n_list : tiles_x * tiles_y;
    if (n <= 0) return;
    if (taps == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(upsample_tile_kernel<8>), dim3(n), dim3(256), 0, st, *a, tw, th, tiles_x, list);
    else           hipLaunchKernelGGL(HIP_KERNEL_NAME(upsample_tile_kernel<4>), dim3(n), dim3(256), 0, st, *a, tw, th, tiles_x, list);
}
