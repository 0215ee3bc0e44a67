/*
 * upsample.hip — SHVC inter-layer up-sampling of a whole base-layer picture
 * (gfx950; overview of the passes: kernels.hip; bit-exactness: tests/test_gpu_parity.py)
 */
#include "kernels_common.h"

/* =========================================================================================
 * SHVC inter-layer up-sampling: upsample_base_layer_frame, hevcdsp_template.c:2164-2438 —
 * 16-phase separable resampling of a base-layer plane into the enhancement layer's geometry, 8-tap luma /
 * 4-tap chroma (tables hevcdsp.c:948-986).  Two launches per plane like the reference's two loops: the
 * horizontal pass writes int16 rows tmp[h_bl][w_el] (no rounding, as the reference's short buffer), the
 * vertical pass rounds (>> 12) and clips.  The edge buffers of the reference are coordinate clamps.
 * One thread per output sample; neighbours share their taps' inputs through the caches (HBM-bound pass).
 * ======================================================================================= */
__constant__ int8_t c_up_luma[16][8] = {
    { 0, 0, 0, 64, 0, 0, 0, 0 }, { 0, 1, -3, 63, 4, -2, 1, 0 }, { -1, 2, -5, 62, 8, -3, 1, 0 }, { -1, 3, -8, 60, 13, -4, 1, 0 },
    { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 52, 26, -8, 3, -1 }, { -1, 3, -9, 47, 31, -10, 4, -1 }, { -1, 4, -11, 45, 34, -10, 4, -1 },
    { -1, 4, -11, 40, 40, -11, 4, -1 }, { -1, 4, -10, 34, 45, -11, 4, -1 }, { -1, 4, -10, 31, 47, -9, 3, -1 }, { -1, 3, -8, 26, 52, -11, 4, -1 },
    { 0, 1, -5, 17, 58, -10, 4, -1 }, { 0, 1, -4, 13, 60, -8, 3, -1 }, { 0, 1, -3, 8, 62, -5, 2, -1 }, { 0, 1, -2, 4, 63, -3, 1, 0 } };
__constant__ int8_t c_up_chroma[16][4] = {
    { 0, 64, 0, 0 }, { -2, 62, 4, 0 }, { -2, 58, 10, -2 }, { -4, 56, 14, -2 }, { -4, 54, 16, -2 }, { -6, 52, 20, -2 }, { -6, 46, 28, -4 }, { -4, 42, 30, -4 },
    { -4, 36, 36, -4 }, { -4, 30, 42, -4 }, { -4, 28, 46, -6 }, { -2, 20, 52, -6 }, { -2, 16, 54, -4 }, { -2, 14, 56, -4 }, { -2, 10, 58, -2 }, { 0, 4, 62, -2 } };

template <int TAPS>
__global__ __launch_bounds__(256) void upsample_h_kernel(const OhUpPlane a)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= a.w_el)
        return;
    const int x = clip3(i, a.left, a.right_end_h);
    const int r16 = ((x - a.left) * a.scale_x + a.add_x) >> 12, phase = r16 & 15, pos = (r16 >> 4) - (TAPS / 2 - 1);
    const GLOBAL uint8_t *__restrict__ row = G_CONST(uint8_t, a.src) + (size_t)j * a.sstride;
    int s = 0;
#pragma unroll
    for (int k = 0; k < TAPS; k++)
        s += (TAPS == 8 ? c_up_luma[phase][k] : c_up_chroma[phase][k]) * row[clip3(pos + k, 0, a.w_bl - 1)];
    G_MUT(int16_t, a.tmp)[(size_t)j * a.w_el + i] = (int16_t)s;
}

template <int TAPS>
__global__ __launch_bounds__(256) void upsample_v_kernel(const OhUpPlane a)
{
    const int i = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
    if (i >= a.w_el)
        return;
    const int y = clip3(j, a.top, a.bottom_end - 1);
    const int r16 = (((y - a.top) * a.scale_y + a.add_y) >> 12) - a.y_bias, phase = r16 & 15, pos = (r16 >> 4) - (TAPS / 2 - 1);
    const int col = clip3(i, a.left, a.right_end_v - 1) - a.left;      /* the reference's source column only advances inside the window */
    const GLOBAL int16_t *__restrict__ t = G_CONST(int16_t, a.tmp) + col;
    int s = 0;
#pragma unroll
    for (int k = 0; k < TAPS; k++)
        s += (TAPS == 8 ? c_up_luma[phase][k] : c_up_chroma[phase][k]) * t[(size_t)clip3(pos + k, 0, a.h_bl - 1) * a.w_el];
    G_MUT(uint8_t, a.dst)[(size_t)j * a.dstride + i] = (uint8_t)clip3((s + 2048) >> 12, 0, 255);
}

extern "C" void ohk_upsample_plane(const OhUpPlane *a, int taps, hipStream_t st)
{
    dim3 gh((a->w_el + 255) / 256, a->h_bl), gv((a->w_el + 255) / 256, a->h_el);
    if (taps == 8) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(upsample_h_kernel<8>), gh, dim3(256), 0, st, *a);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(upsample_v_kernel<8>), gv, dim3(256), 0, st, *a);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(upsample_h_kernel<4>), gh, dim3(256), 0, st, *a);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(upsample_v_kernel<4>), gv, dim3(256), 0, st, *a);
    }
}
