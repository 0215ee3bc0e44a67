/*
 * residual.hip — pass 2: residual (dequant, IDCT / DST / skip / bypass / PCM, add) and cross-component prediction
 * (gfx950; overview of the passes: kernels.hip; bit-exactness: tests/test_gpu_parity.py)
 */
#include "kernels_common.h"

__constant__ uint8_t c_level_scale[6] = { 40, 45, 51, 57, 64, 72 };        /* hevc_cabac.c:1417 */

/* =========================================================================================
 * pass 2: residual — hevcdsp_template.c:114-316 dispatched as hevc_cabac.c:1868-1949;
 * inter blocks are added to the prediction here (transform_add, :45-111), intra blocks leave
 * their residual in f->res for pass 3.
 *
 * The engine sorts the transform blocks by size (DevFrame.tu_first / tu_cnt) and one launch per size runs
 * residual_kernel<PX, LOG2>: a lane owns groups of 4 consecutive elements, so a wave holds sixteen 4x4,
 * four 8x8 or one 16x16 block (a 32x32 block takes 4 groups per lane) and all 64 lanes work whatever the
 * size.  Coefficients, prediction samples and results move as 8-byte (4-byte for 8-bit samples) vectors;
 * every HBM load of a block is issued before the first wait.  The two 1-D passes go through LDS once
 * (pass 1 output is read transposed by pass 2); pass 2 ends in registers in the layout the epilogue stores.
 * ======================================================================================= */
__device__ int8_t g_basis[5][1024];                 /* [log2-2] n x n DCT basis rows, [4] DST-VII; filled by ohk_init() */

template <typename PX, int LOG2>
__global__ __launch_bounds__(64) void residual_kernel(const OhBatch B)
{
    constexpr int N = 1 << LOG2, NG = N * N / 4;                  /* groups of 4 elements per block */
    constexpr int SLOTS = NG >= 64 ? 1 : 64 / NG, LPS = 64 / SLOTS, K = NG > 64 ? NG / 64 : 1;
    __shared__ __attribute__((aligned(16))) int16_t a[SLOTS][N * N];
    __shared__ __attribute__((aligned(16))) int16_t b[SLOTS][N * N];
    __shared__ __attribute__((aligned(16))) int8_t  m[LOG2 == 2 ? 2 : 1][N * N];      /* [1]: DST-VII */
    __shared__ int bbox[2];
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    const uint32_t cnt = f->tu_cnt[LOG2 - 2], t0 = blockIdx.x * SLOTS;
    if (t0 >= cnt)
        return;
    const int lane = threadIdx.x, slot = lane / LPS, sl = lane % LPS;
    const bool live = t0 + slot < cnt;                            /* a dead slot repeats the wave's first block and stores nothing */
    const DevTu dtu = gload(f->tu + f->tu_first[LOG2 - 2] + t0 + (live ? slot : 0));
    const OhTu tu = dtu.t;
    const bool sparse = tu.flags & OH_TUF_SPARSE;
    const int bd = f->pp.bit_depth;
    const bool is_tr = tu.kind == OH_TU_IDCT || tu.kind == OH_TU_DST4;
    /* a cross-component block is finished by cross_kernel once the luma residual of the picture is complete */
    const bool to_pic = (tu.kind == OH_TU_PCM || (tu.flags & OH_TUF_ADD_NOW)) && !(tu.flags & OH_TUF_CROSS);
    const bool add = to_pic && tu.kind != OH_TU_PCM;
    const GLOBAL short4v *__restrict__ cin = (const GLOBAL short4v *)(f->coeffs + tu.coeff_off);
    const uint64_t p0 = (uint64_t)f->cur.p[0], p1 = (uint64_t)f->cur.p[1], p2 = (uint64_t)f->cur.p[2];
    const int ds = tu.c_idx ? f->cur.stride[1] : f->cur.stride[0];
    GLOBAL PX *__restrict__ dst = G_MUT(PX, tu.c_idx == 0 ? p0 : (tu.c_idx == 1 ? p1 : p2)) + (size_t)tu.y * ds + tu.x;

    /* issue every load of the block(s) */
    short4v cv[K];
    int pv[K][4];
    if (LOG2 >= 4 && lane < 2) bbox[lane] = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int g = sl + 64 * k;
        cv[k] = sparse ? short4v{ 0, 0, 0, 0 } : cin[g];
        if (add) load4<PX>(dst + (size_t)((4 * g) >> LOG2) * ds + ((4 * g) & (N - 1)), pv[k]);
    }
    {
        const GLOBAL unsigned *__restrict__ basis = (const GLOBAL unsigned *)g_basis[LOG2 - 2];
#pragma unroll
        for (int k = 0; k < (NG + 63) / 64; k++)
            if (lane + 64 * k < NG) ((unsigned *)m[0])[lane + 64 * k] = basis[lane + 64 * k];
        if (LOG2 == 2 && lane >= 32 && lane < 36) ((unsigned *)m[LOG2 == 2 ? 1 : 0])[lane - 32] = ((const GLOBAL unsigned *)g_basis[4])[lane - 32];
    }
    /* LDS: coefficients; for the big sizes the bounding box of the non-zero coefficients (zero rows /
     * columns contribute nothing: what the reference's col_limit exploits, hevc_cabac.c:1927-1934) */
#pragma unroll
    for (int k = 0; k < K; k++)
        *(short4v *)(a[slot] + 4 * (sl + 64 * k)) = cv[k];
    __syncthreads();
    /* sparse hand-off (ohevc_frame.h): the block arrived as quantised levels; de-quantise (hevc_cabac.c:1478-1494,
     * 1818-1841: level * scale * scale_m + add >> shift, clipped to int16) and scatter them into the zeroed block */
    if (__builtin_amdgcn_ballot_w64(sparse) != 0) {
        if (sparse) {
            const GLOBAL uint32_t *__restrict__ rec = G_CONST(uint32_t, f->sparse) + dtu.sparse_off;
            const uint32_t w0 = rec[0], cnt = w0 & 0xffff, qp = (w0 >> 16) & 0xff, mid = w0 >> 24;
            const int shift = bd + LOG2 - 5;
            const uint32_t qp6 = qp < 74 ? qp : 0;            /* the reference's rem6[] / div6[] end two entries early (hevc_cabac.c:1428-1440): QP 74 / 75 scale like QP 0 */
            const long long radd = 1ll << (shift - 1), scale = (long long)c_level_scale[qp6 % 6] << (qp6 / 6);
            const bool flat = mid == OH_FLAT_MATRIX;
            const GLOBAL uint8_t *__restrict__ mtx = flat ? nullptr : G_CONST(uint8_t, f->scaling->sl[LOG2 - 2][flat ? 0 : mid]);
            const int dc_scale = !flat && LOG2 >= 4 ? G_CONST(uint8_t, f->scaling->sl_dc[LOG2 >= 4 ? LOG2 - 4 : 0])[mid] : 16;
            for (uint32_t k = sl; k < cnt; k += LPS) {
                const uint32_t w = rec[1 + k], pos = w & 0xffff;
                const int x = pos & (N - 1), y = pos >> LOG2;
                int scale_m = 16;
                if (!flat)
                    scale_m = (x || y || LOG2 < 4) ? mtx[LOG2 == 3 ? (y << 3) + x : LOG2 == 4 ? ((y >> 1) << 3) + (x >> 1) : LOG2 == 5 ? ((y >> 2) << 3) + (x >> 2) : (y << 2) + x]
                                                    : dc_scale;
                long long v = ((long long)(int16_t)(w >> 16) * scale * scale_m + radd) >> shift;
                a[slot][pos] = (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
            }
        }
        __syncthreads();
        if (sparse) {
#pragma unroll
            for (int k = 0; k < K; k++) cv[k] = *(const short4v *)(a[slot] + 4 * (sl + 64 * k));
        }
    }
    int my_r = 0, my_c = 0;
    if (LOG2 >= 4) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int g = sl + 64 * k;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (cv[k][j]) { my_r = max(my_r, (4 * g) >> LOG2); my_c = max(my_c, ((4 * g) & (N - 1)) + j); }
        }
    }
    int nr = N, nc = N;
    if (LOG2 >= 4) {
        if (my_r) atomicMax(&bbox[0], my_r);
        if (my_c) atomicMax(&bbox[1], my_c);
        __syncthreads();
        nr = bbox[0] + 1; nc = bbox[1] + 1;                       /* rows / columns that hold coefficients */
    }
    const int8_t *__restrict__ mm = m[LOG2 == 2 && tu.kind == OH_TU_DST4 ? 1 : 0];
    int res[K][4];

    /* stage 1 -> b: transform blocks run pass 1 (down the columns, shift 7; group = output row i, four
     * consecutive columns); the others scale / rotate their coefficients */
    if (is_tr) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int g = sl + 64 * k;
            const int i = (4 * g) >> LOG2, col = (4 * g) & (N - 1);
            int acc[4] = { 0, 0, 0, 0 };
            if (col < nc) {
                if (LOG2 <= 3) {
#pragma unroll
                    for (int kk = 0; kk < N; kk++) {
                        const int c = mm[kk * N + i];
                        const short4v av = *(const short4v *)(a[slot] + kk * N + col);
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[j] += c * av[j];
                    }
                } else {
                    for (int kk = 0; kk < nr; kk++) {
                        const int c = mm[kk * N + i];
                        const short4v av = *(const short4v *)(a[slot] + kk * N + col);
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[j] += c * av[j];
                    }
                }
            }
            short4v o;
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (short)clip16((acc[j] + 64) >> 7);
            *(short4v *)(b[slot] + 4 * g) = o;
        }
    } else {
        const bool skip = tu.kind == OH_TU_SKIP;
        const int shift = 15 - bd - LOG2;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int g = sl + 64 * k;
            short4v o;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = 4 * g + j;
                int c0 = LOG2 == 2 && skip && (tu.flags & OH_TUF_ROTATE) ? a[slot][15 - e] : cv[k][j];   /* hevc_cabac.c:1879-1882, 4x4 only */
                if (skip) c0 = shift > 0 ? (int16_t)((c0 + (1 << (shift - 1))) >> shift) : (int16_t)(c0 << -shift);
                o[j] = (short)c0;
            }
            *(short4v *)(b[slot] + 4 * g) = o;
        }
    }
    __syncthreads();
    /* stage 2: rdpcm accumulation (hevcdsp_template.c:114-136), serial along the direction, one lane per line */
    const bool any_plain = __builtin_amdgcn_ballot_w64(!is_tr) != 0;
    if (any_plain) {
        if (!is_tr && (tu.kind == OH_TU_SKIP || tu.kind == OH_TU_BYPASS) && (tu.flags & OH_TUF_RDPCM) && sl < N) {
            int16_t *bb = b[slot];
            if (tu.flags & OH_TUF_RDPCM_VER)
                for (int y = 1; y < N; y++) bb[y * N + sl] = (int16_t)(bb[y * N + sl] + bb[(y - 1) * N + sl]);
            else
                for (int x = 1; x < N; x++) bb[sl * N + x] = (int16_t)(bb[sl * N + x] + bb[sl * N + x - 1]);
        }
        __syncthreads();
    }
    /* stage 3 -> registers: pass 2 along the rows (group = row, four consecutive outputs i) / plain read */
    if (is_tr) {
        const int shift = 20 - bd, addc = 1 << (shift - 1);
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int g = sl + 64 * k;
            const int row = (4 * g) >> LOG2, i0 = (4 * g) & (N - 1);
            int acc[4] = { 0, 0, 0, 0 };
            if (LOG2 <= 3) {
#pragma unroll
                for (int kk = 0; kk < N; kk++) {
                    const unsigned c4 = *(const unsigned *)(mm + kk * N + i0);
                    const int bv = b[slot][row * N + kk];
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[j] += (int)(int8_t)(c4 >> (8 * j)) * bv;
                }
            } else {
                for (int kk = 0; kk < nc; kk++) {
                    const unsigned c4 = *(const unsigned *)(mm + kk * N + i0);
                    const int bv = b[slot][row * N + kk];
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[j] += (int)(int8_t)(c4 >> (8 * j)) * bv;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; j++) res[k][j] = clip16((acc[j] + addc) >> shift);
        }
    } else {
#pragma unroll
        for (int k = 0; k < K; k++) {
            const short4v o = *(const short4v *)(b[slot] + 4 * (sl + 64 * k));
#pragma unroll
            for (int j = 0; j < 4; j++) res[k][j] = o[j];
        }
    }

    /* epilogue */
    if (!live)
        return;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int g = sl + 64 * k;
        if (to_pic) {
            int o[4];
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = add ? clip_px(pv[k][j] + res[k][j], bd) : (res[k][j] & (sizeof(PX) == 1 ? 0xff : 0xffff));
            store4<PX>(dst + (size_t)((4 * g) >> LOG2) * ds + ((4 * g) & (N - 1)), o[0], o[1], o[2], o[3]);
        }
        if (!to_pic || (tu.flags & OH_TUF_KEEP_RES)) {
            short4v o;
#pragma unroll
            for (int j = 0; j < 4; j++) o[j] = (short)res[k][j];
            *((GLOBAL short4v *)(f->res + tu.coeff_off) + g) = o;
        }
    }
}

/* cross-component prediction (4:4:4 range extension; hevc_cabac.c:1942-1947 for coded chroma blocks, hevc.c:1319-1331 /
 * 1352-1364 for cbf 0): chroma residual += (res_scale_val * luma residual) >> 3 in int16 storage, then the block is added
 * to the prediction (inter) or left in the pool for the intra pass.  Runs after every inverse transform of the picture. */
template <typename PX>
__global__ __launch_bounds__(64) void cross_kernel(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    if (blockIdx.x >= f->n_cross)
        return;
    const DevCross c = gload(f->cross + blockIdx.x);
    const int lane = threadIdx.x, bd = f->pp.bit_depth, log2 = c.log2_size, n = 1 << log2, ng = (n * n) >> 2;
    const int ds = f->cur.stride[c.c_idx];
    GLOBAL PX *__restrict__ dst = G_MUT(PX, f->cur.p[c.c_idx]) + (size_t)c.y * ds + c.x;
    GLOBAL short4v *__restrict__ rc = (GLOBAL short4v *)(f->res + c.res_c);
    const GLOBAL short4v *__restrict__ ry = (const GLOBAL short4v *)(f->res + c.res_y);
    for (int g = lane; g < ng; g += 64) {
        const short4v a = rc[g], y = ry[g];
        short4v r;
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = (short)(a[j] + ((c.scale * y[j]) >> 3));
        if (c.flags & OH_TUF_ADD_NOW) {
            int pv[4];
            GLOBAL PX *__restrict__ d = dst + (size_t)((4 * g) >> log2) * ds + ((4 * g) & (n - 1));
            load4<PX>(d, pv);
            store4<PX>(d, clip_px(pv[0] + r[0], bd), clip_px(pv[1] + r[1], bd), clip_px(pv[2] + r[2], bd), clip_px(pv[3] + r[3], bd));
        } else {
            rc[g] = r;
        }
    }
}

/* =========================================================================================
 * launcher
 * ======================================================================================= */
int ohk_init_residual(void)
{
    static const int8_t c[32] = { 64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67,
                                  64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4 };
    int8_t m[32][32];
    for (int k = 0; k < 32; k++)
        for (int n = 0; n < 32; n++) {
            int a = (k * (2 * n + 1)) & 127;
            if (a > 64) a = 128 - a;
            m[k][n] = (int8_t)(k == 0 ? 64 : (a == 32 ? 0 : (a < 32 ? c[a] : -c[64 - a])));
        }
    static const int8_t dst7[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };
    static int8_t basis[5][1024];
    for (int l = 0; l < 4; l++) {                     /* n-point basis: every (32/n)-th row of the 32-point matrix */
        int n = 4 << l, step = 32 / n;
        for (int k = 0; k < n; k++)
            for (int i = 0; i < n; i++) basis[l][k * n + i] = m[k * step][i];
    }
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < 4; i++) basis[4][k * 4 + i] = dst7[k][i];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_basis), basis, sizeof(basis)) != hipSuccess)
        return -1;
    return 0;
}

extern "C" void ohk_residual(const OhBatch *B, int n, const OhPicParams *p, const uint32_t max_cnt[4], hipStream_t st)
{
    /* one launch per transform size; a wave holds 16 / 4 / 1 / 1 blocks */
#define RES_LAUNCH(PX)                                                                                                       \
    do {                                                                                                                     \
        if (max_cnt[0]) hipLaunchKernelGGL(HIP_KERNEL_NAME(residual_kernel<PX, 2>), dim3((max_cnt[0] + 15) / 16, n), dim3(64), 0, st, *B); \
        if (max_cnt[1]) hipLaunchKernelGGL(HIP_KERNEL_NAME(residual_kernel<PX, 3>), dim3((max_cnt[1] + 3) / 4, n), dim3(64), 0, st, *B);   \
        if (max_cnt[2]) hipLaunchKernelGGL(HIP_KERNEL_NAME(residual_kernel<PX, 4>), dim3(max_cnt[2], n), dim3(64), 0, st, *B);             \
        if (max_cnt[3]) hipLaunchKernelGGL(HIP_KERNEL_NAME(residual_kernel<PX, 5>), dim3(max_cnt[3], n), dim3(64), 0, st, *B);             \
    } while (0)
    if (p->bit_depth == 8) RES_LAUNCH(uint8_t); else RES_LAUNCH(uint16_t);
#undef RES_LAUNCH
}

extern "C" void ohk_cross(const OhBatch *B, int n, const OhPicParams *p, uint32_t max_cross, hipStream_t st)
{
    if (!max_cross) return;
    if (p->bit_depth == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(cross_kernel<uint8_t>), dim3(max_cross, n), dim3(64), 0, st, *B);
    else                   hipLaunchKernelGGL(HIP_KERNEL_NAME(cross_kernel<uint16_t>), dim3(max_cross, n), dim3(64), 0, st, *B);
}
