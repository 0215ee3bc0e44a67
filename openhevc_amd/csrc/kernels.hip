/*
 * kernels.hip — CDNA4 (gfx950) kernels of the block-reconstruction passes.
 *
 * Integer/byte work throughout (no MFMA): the passes are bounded by HBM/L2 traffic and by
 * per-block latency, so the design rules are wave64-sized work units, LDS staging of the
 * separable-filter windows and coefficient blocks, and coalesced row accesses.
 * Arithmetic follows the reference exactly (file:line cited per kernel, paths relative to
 * /root/reference/libavcodec/); tests/ check every kernel bit-for-bit against the CPU checker.
 *
 *   pass 1  mc_kernel         one wave per <=16x16 luma tile (+ its chroma), LDS window + 2-stage filter
 *   pass 2  residual_kernel   one wave per transform block, two LDS matrix passes
 *   pass 3  intra_ctu_kernel  one workgroup per CTU of one wavefront level, waves take the blocks of a sub-level
 *   pass 4  deblock_*_kernel  one lane per 4-line edge segment, V pass then H pass, in place
 *   pass 5  sao_kernel        one lane per sample, cur -> out
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dev_frame.h"
#include "kernels.h"

/* ---- constant tables (H.265 facts; same numbers as hevcdsp.c:879-944,1028-1042, hevcpred_template.c:430-437,
 *      hevc_filter.c:50-60) ---- */
__constant__ int8_t  c_qpel[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 },
                                      { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
__constant__ int8_t  c_epel[8][8] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 },
                                      { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
__constant__ int8_t  c_angle[33] = { 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                     -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };
__constant__ int16_t c_inv_angle[15] = { -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };
__constant__ uint8_t c_tc[54] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                  5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
__constant__ uint8_t c_beta[52] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24,
                                    26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };
__constant__ uint8_t c_qpc[14] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };
__constant__ int8_t  c_dst7[4][4] = { { 29, 55, 74, 84 }, { 74, 74, 0, -74 }, { 84, -29, -74, 55 }, { 55, -84, 74, -29 } };
/* 32-point inverse DCT basis, filled by ohk_init() from the folded cosine table */
__device__ int8_t g_dct[32][32];

static __device__ __forceinline__ int clip3(int v, int lo, int hi) { return min(max(v, lo), hi); }
static __device__ __forceinline__ int clip_px(int v, int bd) { return min(max(v, 0), (1 << bd) - 1); }
static __device__ __forceinline__ int clip16(int v) { return min(max(v, -32768), 32767); }
static __device__ __forceinline__ int hsh(const OhPicParams &p, int c) { return c && (p.chroma_format_idc == 1 || p.chroma_format_idc == 2); }
static __device__ __forceinline__ int vsh(const OhPicParams &p, int c) { return c && p.chroma_format_idc == 1; }

/* =========================================================================================
 * pass 1: inter prediction — hevcdsp_template.c:610-1609 through the drivers hevc.c:1641-1949;
 * picture-edge emulation (videodsp_template.c:26-101) is coordinate clamping while loading.
 * ======================================================================================= */
#define WIN_STRIDE 24
template <typename PX>
__global__ __launch_bounds__(64) void mc_kernel(const DevFrame *__restrict__ f)
{
    __shared__ uint16_t win[23 * WIN_STRIDE];
    __shared__ int16_t  tmp[23 * 16];
    const int lane = threadIdx.x;
    const int c = blockIdx.y;
    const DevTile t = f->tiles[blockIdx.x];
    const OhPu pu = f->pu[t.pu];
    const OhPicParams &pp = f->pp;
    const int bd = pp.bit_depth;
    const int hs = hsh(pp, c), vs = vsh(pp, c);
    const int taps = c ? 4 : 8, before = taps / 2 - 1;
    const int bx = (pu.x + t.ox) >> hs, by = (pu.y + t.oy) >> vs;
    const int bw = t.w >> hs, bh = t.h >> vs;
    const int npx = bw * bh;
    int v[2][4];

    for (int l = 0; l < 2; l++) {
        if (pu.ref[l] == OH_NO_REF)
            continue;
        const DevPlanes &rp = f->refs[pu.ref[l]];
        const PX *__restrict__ src = (const PX *)rp.p[c];
        const int sstride = rp.stride[c], pw = rp.w[c], ph = rp.h[c];
        const int mvx = pu.mv[l][0], mvy = pu.mv[l][1];
        int fx, fy, ix, iy;
        if (c == 0) {
            fx = mvx & 3; fy = mvy & 3; ix = mvx >> 2; iy = mvy >> 2;
        } else {                                    /* hevc.c:1807-1813 */
            fx = (mvx & ((1 << (2 + hs)) - 1)) << (1 - hs);
            fy = (mvy & ((1 << (2 + vs)) - 1)) << (1 - vs);
            ix = mvx >> (2 + hs); iy = mvy >> (2 + vs);
        }
        const int ww = bw + taps - 1, wh = bh + taps - 1;
        const int wx0 = bx + ix - before, wy0 = by + iy - before;
        for (int e = lane; e < ww * wh; e += 64) {
            int wy = e / ww, wx = e - wy * ww;
            int sx = clip3(wx0 + wx, 0, pw - 1), sy = clip3(wy0 + wy, 0, ph - 1);
            win[wy * WIN_STRIDE + wx] = src[(size_t)sy * sstride + sx];
        }
        __syncthreads();
        const int8_t *cx = c ? c_epel[fx] : c_qpel[fx];
        const int8_t *cy = c ? c_epel[fy] : c_qpel[fy];
        for (int e = lane; e < wh * bw; e += 64) {
            int r = e / bw, x = e - r * bw;
            int val;
            if (fx) {
                int s = 0;
                for (int k = 0; k < taps; k++)
                    s += cx[k] * win[r * WIN_STRIDE + x + k];
                val = s >> (bd - 8);
            } else {
                val = win[r * WIN_STRIDE + x + before];
            }
            tmp[r * 16 + x] = (int16_t)val;
        }
        __syncthreads();
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) {
            int idx = lane + 64 * k4;
            int val = 0;
            if (idx < npx) {
                int py = idx / bw, px = idx - py * bw;
                if (fy) {
                    int s = 0;
                    for (int k = 0; k < taps; k++)
                        s += cy[k] * tmp[(py + k) * 16 + px];
                    val = fx ? (s >> 6) : (s >> (bd - 8));
                } else {
                    int tv = tmp[(py + before) * 16 + px];
                    val = fx ? tv : (tv << (14 - bd));
                }
            }
            v[l][k4] = val;
        }
        __syncthreads();
    }

    PX *__restrict__ dst = (PX *)f->cur.p[c];
    const int dstride = f->cur.stride[c];
    const bool u0 = pu.ref[0] != OH_NO_REF, u1 = pu.ref[1] != OH_NO_REF;
    const bool weighted = pu.wp != OH_NO_WP;
    int w0 = 0, w1 = 0, o0 = 0, o1 = 0, denom = 0;
    if (weighted) {
        const OhWeights wp = f->wp[pu.wp];
        w0 = wp.w[0][c]; w1 = wp.w[1][c];
        o0 = wp.o[0][c] * (1 << (bd - 8)); o1 = wp.o[1][c] * (1 << (bd - 8));
        denom = wp.log2_denom[c ? 1 : 0];
    }
#pragma unroll
    for (int k4 = 0; k4 < 4; k4++) {
        int idx = lane + 64 * k4;
        if (idx >= npx)
            continue;
        int py = idx / bw, px = idx - py * bw;
        int r;
        if (u0 && u1) {
            int a = (int16_t)v[0][k4], b = v[1][k4];            /* list 0 went through an int16 tmp, hevc.c:1761 */
            if (!weighted) {
                int shift = 15 - bd;
                r = (b + a + (1 << (shift - 1))) >> shift;
            } else {
                int log2wd = denom + 14 - bd;
                r = (b * w1 + a * w0 + ((o0 + o1 + 1) << log2wd)) >> (log2wd + 1);
            }
        } else {
            int a = u0 ? v[0][k4] : v[1][k4];
            if (!weighted) {
                int shift = 14 - bd;
                r = (a + (1 << (shift - 1))) >> shift;
            } else {
                int shift = denom + 14 - bd;
                r = ((a * (u0 ? w0 : w1) + (1 << (shift - 1))) >> shift) + (u0 ? o0 : o1);
            }
        }
        dst[(size_t)(by + py) * dstride + bx + px] = (PX)clip_px(r, bd);
    }
}

/* =========================================================================================
 * pass 2: residual — hevcdsp_template.c:114-316 dispatched as hevc_cabac.c:1868-1949;
 * inter blocks are added to the prediction here (transform_add, :45-111), intra blocks leave
 * their residual in f->res for pass 3.
 * ======================================================================================= */
template <typename PX>
__global__ __launch_bounds__(64) void residual_kernel(const DevFrame *__restrict__ f)
{
    __shared__ int16_t a[1024];
    __shared__ int16_t b[1024];
    __shared__ int8_t  m[32 * 32];
    const int lane = threadIdx.x;
    const OhTu tu = f->tu[blockIdx.x];
    const int bd = f->pp.bit_depth;
    const int log2 = tu.log2_size, n = 1 << log2, n2 = n * n;
    const int16_t *__restrict__ cin = f->coeffs + tu.coeff_off;

    for (int e = lane; e < n2; e += 64)
        a[e] = cin[e];

    if (tu.kind == OH_TU_IDCT || tu.kind == OH_TU_DST4) {
        /* basis rows: every (32/n)-th row of the 32-point matrix, or the DST-VII matrix */
        const int step = 32 >> log2;
        for (int e = lane; e < n2; e += 64) {
            int k = e >> log2, i = e & (n - 1);
            m[e] = tu.kind == OH_TU_DST4 ? c_dst7[k][i] : g_dct[k * step][i];
        }
        __syncthreads();
        for (int e = lane; e < n2; e += 64) {          /* pass 1: down the columns, shift 7 */
            int i = e >> log2, col = e & (n - 1);
            int acc = 0;
            for (int k = 0; k < n; k++)
                acc += m[k * n + i] * a[k * n + col];
            b[i * n + col] = (int16_t)clip16((acc + 64) >> 7);
        }
        __syncthreads();
        const int shift = 20 - bd, add = 1 << (shift - 1);
        for (int e = lane; e < n2; e += 64) {          /* pass 2: along the rows */
            int row = e >> log2, i = e & (n - 1);
            int acc = 0;
            for (int k = 0; k < n; k++)
                acc += m[k * n + i] * b[row * n + k];
            a[e] = (int16_t)clip16((acc + add) >> shift);
        }
        __syncthreads();
    } else if (tu.kind == OH_TU_SKIP || tu.kind == OH_TU_BYPASS) {
        __syncthreads();
        if (tu.kind == OH_TU_SKIP) {
            if (tu.flags & OH_TUF_ROTATE) {            /* hevc_cabac.c:1879-1882, 4x4 only */
                int16_t t0 = lane < 16 ? a[15 - lane] : 0;
                __syncthreads();
                if (lane < 16) a[lane] = t0;
                __syncthreads();
            }
            const int shift = 15 - bd - log2;
            for (int e = lane; e < n2; e += 64) {
                int cv = a[e];
                a[e] = shift > 0 ? (int16_t)((cv + (1 << (shift - 1))) >> shift) : (int16_t)(cv << -shift);
            }
            __syncthreads();
        }
        if (tu.flags & OH_TUF_RDPCM) {                 /* hevcdsp_template.c:114-136 */
            if (lane < n) {
                if (tu.flags & OH_TUF_RDPCM_VER)
                    for (int y = 1; y < n; y++) a[y * n + lane] = (int16_t)(a[y * n + lane] + a[(y - 1) * n + lane]);
                else
                    for (int x = 1; x < n; x++) a[lane * n + x] = (int16_t)(a[lane * n + x] + a[lane * n + x - 1]);
            }
            __syncthreads();
        }
    } else {
        __syncthreads();
    }

    if (tu.kind == OH_TU_PCM || (tu.flags & OH_TUF_ADD_NOW)) {
        PX *__restrict__ dst = (PX *)f->cur.p[tu.c_idx] + (size_t)tu.y * f->cur.stride[tu.c_idx] + tu.x;
        const int ds = f->cur.stride[tu.c_idx];
        for (int e = lane; e < n2; e += 64) {
            int y = e >> log2, x = e & (n - 1);
            if (tu.kind == OH_TU_PCM) dst[(size_t)y * ds + x] = (PX)a[e];
            else                      dst[(size_t)y * ds + x] = (PX)clip_px(dst[(size_t)y * ds + x] + a[e], bd);
        }
    } else {
        int16_t *__restrict__ r = f->res + tu.coeff_off;
        for (int e = lane; e < n2; e += 64)
            r[e] = a[e];
    }
}

/* =========================================================================================
 * pass 3: intra prediction as a CTU wavefront — hevcpred_template.c:30-538
 * (constrained_intra_pred_flag == 0), each block followed by its residual (transform_add).
 *
 * One workgroup reconstructs one CTU; its waves take the blocks of the current SUB-LEVEL (blocks
 * of one sub-level never read each other), a workgroup barrier separates sub-levels, so the
 * dependent chain inside a CTU costs barriers inside one CU instead of kernel launches.  CTUs of
 * one launch are mutually independent (same wavefront level, recorder.c).
 * L[] / T[] hold left[-1..2n-1] / top[-1..2n-1] at index i+1 (one set per wave).
 * ======================================================================================= */
#define INTRA_WAVES 8
struct IntraLds { int L[66], T[66], FL[66], FT[66], R[3 * 32 + 4]; };

/* LDS hand-offs between lanes of ONE wave: DS operations of a wave execute in order, the fence
 * only stops the compiler from moving them */
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

template <typename PX>
static __device__ __forceinline__ void intra_block(const DevFrame *__restrict__ f, const OhIntra it, IntraLds &s, const int lane)
{
    int *L = s.L, *T = s.T, *FL = s.FL, *FT = s.FT, *R = s.R;
    const OhPicParams &pp = f->pp;
    const int bd = pp.bit_depth, c = it.c_idx, log2 = it.log2_size, n = 1 << log2, mode = it.mode;
    const int stride = f->cur.stride[c], pw = f->cur.w[c], ph = f->cur.h[c];
    PX *__restrict__ src = (PX *)f->cur.p[c] + (size_t)it.y * stride + it.x;
    bool a_bl = it.avail & OH_AV_BOTTOM_LEFT, a_l = it.avail & OH_AV_LEFT, a_ul = it.avail & OH_AV_UP_LEFT;
    bool a_u = it.avail & OH_AV_UP, a_ur = it.avail & OH_AV_UP_RIGHT;
    const int bl_size = max(min(it.y + 2 * n, ph) - (it.y + n), 0);     /* :111-114 */
    const int tr_size = max(min(it.x + 2 * n, pw) - (it.x + n), 0);
    const int i = lane;                                                /* element this lane owns */

    /* gather (:164-183) */
    if (i < 2 * n) {
        int tv = 0, lv = 0;
        if (i < n) {
            if (a_u) tv = src[-(ptrdiff_t)stride + i];
            if (a_l) lv = src[(ptrdiff_t)i * stride - 1];
        } else {
            if (a_ur) tv = src[-(ptrdiff_t)stride + (i - n < tr_size ? i : n + tr_size - 1)];
            if (a_bl) lv = src[(ptrdiff_t)(i - n < bl_size ? i : n + bl_size - 1) * stride - 1];
        }
        T[i + 1] = tv; L[i + 1] = lv;
    }
    if (lane == 0) { int cv = a_ul ? (int)src[-(ptrdiff_t)stride - 1] : 0; L[0] = cv; T[0] = cv; }
    WSYNC();

    /* substitution of unavailable samples (:251-286) */
    if (!a_bl) {
        if (a_l) {
            int val = L[n];                              /* left[n-1] */
            WSYNC();
            if (i >= n && i < 2 * n) L[i + 1] = val;
        } else if (a_ul) {
            int val = L[0];
            WSYNC();
            if (i < 2 * n) L[i + 1] = val;
            a_l = true;
        } else if (a_u) {
            int val = T[1];
            WSYNC();
            if (i < 2 * n) L[i + 1] = val;
            if (lane == 0) L[0] = val;
            a_ul = a_l = true;
        } else if (a_ur) {
            int val = T[n + 1];
            WSYNC();
            if (i < n) T[i + 1] = val;
            if (i < 2 * n) L[i + 1] = val;
            if (lane == 0) L[0] = val;
            a_u = a_ul = a_l = true;
        } else {
            int val = 1 << (bd - 1);
            if (i < 2 * n) { T[i + 1] = val; L[i + 1] = val; }
            if (lane == 0) L[0] = val;
        }
        WSYNC();
    }
    if (!a_l) {
        int val = L[n + 1];                              /* left[n] */
        WSYNC();
        if (i < n) L[i + 1] = val;
        WSYNC();
    }
    if (!a_ul) {
        if (lane == 0) L[0] = L[1];
        WSYNC();
    }
    if (!a_u) {
        int val = L[0];
        WSYNC();
        if (i < n) T[i + 1] = val;
        WSYNC();
    }
    if (!a_ur) {
        int val = T[n];                                  /* top[n-1] */
        WSYNC();
        if (i >= n && i < 2 * n) T[i + 1] = val;
    }
    if (lane == 0) T[0] = L[0];
    WSYNC();

    /* smoothing (:288-326) */
    const int *left = L + 1, *top = T + 1;
    if (!pp.intra_smoothing_disabled && (c == 0 || pp.chroma_format_idc == 3) && mode != 1 && n != 4) {
        int d26 = abs(mode - 26), d10 = abs(mode - 10);
        int dist = min(d26, d10);
        int thresh = log2 == 3 ? 7 : (log2 == 4 ? 1 : 0);
        if (dist > thresh) {
            bool strong = false;
            if (pp.strong_intra_smoothing && c == 0 && log2 == 5) {
                int lim = 1 << (bd - 5);
                strong = abs(top[-1] + top[63] - 2 * top[31]) < lim && abs(left[-1] + left[63] - 2 * left[31]) < lim;
            }
            if (strong) {
                if (i < 63) {
                    FT[i + 1] = ((63 - i) * top[-1] + (i + 1) * top[63] + 32) >> 6;
                    FL[i + 1] = ((63 - i) * left[-1] + (i + 1) * left[63] + 32) >> 6;
                } else {
                    FT[64] = top[63]; FL[64] = left[63];
                    FT[0] = top[-1]; FL[0] = left[-1];
                }
            } else {
                if (i < 2 * n - 1) {
                    FL[i + 1] = (left[i + 1] + 2 * left[i] + left[i - 1] + 2) >> 2;
                    FT[i + 1] = (top[i + 1] + 2 * top[i] + top[i - 1] + 2) >> 2;
                } else if (i == 2 * n - 1) {
                    FL[i + 1] = left[i]; FT[i + 1] = top[i];
                }
                if (lane == 0) FT[0] = FL[0] = (left[0] + 2 * left[-1] + top[0] + 2) >> 2;
            }
            left = FL + 1; top = FT + 1;
            WSYNC();
        }
    }

    /* prediction (:359-538): pixel idx = lane + 64*k -> (y = idx / n, x = idx % n) */
    const int npx = n * n;
    const int16_t *res = it.tu != OH_NO_COEFF ? f->res + f->tu[it.tu].coeff_off : nullptr;
    if (mode >= 2) {
        const int angle = c_angle[mode - 2];
        const bool vertical = mode >= 18;
        const int *mainr = vertical ? top : left, *side = vertical ? left : top;
        int *ref = R + 32;                                /* ref[k] == main[k-1] */
        const int last = (n * angle) >> 5;
        for (int k = lane; k <= 2 * n; k += 64) ref[k] = mainr[k - 1];
        if (angle < 0 && last < -1) {
            int inv = c_inv_angle[mode - 11];
            int k = last + lane;
            if (k <= -1) ref[k] = side[-1 + ((k * inv + 128) >> 8)];
        }
        WSYNC();
        for (int idx = lane; idx < npx; idx += 64) {
            int y = idx >> log2, x = idx & (n - 1);
            int aa = vertical ? y : x, bb = vertical ? x : y;
            int id = ((aa + 1) * angle) >> 5, fact = ((aa + 1) * angle) & 31;
            int v = fact ? ((32 - fact) * ref[bb + id + 1] + fact * ref[bb + id + 2] + 16) >> 5 : ref[bb + id + 1];
            if (c == 0 && n < 32) {                       /* :474-477, :501-508 */
                if (mode == 26 && x == 0) v = clip_px(top[0] + ((left[y] - left[-1]) >> 1), bd);
                if (mode == 10 && y == 0) v = clip_px(left[0] + ((top[x] - top[-1]) >> 1), bd);
            }
            if (res) v = clip_px(v + res[idx], bd);
            src[(size_t)y * stride + x] = (PX)v;
        }
    } else if (mode == 0) {
        for (int idx = lane; idx < npx; idx += 64) {
            int y = idx >> log2, x = idx & (n - 1);
            int v = ((n - 1 - x) * left[y] + (x + 1) * top[n] + (n - 1 - y) * top[x] + (y + 1) * left[n] + n) >> (log2 + 1);
            if (res) v = clip_px(v + res[idx], bd);
            src[(size_t)y * stride + x] = (PX)v;
        }
    } else {
        int part = 0;
        if (i < n) part = left[i] + top[i];
        for (int o = 32; o > 0; o >>= 1) part += __shfl_down(part, o);
        int dc = (__shfl(part, 0) + n) >> (log2 + 1);
        for (int idx = lane; idx < npx; idx += 64) {
            int y = idx >> log2, x = idx & (n - 1);
            int v = dc;
            if (c == 0 && n < 32) {                       /* :410-416 */
                if (x == 0 && y == 0) v = (left[0] + 2 * dc + top[0] + 2) >> 2;
                else if (y == 0)      v = (top[x] + 3 * dc + 2) >> 2;
                else if (x == 0)      v = (left[y] + 3 * dc + 2) >> 2;
            }
            if (res) v = clip_px(v + res[idx], bd);
            src[(size_t)y * stride + x] = (PX)v;
        }
    }
    WSYNC();                                              /* LDS is reused by this wave's next block */
}

template <typename PX>
__global__ __launch_bounds__(64 * INTRA_WAVES) void intra_ctu_kernel(const DevFrame *__restrict__ f, uint32_t first_ctu)
{
    __shared__ IntraLds lds[INTRA_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const OhIntraCtu ctu = f->ictu[first_ctu + blockIdx.x];
    const uint32_t *__restrict__ ss = f->sub_start + ctu.sub_first;
    for (int s = 0; s < ctu.n_sub; s++) {
        const uint32_t b0 = ss[s], b1 = ss[s + 1];
        for (uint32_t b = b0 + wave; b < b1; b += INTRA_WAVES)
            intra_block<PX>(f, f->intra[b], lds[wave], lane);
        /* workgroup barrier + workgroup-scope release/acquire of the global stores: the next
         * sub-level reads the samples just written by other waves of this CU */
        __syncthreads();
    }
}

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
__global__ __launch_bounds__(256) void deblock_luma_kernel(const DevFrame *__restrict__ f)
{
    const OhPicParams &pp = f->pp;
    const int W = pp.width, H = pp.height, bd = pp.bit_depth;
    /* x index runs fastest in both directions so that a wave touches neighbouring addresses */
    const int gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y;
    int x, y;
    if (!HORIZ) { x = 8 * (gx + 1); y = 4 * gy; } else { x = 4 * gx; y = 8 * (gy + 1); }
    if (x >= W || y >= H)
        return;
    const int bsw = W >> 2;
    const int bs = (HORIZ ? f->hbs : f->vbs)[(x + y * bsw) >> 2];
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
    PX *pix = (PX *)f->cur.p[0] + (size_t)y * stride + x;
    const ptrdiff_t xs = HORIZ ? stride : 1, ys = HORIZ ? 1 : stride;

    int P[4][4], Q[4][4];                                  /* [line][distance from the edge] */
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            P[d][k] = pix[d * ys - (k + 1) * xs];
            Q[d][k] = pix[d * ys + k * xs];
        }
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
                pix[d * ys - 1 * xs] = (PX)(p0 + clip3(((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3) - p0, -tc2, tc2));
                pix[d * ys - 2 * xs] = (PX)(p1 + clip3(((p2 + p1 + p0 + q0 + 2) >> 2) - p1, -tc2, tc2));
                pix[d * ys - 3 * xs] = (PX)(p2 + clip3(((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3) - p2, -tc2, tc2));
            }
            if (!no_q) {
                pix[d * ys + 0 * xs] = (PX)(q0 + clip3(((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3) - q0, -tc2, tc2));
                pix[d * ys + 1 * xs] = (PX)(q1 + clip3(((p0 + q0 + q1 + q2 + 2) >> 2) - q1, -tc2, tc2));
                pix[d * ys + 2 * xs] = (PX)(q2 + clip3(((2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3) - q2, -tc2, tc2));
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
            if (!no_p) pix[d * ys - xs] = (PX)clip_px(p0 + delta, bd);
            if (!no_q) pix[d * ys]      = (PX)clip_px(q0 - delta, bd);
            if (!no_p && nd_p) pix[d * ys - 2 * xs] = (PX)clip_px(p1 + clip3((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1, -tc_2, tc_2), bd);
            if (!no_q && nd_q) pix[d * ys + xs]     = (PX)clip_px(q1 + clip3((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1, -tc_2, tc_2), bd);
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
__global__ __launch_bounds__(256) void deblock_chroma_kernel(const DevFrame *__restrict__ f)
{
    const OhPicParams &pp = f->pp;
    const int W = pp.width, H = pp.height, bd = pp.bit_depth;
    const int hs = hsh(pp, 1), vs = vsh(pp, 1), hh = 1 << hs, vv = 1 << vs;
    const int gx = blockIdx.x * blockDim.x + threadIdx.x, gy = blockIdx.y, c = 1 + blockIdx.z;
    int x, y;                                               /* luma coordinates of the segment */
    if (!HORIZ) { x = 8 * hh * (gx + 1); y = 4 * vv * gy; } else { x = 4 * hh * gx; y = 8 * vv * (gy + 1); }
    if (x >= W || y >= H)
        return;
    const int bsw = W >> 2;
    const int bs = (HORIZ ? f->hbs : f->vbs)[(x + y * bsw) >> 2];
    if (bs != 2)
        return;
    const int lc = pp.log2_ctb_size, ctbw = (W + (1 << lc) - 1) >> lc;
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
    PX *pix = (PX *)f->cur.p[c] + (size_t)(y >> vs) * stride + (x >> hs);
    const ptrdiff_t xs = HORIZ ? stride : 1, ys = HORIZ ? 1 : stride;
#pragma unroll
    for (int d = 0; d < 4; d++) {
        int p1 = pix[d * ys - 2 * xs], p0 = pix[d * ys - xs], q0 = pix[d * ys], q1 = pix[d * ys + xs];
        int delta = clip3((((q0 - p0) * 4) + p1 - q1 + 4) >> 3, -tc, tc);
        if (!no_p) pix[d * ys - xs] = (PX)clip_px(p0 + delta, bd);
        if (!no_q) pix[d * ys]      = (PX)clip_px(q0 - delta, bd);
    }
}

/* =========================================================================================
 * pass 5: SAO — hevcdsp_template.c:340-567 driven per CTB by sao_filter_CTB (hevc_filter.c:197-322),
 * here one whole-picture pass from the deblocked planes (cur) into the output planes (out).
 * ======================================================================================= */
template <typename PX>
__global__ __launch_bounds__(256) void sao_kernel(const DevFrame *__restrict__ f)
{
    const OhPicParams &pp = f->pp;
    const int c = blockIdx.z;
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    const int pw = f->cur.w[c], ph = f->cur.h[c];
    if (x >= pw || y >= ph)
        return;
    const int bd = pp.bit_depth, hs = hsh(pp, c), vs = vsh(pp, c), lc = pp.log2_ctb_size;
    const int ctbw = (pp.width + (1 << lc) - 1) >> lc, ctbh = (pp.height + (1 << lc) - 1) >> lc;
    const int sstride = f->cur.stride[c];
    const PX *__restrict__ src = (const PX *)f->cur.p[c];
    PX *__restrict__ dst = (PX *)f->out.p[c];
    const int cx = (x << hs) >> lc, cy = (y << vs) >> lc;
    const OhSaoCtb *s = &f->sao[cy * ctbw + cx];
    const int type = s->type_idx[c];
    const int v = src[(size_t)y * sstride + x];
    int r = v;
    if (type == 1) {                                        /* band, :340-365 */
        int k = ((v >> (bd - 5)) - s->band_position[c]) & 31;
        if (k < 4) r = clip_px(v + s->offset_val[c][k + 1], bd);
    } else if (type == 2) {                                 /* edge, :372-567 (per-sample form) */
        const int eo = s->eo_class[c];
        const int x0 = (cx << lc) >> hs, y0 = (cy << lc) >> vs;
        const int w = min((1 << lc) >> hs, pw - x0), h = min((1 << lc) >> vs, ph - y0);
        const int lx = x - x0, ly = y - y0;
        const int dx0 = eo == 1 ? 0 : (eo == 3 ? 1 : -1), dy0 = eo == 0 ? 0 : -1;
        const int flags = s->edge_flags;
        bool keep = false;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            int nx = lx + (k ? -dx0 : dx0), ny = ly + (k ? -dy0 : dy0);
            int rx = nx < 0 ? -1 : (nx >= w ? 1 : 0), ry = ny < 0 ? -1 : (ny >= h ? 1 : 0);
            if ((rx < 0 && cx == 0) || (rx > 0 && cx == ctbw - 1) || (ry < 0 && cy == 0) || (ry > 0 && cy == ctbh - 1))
                keep = true;
            else if (flags && (rx || ry)) {
                int bit;
                if (rx && ry) bit = 4 + (ry < 0 ? (rx < 0 ? 0 : 1) : (rx > 0 ? 2 : 3));
                else if (rx)  bit = rx > 0 ? 1 : 0;
                else          bit = ry > 0 ? 3 : 2;
                keep = keep || ((flags >> bit) & 1);
            }
        }
        if (!keep) {
            int a = src[(size_t)(y + dy0) * sstride + x + dx0], b = src[(size_t)(y - dy0) * sstride + x - dx0];
            int sum = (v > a) - (v < a) + (v > b) - (v < b);
            int cat = sum == 0 ? 0 : (sum == -2 ? 1 : (sum == -1 ? 2 : (sum == 1 ? 3 : 4)));
            r = clip_px(v + s->offset_val[c][cat], bd);
        }
    }
    if (type && f->is_pcm && (pp.transquant_bypass_enable || pp.pcm_loop_filter_disable)) {
        /* restore_tqb_pixels (hevc_filter.c:163-193) with its geometry quirks: the min-PU range is
         * derived from the CTB's LUMA origin plus the COMPONENT's size, and the row copy length
         * is (min_pu >> hshift) BYTES whatever the sample size. */
        const int l = pp.log2_min_pu_size, mpw = pp.width >> l;
        const int X0 = cx << lc, Y0 = cy << lc;
        const int wc = min((1 << lc) >> hs, pw - (X0 >> hs)), hc = min((1 << lc) >> vs, ph - (Y0 >> vs));
        const int px = (x << hs) >> l, py = (y << vs) >> l;
        if (px >= (X0 >> l) && px < ((X0 + wc) >> l) && py >= (Y0 >> l) && py < ((Y0 + hc) >> l) &&
            f->is_pcm[py * mpw + px]) {
            int sx = (px << l) >> hs;
            int len_samples = ((1 << l) >> hs) / (int)sizeof(PX);
            if (x - sx < len_samples)
                r = v;
        }
    }
    dst[(size_t)y * f->out.stride[c] + x] = (PX)r;
}

/* =========================================================================================
 * launchers
 * ======================================================================================= */
extern "C" int ohk_init(void)
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
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dct), m, sizeof(m)) == hipSuccess ? 0 : -1;
}

#define LAUNCH_BY_DEPTH(bd, kern, grid, block, stream, ...)                                   \
    do {                                                                                      \
        if ((bd) == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(kern<uint8_t>), grid, block, 0, stream, __VA_ARGS__);   \
        else           hipLaunchKernelGGL(HIP_KERNEL_NAME(kern<uint16_t>), grid, block, 0, stream, __VA_ARGS__);  \
    } while (0)

extern "C" void ohk_inter(const DevFrame *df, const OhPicParams *p, uint32_t n_tiles, hipStream_t st)
{
    if (!n_tiles) return;
    dim3 grid(n_tiles, p->chroma_format_idc ? 3 : 1);
    LAUNCH_BY_DEPTH(p->bit_depth, mc_kernel, grid, dim3(64), st, df);
}

extern "C" void ohk_residual(const DevFrame *df, const OhPicParams *p, uint32_t n_tu, hipStream_t st)
{
    if (!n_tu) return;
    LAUNCH_BY_DEPTH(p->bit_depth, residual_kernel, dim3(n_tu), dim3(64), st, df);
}

extern "C" void ohk_intra_level(const DevFrame *df, const OhPicParams *p, uint32_t first_ctu, uint32_t n_ctu, hipStream_t st)
{
    if (!n_ctu) return;
    LAUNCH_BY_DEPTH(p->bit_depth, intra_ctu_kernel, dim3(n_ctu), dim3(64 * INTRA_WAVES), st, df, first_ctu);
}

extern "C" void ohk_deblock(const DevFrame *df, const OhPicParams *p, int horiz, hipStream_t st)
{
    const int W = p->width, H = p->height;
    const int hs = p->chroma_format_idc == 1 || p->chroma_format_idc == 2, vs = p->chroma_format_idc == 1;
    if (!horiz) {
        dim3 g(W / 8 / 256 + 1, H / 4), gc(W / (8 << hs) / 256 + 1, (H + (4 << vs) - 1) / (4 << vs), 2);
        if (p->bit_depth == 8) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_luma_kernel<uint8_t, 0>), g, dim3(256), 0, st, df);
            if (p->chroma_format_idc) hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_chroma_kernel<uint8_t, 0>), gc, dim3(256), 0, st, df);
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_luma_kernel<uint16_t, 0>), g, dim3(256), 0, st, df);
            if (p->chroma_format_idc) hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_chroma_kernel<uint16_t, 0>), gc, dim3(256), 0, st, df);
        }
    } else {
        dim3 g(W / 4 / 256 + 1, H / 8), gc(W / (4 << hs) / 256 + 1, (H + (8 << vs) - 1) / (8 << vs), 2);
        if (p->bit_depth == 8) {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_luma_kernel<uint8_t, 1>), g, dim3(256), 0, st, df);
            if (p->chroma_format_idc) hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_chroma_kernel<uint8_t, 1>), gc, dim3(256), 0, st, df);
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_luma_kernel<uint16_t, 1>), g, dim3(256), 0, st, df);
            if (p->chroma_format_idc) hipLaunchKernelGGL(HIP_KERNEL_NAME(deblock_chroma_kernel<uint16_t, 1>), gc, dim3(256), 0, st, df);
        }
    }
}

extern "C" void ohk_sao(const DevFrame *df, const OhPicParams *p, hipStream_t st)
{
    dim3 grid((p->width + 255) / 256, p->height, p->chroma_format_idc ? 3 : 1);
    LAUNCH_BY_DEPTH(p->bit_depth, sao_kernel, grid, dim3(256), st, df);
}
