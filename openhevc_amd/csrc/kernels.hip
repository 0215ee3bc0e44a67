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

/* ---- constant tables (H.265 facts; same numbers as hevc_filter.c:50-60, hevc_cabac.c:1417).  The interpolation taps and
 *      the transform bases are packed by ohk_init() (g_mctab, g_basis); the intra angles are resolved per block on the
 *      host (engine.hip, DevIntra) ---- */
__constant__ uint8_t c_level_scale[6] = { 40, 45, 51, 57, 64, 72 };        /* hevc_cabac.c:1417 */
__constant__ uint8_t c_tc[54] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
                                  5, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 22, 24 };
__constant__ uint8_t c_beta[52] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 20, 22, 24,
                                    26, 28, 30, 32, 34, 36, 38, 40, 42, 44, 46, 48, 50, 52, 54, 56, 58, 60, 62, 64 };
__constant__ uint8_t c_qpc[14] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };

/* Pointers read out of DevFrame are generic to the compiler, which then emits flat_* accesses;
 * those count on lgkmcnt as well as vmcnt, so every LDS wait would also wait for stores in flight.
 * All of them point to HBM: say so. */
#define GLOBAL __attribute__((address_space(1)))
#define G_CONST(T, p) ((const GLOBAL T *)(p))
#define G_MUT(T, p)   ((GLOBAL T *)(p))

/* struct load from HBM (C++ cannot copy-construct from an address-space-qualified lvalue) */
template <typename T>
static __device__ __forceinline__ T gload(const T *p)
{
    static_assert(sizeof(T) % 4 == 0, "dword-sized structs only");
    T out;
    const GLOBAL uint32_t *s = (const GLOBAL uint32_t *)p;
    uint32_t *d = (uint32_t *)&out;
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; i++)
        d[i] = s[i];
    return out;
}

typedef short short4v __attribute__((ext_vector_type(4)));
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
typedef unsigned int uint4v __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ int clip3(int v, int lo, int hi) { return min(max(v, lo), hi); }
static __device__ __forceinline__ int clip_px(int v, int bd) { return min(max(v, 0), (1 << bd) - 1); }
static __device__ __forceinline__ int clip16(int v) { return min(max(v, -32768), 32767); }
static __device__ __forceinline__ int hsh(const OhPicParams &p, int c) { return c && (p.chroma_format_idc == 1 || p.chroma_format_idc == 2); }
static __device__ __forceinline__ int vsh(const OhPicParams &p, int c) { return c && p.chroma_format_idc == 1; }

/* four consecutive samples as one 4-byte (8 bit) or 8-byte (>8 bit) access */
template <typename PX>
static __device__ __forceinline__ void load4(const GLOBAL PX *p, int v[4])
{
    if (sizeof(PX) == 1) {
        unsigned r = *(const GLOBAL unsigned *)p;
        v[0] = r & 0xff; v[1] = (r >> 8) & 0xff; v[2] = (r >> 16) & 0xff; v[3] = r >> 24;
    } else {
        uint2v r = *(const GLOBAL uint2v *)p;
        v[0] = r[0] & 0xffff; v[1] = r[0] >> 16; v[2] = r[1] & 0xffff; v[3] = r[1] >> 16;
    }
}
template <typename PX>
static __device__ __forceinline__ void store4(GLOBAL PX *p, int a, int b, int c, int d)
{
    if (sizeof(PX) == 1) {
        *(GLOBAL unsigned *)p = (unsigned)(a | (b << 8) | (c << 16) | (d << 24));
    } else {
        uint2v r = { (unsigned)(a | (b << 16)), (unsigned)(c | (d << 16)) };
        *(GLOBAL uint2v *)p = r;
    }
}


/* =========================================================================================
 * pass 1: inter prediction — hevcdsp_template.c:610-1609 through the drivers hevc.c:1641-1949;
 * picture-edge emulation (videodsp_template.c:26-101) is coordinate clamping while loading.
 *
 * The pass is bound by VALU issue (a wave64 instruction occupies its SIMD for 4 cycles), so the kernel
 * is built to spend few instructions per sample and to keep all 64 lanes busy whatever the PU size:
 *   - the unit of work is a <=8x8 block of one plane (DevMcJob); a wave runs four of them, 16 lanes each;
 *   - samples travel as 16-bit pairs in one dword and the taps are applied with v_dot2_i32_i16
 *     (2 multiply-adds per instruction).  A filter output at an odd position uses the taps shifted by
 *     one inside the pairs ((0,c0)(c1,c2)...(c7,0)), so no pair is ever re-aligned;
 *   - the h-pass lane owns 2 rows x 4 columns and writes its result as VERTICAL pairs, which is the
 *     operand layout the v-pass needs; the v-pass lane owns 2 x 2 outputs and stores them as pairs;
 *   - full-sample positions run through the same code with a unit filter (shift 0), which gives exactly
 *     the reference's copy / h-only / v-only variants (:610-700) without a divergent branch;
 *   - the windows of both lists are fetched before the first wait.
 * The h-pass result is kept as int16 exactly like the reference's tmp_array (:776).
 * ======================================================================================= */
/* packed tap pairs of every fraction, [luma / chroma][bit_depth - 8][McGeom::CS * (NFR + 1)]; filled by ohk_init() */
__device__ unsigned g_mctab[2][5][64];

template <int TAPS> struct McGeom {
    static constexpr int WROWS = 8 + TAPS;                 /* window rows kept: bh + TAPS - 1 <= WROWS - 1 */
    static constexpr int NSEG  = TAPS == 8 ? 4 : 3;        /* 4-sample segments per window row              */
    static constexpr int WP    = TAPS == 8 ? 10 : 6;       /* window row pitch, dwords (sample pairs)       */
    static constexpr int NIT   = WROWS / 4;                /* load steps: 4 rows x 4 segments per block     */
    static constexpr int NPD   = TAPS / 2 + 2;             /* pairs an h-pass lane reads per row            */
    static constexpr int NCO   = TAPS + 1;                 /* packed taps: TAPS/2 even-position + TAPS/2+1 odd-position pairs */
    static constexpr int CS    = TAPS + 2;                 /* pitch of one fraction in the tap table        */
    static constexpr int NFR   = TAPS == 8 ? 4 : 8;        /* fractions; entry NFR = unit << (14 - bit_depth) */
};

typedef short short2v __attribute__((ext_vector_type(2)));
typedef uint2v uint2v_a2 __attribute__((aligned(2)));
typedef unsigned unsigned_a1 __attribute__((aligned(1)));
static __device__ __forceinline__ int dot2(unsigned a, unsigned b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), c, false);
}
static __device__ __forceinline__ unsigned pack2(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }

/* four consecutive samples at any sample address -> two dwords of 16-bit pairs */
static __device__ __forceinline__ uint2v load4_pairs(const GLOBAL uint8_t *p)
{
    const unsigned b = *(const GLOBAL unsigned_a1 *)p;
    return uint2v{ __builtin_amdgcn_perm(0, b, 0x0c010c00), __builtin_amdgcn_perm(0, b, 0x0c030c02) };
}
static __device__ __forceinline__ uint2v load4_pairs(const GLOBAL uint16_t *p) { return *(const GLOBAL uint2v_a2 *)p; }

template <typename PX, int TAPS>
__global__ __launch_bounds__(64) void mc_kernel(const OhBatch B)
{
    typedef McGeom<TAPS> G;
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    constexpr bool LUMA = TAPS == 8;
    constexpr int before = TAPS / 2 - 1, HT = TAPS / 2;
    __shared__ __attribute__((aligned(16))) unsigned win[4][G::WROWS * G::WP + 8];   /* +8: the four blocks start on different banks */
    __shared__ __attribute__((aligned(16))) unsigned tmp[4][G::WROWS / 2 * 8];
    __shared__ __attribute__((aligned(8)))  unsigned ctab[(G::NFR + 1) * G::CS];
    __shared__ const void *refp[OH_MAX_REFS][2];
    const int lane = threadIdx.x, s = lane >> 4, sl = lane & 15;
    const OhPicParams &pp = f->pp;
    const int bd = pp.bit_depth;
    const int hs = LUMA ? 0 : hsh(pp, 1), vs = LUMA ? 0 : vsh(pp, 1);

    /* XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (b and b+8 share an L2), blocks
     * are listed in CTU / z-scan order.  Every XCD takes one CONTIGUOUS eighth of the list so that the
     * overlapping windows of neighbouring blocks hit the same L2. */
    const uint32_t nj = LUMA ? f->n_mc_luma : f->n_mc_chroma;
    const uint32_t nw = (nj + 3) >> 2, per = (nw + 7) >> 3;
    const uint32_t widx = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (widx >= nw)
        return;
    const uint32_t jidx = widx * 4 + s;
    const bool live = jidx < nj;                            /* a dead quarter repeats the last block and stores nothing */
    const DevMcJob job = gload((LUMA ? f->mc_luma : f->mc_chroma) + (live ? jidx : nj - 1));

    /* per-wave tables: reference plane pointers and the packed taps (built once by ohk_init() in the LDS layout);
     * both loads are issued before either is waited for */
    {
        const bool has_rp = lane < OH_MAX_REFS * (LUMA ? 1 : 2), has_ct = lane < (G::NFR + 1) * G::CS;
        const int r = LUMA ? lane : lane >> 1, pl = LUMA ? 0 : lane & 1;
        const void *rp = has_rp ? f->refs[r & (OH_MAX_REFS - 1)].p[LUMA ? 0 : 1 + pl] : nullptr;
        const unsigned ct = has_ct ? g_mctab[LUMA ? 0 : 1][bd - 8][lane] : 0u;
        if (has_rp) refp[r][pl] = rp;
        if (has_ct) ctab[lane] = ct;
    }
    const int pw = f->cur.w[LUMA ? 0 : 1], ph = f->cur.h[LUMA ? 0 : 1], stride = f->cur.stride[LUMA ? 0 : 1];
    const int bw = job.w, bh = job.h, wh = bh + TAPS - 1;
    const bool two = job.ref[1] != OH_NO_REF;
    const bool any_two = __builtin_amdgcn_ballot_w64(two) != 0;
    __syncthreads();

    /* window of one list -> registers: step `it` covers rows 4it..4it+3, lane = (row, 4-sample segment) */
    const int lrow = sl >> 2, seg = sl & 3;
    auto fetch = [&](const int l, const bool on, uint2v (&W)[G::NIT], int &fx, int &fy) {
        const int mvx = job.mv[l][0], mvy = job.mv[l][1];
        int ix, iy;
        if (LUMA) {
            fx = mvx & 3; fy = mvy & 3; ix = mvx >> 2; iy = mvy >> 2;
        } else {                                            /* hevc.c:1807-1813 */
            fx = (mvx & ((1 << (2 + hs)) - 1)) << (1 - hs);
            fy = (mvy & ((1 << (2 + vs)) - 1)) << (1 - vs);
            ix = mvx >> (2 + hs); iy = mvy >> (2 + vs);
        }
#pragma unroll
        for (int it = 0; it < G::NIT; it++) W[it] = uint2v{ 0, 0 };
        if (!on || seg >= G::NSEG)
            return;
        const GLOBAL PX *__restrict__ src = (const GLOBAL PX *)refp[job.ref[l]][LUMA ? 0 : job.c_idx - 1];
        const int gx = job.x + ix - before + 4 * seg, wy0 = job.y + iy - before;
        /* row offsets fit 24 bits x 24 bits (v_mul_u32_u24 is full rate); the in-picture test of a lane's
         * segment does not depend on the row, so it is taken once */
        if (gx >= 0 && gx + 3 < pw) {
            const GLOBAL PX *__restrict__ col = src + gx;
#pragma unroll
            for (int it = 0; it < G::NIT; it++) {
                const int row = 4 * it + lrow;
                if (row < wh)
                    W[it] = load4_pairs(col + __umul24(clip3(wy0 + row, 0, ph - 1), stride));
            }
        } else {
            const int x0 = clip3(gx, 0, pw - 1), x1 = clip3(gx + 1, 0, pw - 1), x2 = clip3(gx + 2, 0, pw - 1), x3 = clip3(gx + 3, 0, pw - 1);
#pragma unroll
            for (int it = 0; it < G::NIT; it++) {
                const int row = 4 * it + lrow;
                if (row >= wh)
                    continue;
                const GLOBAL PX *rowp = src + __umul24(clip3(wy0 + row, 0, ph - 1), stride);
                W[it] = uint2v{ pack2(rowp[x0], rowp[x1]), pack2(rowp[x2], rowp[x3]) };
            }
        }
    };
    /* one list: registers -> LDS window -> h-pass -> vertical pairs -> v-pass -> v[0..3] = (row 0: col 0, col 1; row 1: col 0, col 1) */
    auto filter = [&](const uint2v (&W)[G::NIT], const int fx, const int fy, int (&v)[4]) {
        if (seg < G::NSEG) {
#pragma unroll
            for (int it = 0; it < G::NIT; it++)
                *(uint2v *)&win[s][(4 * it + lrow) * G::WP + 2 * seg] = W[it];
        }
        __syncthreads();
        {
            const int i = sl >> 1, g = sl & 1;              /* row pair i, columns 4g..4g+3 */
            const int sh = fx ? bd - 8 : 0;
            unsigned co[G::NCO];
#pragma unroll
            for (int q = 0; q < G::NCO; q++) co[q] = ctab[fx * G::CS + q];
            if (i < G::WROWS / 2) {
                int o[2][4];
#pragma unroll
                for (int rr = 0; rr < 2; rr++) {
                    unsigned P[G::NPD];
#pragma unroll
                    for (int q = 0; q < G::NPD; q += 2) {
                        const uint2v t = *(const uint2v *)&win[s][(2 * i + rr) * G::WP + 2 * g + q];
                        P[q] = t.x; P[q + 1] = t.y;
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        int acc = 0;
                        if (j & 1) {
#pragma unroll
                            for (int q = 0; q <= HT; q++) acc = dot2(P[(j >> 1) + q], co[HT + q], acc);
                        } else {
#pragma unroll
                            for (int q = 0; q < HT; q++) acc = dot2(P[(j >> 1) + q], co[q], acc);
                        }
                        o[rr][j] = acc >> sh;
                    }
                }
                *(uint4v *)&tmp[s][i * 8 + 4 * g] = uint4v{ pack2(o[0][0], o[1][0]), pack2(o[0][1], o[1][1]), pack2(o[0][2], o[1][2]), pack2(o[0][3], o[1][3]) };
            }
        }
        __syncthreads();
        {
            const int cp = sl & 3, rp = sl >> 2;            /* columns 2cp, 2cp+1; rows 2rp, 2rp+1 */
            const int fr = fy ? fy : (fx ? 0 : G::NFR);
            const int sh = fy ? (fx ? 6 : bd - 8) : 0;
            unsigned co[G::NCO];
#pragma unroll
            for (int q = 0; q < G::NCO; q++) co[q] = ctab[fr * G::CS + q];
            uint2v T[HT + 1];
#pragma unroll
            for (int q = 0; q <= HT; q++) T[q] = *(const uint2v *)&tmp[s][(rp + q) * 8 + 2 * cp];
            int e0 = 0, e1 = 0, o0 = 0, o1 = 0;
#pragma unroll
            for (int q = 0; q < HT; q++) { e0 = dot2(T[q].x, co[q], e0); e1 = dot2(T[q].y, co[q], e1); }
#pragma unroll
            for (int q = 0; q <= HT; q++) { o0 = dot2(T[q].x, co[HT + q], o0); o1 = dot2(T[q].y, co[HT + q], o1); }
            v[0] = e0 >> sh; v[1] = e1 >> sh; v[2] = o0 >> sh; v[3] = o1 >> sh;
        }
    };

    uint2v WA[G::NIT], WB[G::NIT];
    int fxa, fya, fxb = 0, fyb = 0, va[4], vb[4] = { 0, 0, 0, 0 };
    fetch(0, true, WA, fxa, fya);
    if (any_two)
        fetch(1, two, WB, fxb, fyb);
    filter(WA, fxa, fya, va);
    if (any_two) {
        __syncthreads();                                    /* the v-pass of list 0 has read tmp */
        filter(WB, fxb, fyb, vb);
    }

    const int x = 2 * (sl & 3), y = 2 * (sl >> 2);
    if (!live || x >= bw || y >= bh)
        return;
    const int c = LUMA ? 0 : job.c_idx;
    const bool weighted = job.wp != OH_NO_WP;
    const bool from_l1 = job.flags & OH_MCF_FROM_L1;
    int w0 = 0, w1 = 0, o0 = 0, o1 = 0, denom = 0;
    if (weighted) {
        const OhWeights wp = gload(f->wp + job.wp);
        w0 = wp.w[0][c]; w1 = wp.w[1][c];
        o0 = wp.o[0][c] * (1 << (bd - 8)); o1 = wp.o[1][c] * (1 << (bd - 8));
        denom = wp.log2_denom[c ? 1 : 0];
        if (from_l1) { w0 = w1; o0 = o1; }
    }
    int r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (two) {
            const int a = (int16_t)va[j], b = vb[j];            /* list 0 went through an int16 tmp, hevc.c:1761 */
            if (!weighted) {
                const int shift = 15 - bd;
                r[j] = (b + a + (1 << (shift - 1))) >> shift;
            } else {
                const int log2wd = denom + 14 - bd;
                r[j] = (b * w1 + a * w0 + ((o0 + o1 + 1) << log2wd)) >> (log2wd + 1);
            }
        } else {
            const int a = va[j];
            if (!weighted) {
                const int shift = 14 - bd;
                r[j] = (a + (1 << (shift - 1))) >> shift;
            } else {
                const int shift = denom + 14 - bd;
                r[j] = ((a * w0 + (1 << (shift - 1))) >> shift) + o0;
            }
        }
        r[j] = clip_px(r[j], bd);
    }
    GLOBAL PX *__restrict__ dst = G_MUT(PX, f->cur.p[c]) + (size_t)(job.y + y) * stride + job.x + x;
    if (sizeof(PX) == 1) {
        *(GLOBAL uint16_t *)dst = (uint16_t)(r[0] | (r[1] << 8));
        *(GLOBAL uint16_t *)(dst + stride) = (uint16_t)(r[2] | (r[3] << 8));
    } else {
        *(GLOBAL unsigned *)dst = pack2(r[0], r[1]);
        *(GLOBAL unsigned *)(dst + stride) = pack2(r[2], r[3]);
    }
}

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
            const long long radd = 1ll << (shift - 1), scale = (long long)c_level_scale[qp % 6] << (qp / 6);
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
 * pass 3: intra prediction as a CTU wavefront — hevcpred_template.c:30-538
 * (constrained_intra_pred_flag == 0), each block followed by its residual (transform_add).
 *
 * One workgroup reconstructs one CTU.  The CTU's samples (with the one-sample border above and
 * to the left that intra_pred() gathers from, :164-183), the CTU's block descriptors and residual
 * blocks are staged in LDS once; the waves then take the blocks of the current SUB-LEVEL (blocks
 * of one sub-level never read each other).  Per block a wave
 *   - reads its 32-byte descriptor (everything that depends only on the block's geometry and mode
 *     was resolved on the host at upload: LDS offsets, edge sizes, filter / class flags, angles),
 *   - gathers left[]/top[] from the staged CTU, one element per lane, and substitutes missing
 *     samples with wave-uniform lane reads (v_readlane) instead of the reference's serial fills,
 *   - smooths with whole-wave DPP shifts, publishes left[]/top[] in LDS once,
 *   - predicts 4 consecutive samples per lane in a mode-class specific loop, adds the residual and
 *     writes LDS (for the next sub-level) and HBM (dword stores, never waited for).
 * Sub-levels are separated by an LDS-only workgroup barrier, so the dependent chain inside a CTU
 * costs a handful of LDS round trips per block inside one CU — no kernel launch, no HBM round
 * trip.  CTUs of one launch are mutually independent (same wavefront level, recorder.c).
 * ======================================================================================= */
#define INTRA_MAX_WAVES 8
/* diagnostic build (-DOH_STAMPS, tools/intra_stamps.py): in-kernel cycle accounting of workgroup 0 */
#ifdef OH_STAMPS
#define STAMP(var) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); var = t_; } while (0)
#define ACC(slot, t0, t1) (acc[slot] += (t1) - (t0))
#else
#define STAMP(var) do { } while (0)
#define ACC(slot, t0, t1) do { } while (0)
#endif
struct IntraLds { int E[OH_INTRA_WAVE_LDS / 4]; };                   /* E[0..65] = left[-1..64), E[66..131] = top[-1..64), per wave */
#define WSYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
/* workgroup barrier that waits for LDS traffic only (global stores of finished samples stay in flight) */
#define LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

/* lane i <- lane i-1 (lane 0 keeps `fill`) / lane i <- lane i+1 (lane 63 keeps `fill`): GFX9 whole-wave DPP shifts */
static __device__ __forceinline__ int wave_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false); }
static __device__ __forceinline__ int wave_shl1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x130, 0xf, 0xf, false); }

struct PlaneRegs { uint64_t base[3]; int stride[2]; };               /* wave-uniform (SGPRs): [0] luma, [1] chroma */

template <typename PX>
static __device__ __forceinline__ void put4(uint16_t *__restrict__ lds, GLOBAL PX *__restrict__ g, int v0, int v1, int v2, int v3)
{
    const uint2v pk16 = { (unsigned)(v0 | (v1 << 16)), (unsigned)(v2 | (v3 << 16)) };
    *(uint2v *)lds = pk16;                                           /* 8-byte aligned by construction */
    if (sizeof(PX) == 1) *(GLOBAL uint32_t *)g = v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
    else                 *(GLOBAL uint2v *)g = pk16;
}

/* constrained_intra_pred (hevcpred_template.c:185-286) for one block, run by ONE lane over the wave's edge arrays
 * left[k] = E[1 + k], top[k] = E[67 + k] (k = -1..63) after the gather with the re-derived candidate flags.
 * lm / tm: bit g = the 4-sample group g of the left column / top row lies in an intra CU; corner likewise. */
typedef __attribute__((address_space(3))) int lds_int;     /* keeps the accesses ds_* (a generic pointer would make them flat_*,
                                                              which are not ordered against ds_* of the same wave) */
static __device__ __forceinline__ void cip_patch(lds_int *E, const int n, const int avail, const unsigned lm, const unsigned tm, const bool corner_intra,
                                              const int size_max_x, const int size_max_y, const int bl_size,
                                              const bool x_nz, const bool y_nz, const int bd)
{
    lds_int *left = E + 1, *top = E + 67;
    bool a_bl = avail & OH_AV_BOTTOM_LEFT, a_l = avail & OH_AV_LEFT, a_ul = avail & OH_AV_UP_LEFT, a_u = avail & OH_AV_UP, a_ur = avail & OH_AV_UP_RIGHT;
    auto isl = [&](int j) { return j < 0 ? corner_intra : ((lm >> (j >> 2)) & 1) != 0; };
    auto ist = [&](int j) { return j < 0 ? corner_intra : ((tm >> (j >> 2)) & 1) != 0; };
    if (a_bl || a_l || a_ul || a_u || a_ur) {
        int j = n + bl_size - 1;
        if (a_bl || a_l || a_ul) {
            while (j > -1 && !isl(j)) j--;
            if (!isl(j)) {
                j = 0;
                while (j < size_max_x && !ist(j)) j++;
                for (int i = j; i > -1; i--)
                    if (!ist(i - 1)) top[i - 1] = top[i];
                left[-1] = top[-1];
            }
        } else {
            j = 0;
            while (j < size_max_x && !ist(j)) j++;
            if (j > 0) {
                for (int i = j; i > (x_nz ? -1 : 0); i--)
                    if (!ist(i - 1)) top[i - 1] = top[i];
                if (!x_nz) top[-1] = top[0];
            }
        }
        left[-1] = top[-1];
        if (a_bl || a_l) {
            int a = left[-1];
            for (int i = 0; i < size_max_y; i += 4) {
                if (!isl(i)) { left[i] = a; left[i + 1] = a; left[i + 2] = a; left[i + 3] = a; }
                else a = left[i + 3];
            }
        }
        if (!a_l)  for (int i = 0; i < n; i++) left[i] = left[-1];
        if (!a_bl) { const int v = left[n - 1]; for (int i = 0; i < n; i++) left[n + i] = v; }
        if (!x_nz) {
            for (int i = 0; i < size_max_y; i++) left[i] = 0;
        } else {
            int a = left[size_max_y - 1];
            for (int i = size_max_y - 1; i > -1; i -= 4) {
                if (!isl(i - 3)) { left[i - 3] = a; left[i - 2] = a; left[i - 1] = a; left[i] = a; }
                else a = left[i - 3];
            }
            if (y_nz && !corner_intra) left[-1] = left[0];
        }
        top[-1] = left[-1];
        if (y_nz) {
            int a = left[-1];
            for (int i = 0; i < size_max_x; i += 4) {
                if (!ist(i)) { top[i] = a; top[i + 1] = a; top[i + 2] = a; top[i + 3] = a; }
                else a = top[i + 3];
            }
        }
    }
    /* missing samples, :251-286 */
    if (!a_bl) {
        if (a_l) {
            const int v = left[n - 1];
            for (int i = 0; i < n; i++) left[n + i] = v;
        } else if (a_ul) {
            for (int i = 0; i < 2 * n; i++) left[i] = left[-1];
            a_l = true;
        } else if (a_u) {
            left[-1] = top[0];
            for (int i = 0; i < 2 * n; i++) left[i] = left[-1];
            a_ul = a_l = true;
        } else if (a_ur) {
            for (int i = 0; i < n; i++) top[i] = top[n];
            left[-1] = top[n];
            for (int i = 0; i < 2 * n; i++) left[i] = left[-1];
            a_u = a_ul = a_l = true;
        } else {
            left[-1] = 1 << (bd - 1);
            for (int i = 0; i < 2 * n; i++) { top[i] = left[-1]; left[i] = left[-1]; }
        }
    }
    if (!a_l)  { const int v = left[n]; for (int i = 0; i < n; i++) left[i] = v; }
    if (!a_ul) left[-1] = left[0];
    if (!a_u)  for (int i = 0; i < n; i++) top[i] = left[-1];
    if (!a_ur) { const int v = top[n - 1]; for (int i = 0; i < n; i++) top[n + i] = v; }
    top[-1] = left[-1];
}

template <typename PX, bool CIP, bool STAGED>
static __device__ __forceinline__ void intra_block(const DevFrame *__restrict__ f, const int bd, const PlaneRegs &pr,
                                                   const uint4v *__restrict__ item, IntraLds &s, uint16_t *__restrict__ M,
                                                   const int16_t *__restrict__ res_lds_base, const int lane, unsigned long long *acc)
{
    unsigned long long ta = 0, tb = 0, tc = 0, td = 0; (void)ta; (void)tb; (void)tc; (void)td; (void)acc;
    STAMP(ta);
    /* the descriptor is the same for every lane: keep it in scalar registers */
    const uint4v q0 = item[0], q1 = item[1];
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(q0[0]), w1 = __builtin_amdgcn_readfirstlane(q0[1]);
    const uint32_t res_off = __builtin_amdgcn_readfirstlane(q0[2]), w3 = __builtin_amdgcn_readfirstlane(q0[3]);
    const uint32_t w4 = __builtin_amdgcn_readfirstlane(q1[0]), w5 = __builtin_amdgcn_readfirstlane(q1[1]);
    const uint32_t res_lds = __builtin_amdgcn_readfirstlane(q1[2]);
    const int bx = w0 & 0xffff, by = w0 >> 16, c = w1 & 0xff, log2 = (w1 >> 8) & 0xff, avail = w1 >> 24;
    const int cm_off = w3 & 0xffff, top_off = w3 >> 16, rs = w4 & 0xffff, tr_size = (w4 >> 16) & 0xff, bl_size = w4 >> 24;
    const int angle = (int)(int8_t)(w5 & 0xff), flags = (w5 >> 8) & 0xff, inv_a = (int)(int16_t)(w5 >> 16);
    const int n = 1 << log2, cls = (flags >> 4) & 7;
    const uint32_t w7 = __builtin_amdgcn_readfirstlane(q1[3]);           /* cip_left | cip_top << 16 */
    const bool a_bl = avail & OH_AV_BOTTOM_LEFT, a_l = avail & OH_AV_LEFT, a_ul = avail & OH_AV_UP_LEFT;
    const bool a_u = avail & OH_AV_UP, a_ur = avail & OH_AV_UP_RIGHT;
    const int i = lane;                                                /* element this lane owns */
    const bool has_res = res_off != OH_NO_COEFF;
    const int ngroups = (n * n) >> 2;

    /* residual: requested now, consumed at the very end */
    short4v rv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        rv[k] = short4v{ 0, 0, 0, 0 };
        const int g = lane + 64 * k;
        if (has_res && g < ngroups) {
            if (STAGED) rv[k] = *(const short4v *)(res_lds_base + res_lds + 4 * g);
            else        rv[k] = *((const GLOBAL short4v *)(f->res + res_off) + g);       /* slow path: dependent HBM load */
        }
    }

    /* gather (:164-183) from the staged CTU: lane i owns top[i] and left[i]; branch-free addresses */
    int tv = 0, lv = 0, cv = 0;
    const bool t_ok = i < n ? a_u : (i < 2 * n && a_ur), l_ok = i < n ? a_l : (i < 2 * n && a_bl);
    {
        const int ti = i < n ? i : (i - n < tr_size ? i : n + tr_size - 1);
        const int li = i < n ? i : (i - n < bl_size ? i : n + bl_size - 1);
        if (t_ok) tv = M[top_off + ti];
        if (l_ok) lv = M[cm_off - 1 + li * rs];
        if (a_ul) cv = M[top_off - 1];
    }
    /* substitution (:251-286) in closed form: the reference's cascaded fills only ever copy one of
     * these wave-uniform values */
    const int l_0 = __builtin_amdgcn_readlane(lv, 0), l_n1 = __builtin_amdgcn_readlane(lv, n - 1), l_n = __builtin_amdgcn_readlane(lv, n & 63);
    const int t_0 = __builtin_amdgcn_readlane(tv, 0), t_n1 = __builtin_amdgcn_readlane(tv, n - 1), t_n = __builtin_amdgcn_readlane(tv, n & 63);
    int corner, left_i, top_i;
    if (CIP && (flags & OH_IF_CIP)) {
        /* constrained intra prediction (own kernel instantiation, so the common one carries none of this): rare, so one lane replays the reference's sweeps over the published edges
         * (cip_patch) instead of a lane-parallel closed form */
        const int fill = sizeof(PX) == 1 ? 128 : 0x8080;                   /* memset(.., 128, ..) over 16-bit samples, :158-160 */
        int *E = s.E;
        E[1 + i] = l_ok ? lv : fill;
        E[67 + i] = t_ok ? tv : fill;
        if (lane == 0) { E[0] = a_ul ? cv : 0; E[66] = a_ul ? cv : 128; }
        WSYNC();
        if (lane == 0)
            cip_patch((lds_int *)E, n, avail, w7 & 0xffff, w7 >> 16, (flags & OH_IF_CIP_CORNER) != 0, a_ur ? n + tr_size : n, a_bl ? n + bl_size : n,
                      a_bl ? bl_size : 0, bx != 0, by != 0, bd);
        WSYNC();
        left_i = E[1 + i]; top_i = E[67 + i]; corner = E[0];
    } else if (a_bl || a_l) {
        left_i = i < n ? (a_l ? lv : l_n) : (a_bl ? lv : l_n1);
        corner = a_ul ? cv : (a_l ? l_0 : l_n);
    } else {
        corner = a_ul ? cv : (a_u ? t_0 : (a_ur ? t_n : (1 << (bd - 1))));
        left_i = corner;
    }
    if (!(CIP && (flags & OH_IF_CIP)))
        top_i = i < n ? (a_u ? tv : corner) : (a_ur ? tv : (a_u ? t_n1 : corner));

    /* smoothing (:288-326) with whole-wave DPP shifts; the mode/size test was done on the host */
    if (flags & OH_IF_FILTER) {
        bool strong = false;
        int t63 = 0, l63 = 0;
        if (flags & OH_IF_STRONG_CAND) {
            t63 = __builtin_amdgcn_readlane(top_i, 63); l63 = __builtin_amdgcn_readlane(left_i, 63);
            const int t31 = __builtin_amdgcn_readlane(top_i, 31), l31 = __builtin_amdgcn_readlane(left_i, 31);
            const int lim = 1 << (bd - 5);
            strong = abs(corner + t63 - 2 * t31) < lim && abs(corner + l63 - 2 * l31) < lim;
        }
        if (strong) {
            if (i < 63) {
                top_i  = ((63 - i) * corner + (i + 1) * t63 + 32) >> 6;
                left_i = ((63 - i) * corner + (i + 1) * l63 + 32) >> 6;
            }
        } else {
            const int lp = wave_shr1(left_i, corner), ln = wave_shl1(left_i, 0);
            const int tp = wave_shr1(top_i, corner), tn = wave_shl1(top_i, 0);
            const int l0v = __builtin_amdgcn_readlane(left_i, 0), t0v = __builtin_amdgcn_readlane(top_i, 0);
            if (i < 2 * n - 1) {
                left_i = (ln + 2 * left_i + lp + 2) >> 2;
                top_i  = (tn + 2 * top_i + tp + 2) >> 2;
            }
            corner = (l0v + 2 * corner + t0v + 2) >> 2;
        }
    }
    /* publish the edges once */
    int *E = s.E;
    const int LB = 1, TB = 67;                            /* left[k] = E[LB + k], top[k] = E[TB + k] */
    if (i < 2 * n) { E[LB + i] = left_i; E[TB + i] = top_i; }
    if (lane == 0) { E[0] = corner; E[66] = corner; }
    STAMP(tb);
    WSYNC();
    STAMP(tc);

    /* prediction (:359-538): lane group g = lane + 64k owns samples 4g..4g+3 of the block (one row).
     * One straight-line loop per mode class; all LDS reads of a group are issued before use. */
    const bool edge = flags & OH_IF_EDGE;
    GLOBAL PX *__restrict__ dst = G_MUT(PX, c == 0 ? pr.base[0] : (c == 1 ? pr.base[1] : pr.base[2])) +
                                  (size_t)by * (c ? pr.stride[1] : pr.stride[0]) + bx;
    const int gstride = c ? pr.stride[1] : pr.stride[0];
    uint16_t *__restrict__ cm = M + cm_off;
#define GROUP_LOOP_BEGIN                                                                  \
    _Pragma("unroll") for (int k = 0; k < 4; k++) {                                       \
        const int g = lane + 64 * k;                                                      \
        if (g >= ngroups) break;                                                          \
        const int y = (4 * g) >> log2, x0 = (4 * g) & (n - 1);                            \
        int v[4];
#define GROUP_LOOP_END                                                                    \
        if (has_res) { _Pragma("unroll") for (int j = 0; j < 4; j++) v[j] = clip_px(v[j] + rv[k][j], bd); } \
        put4<PX>(cm + y * rs + x0 + 4 - 4, dst + (size_t)y * gstride + x0, v[0], v[1], v[2], v[3]);         \
    }
    /* note: cm already points at the block's sample (0,0) which sits at column index +4 of its row */
    if (cls == OH_IC_PLANAR) {
        const int tn_ = E[TB + n], ln_ = E[LB + n];
        GROUP_LOOP_BEGIN
            const int ly_ = E[LB + y];
            int tx[4];
#pragma unroll
            for (int j = 0; j < 4; j++) tx[j] = E[TB + x0 + j];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = x0 + j;
                v[j] = ((n - 1 - x) * ly_ + (x + 1) * tn_ + (n - 1 - y) * tx[j] + (y + 1) * ln_ + n) >> (log2 + 1);
            }
        GROUP_LOOP_END
    } else if (cls == OH_IC_DC) {
        int part = i < n ? left_i + top_i : 0;
        for (int m = 1; m < n; m <<= 1) part += __shfl_xor(part, m);
        const int dc = (__builtin_amdgcn_readlane(part, 0) + n) >> (log2 + 1);
        const int l0_ = E[LB], t0_ = E[TB];
        GROUP_LOOP_BEGIN
            const int ly_ = E[LB + y];
            int tx[4];
#pragma unroll
            for (int j = 0; j < 4; j++) tx[j] = E[TB + x0 + j];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = x0 + j;
                int pv = dc;
                if (edge) {                               /* :410-416 */
                    if (x == 0 && y == 0) pv = (l0_ + 2 * dc + t0_ + 2) >> 2;
                    else if (y == 0)      pv = (tx[j] + 3 * dc + 2) >> 2;
                    else if (x == 0)      pv = (ly_ + 3 * dc + 2) >> 2;
                }
                v[j] = pv;
            }
        GROUP_LOOP_END
    } else if (cls == OH_IC_PURE_V) {                     /* mode 26: copy of the row above, :474-477 */
        const int t0_ = E[TB], lm1 = E[LB - 1];
        GROUP_LOOP_BEGIN
            const int ly_ = E[LB + y];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = E[TB + x0 + j];
            if (edge && x0 == 0) v[0] = clip_px(t0_ + ((ly_ - lm1) >> 1), bd);
        GROUP_LOOP_END
    } else if (cls == OH_IC_PURE_H) {                     /* mode 10: copy of the left column, :501-508 */
        const int l0_ = E[LB], tm1 = E[TB - 1];
        GROUP_LOOP_BEGIN
            const int ly_ = E[LB + y];
            int tx[4];
#pragma unroll
            for (int j = 0; j < 4; j++) tx[j] = E[TB + x0 + j];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (edge && y == 0) ? clip_px(l0_ + ((tx[j] - tm1) >> 1), bd) : ly_;
        GROUP_LOOP_END
    } else if (cls == OH_IC_ANG_V) {                      /* modes 18..34 except 26: one (idx, fact) per row */
        GROUP_LOOP_BEGIN
            const int id = ((y + 1) * angle) >> 5, fact = ((y + 1) * angle) & 31;
            int r[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                /* ref[k] == top[k-1] for k >= 0, the projected left sample for k < 0 (:447-453) */
                const int kk = x0 + j + id + 1;
                r[j] = E[kk >= 0 ? TB + kk - 1 : LB - 1 + ((kk * inv_a + 128) >> 8)];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = fact ? ((32 - fact) * r[j] + fact * r[j + 1] + 16) >> 5 : r[j];
        GROUP_LOOP_END
    } else {                                              /* modes 2..17 except 10: one (idx, fact) per column */
        GROUP_LOOP_BEGIN
            int r0[4], r1[4], fact[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = x0 + j;
                const int id = ((x + 1) * angle) >> 5;
                fact[j] = ((x + 1) * angle) & 31;
                const int k0 = y + id + 1, k1 = k0 + 1;   /* ref[k] == left[k-1], projected top sample for k < 0 (:480-486) */
                r0[j] = E[k0 >= 0 ? LB + k0 - 1 : TB - 1 + ((k0 * inv_a + 128) >> 8)];
                r1[j] = E[k1 >= 0 ? LB + k1 - 1 : TB - 1 + ((k1 * inv_a + 128) >> 8)];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = fact[j] ? ((32 - fact[j]) * r0[j] + fact[j] * r1[j] + 16) >> 5 : r0[j];
        GROUP_LOOP_END
    }
#undef GROUP_LOOP_BEGIN
#undef GROUP_LOOP_END
    WSYNC();                                              /* this wave's edge arrays are reused by its next block */
    STAMP(td);
    ACC(2, ta, tb); ACC(3, tb, tc); ACC(4, tc, td);
}

/* Four blocks of <= 8x8 samples per wave, one per 16-lane slot (DPP row).  Same arithmetic as intra_block with the
 * block descriptor held per lane instead of in scalar registers: slot-local lane reads are ds_bpermute, the smoothing
 * shifts are DPP row shifts (they stop at the slot boundary by construction), the DC sum is an xor-butterfly inside the
 * slot.  Blocks of a sub-level are independent, so the four of a pass need no ordering.  edges: 4 x 36 ints,
 * per slot [0] = left[-1], [1..16] = left[0..15], [17] = top[-1], [18..33] = top[0..15]. */
static __device__ __forceinline__ int row_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false); }
static __device__ __forceinline__ int row_shl1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x101, 0xf, 0xf, false); }

template <typename PX>
static __device__ __forceinline__ void intra_slots(const int bd, const PlaneRegs &pr, const DevIntra *__restrict__ items, const uint32_t first,
                                                   const int count, int *__restrict__ edges, uint16_t *__restrict__ M,
                                                   const int16_t *__restrict__ res_lds_base, const int lane)
{
    const int slot = lane >> 4, sl = lane & 15, base = lane & 48;
    const bool act = slot < count;
    const uint4v *__restrict__ item = (const uint4v *)&items[first + (act ? slot : 0)];
    const uint4v q0 = item[0], q1 = item[1];
    const uint32_t w0 = q0[0], w1 = q0[1], res_off = q0[2], w3 = q0[3], w4 = q1[0], w5 = q1[1], res_lds = q1[2];
    const int bx = w0 & 0xffff, by = w0 >> 16, c = w1 & 0xff, log2 = (w1 >> 8) & 0xff, avail = w1 >> 24;
    const int cm_off = w3 & 0xffff, top_off = w3 >> 16, rs = w4 & 0xffff, tr_size = (w4 >> 16) & 0xff, bl_size = w4 >> 24;
    const int angle = (int)(int8_t)(w5 & 0xff), flags = (w5 >> 8) & 0xff, inv_a = (int)(int16_t)(w5 >> 16);
    const int n = 1 << log2, cls = (flags >> 4) & 7, ngroups = (n * n) >> 2;
    const bool a_bl = avail & OH_AV_BOTTOM_LEFT, a_l = avail & OH_AV_LEFT, a_ul = avail & OH_AV_UP_LEFT;
    const bool a_u = avail & OH_AV_UP, a_ur = avail & OH_AV_UP_RIGHT;
    const int i = sl;                                                  /* edge element this lane owns (2n <= 16) */
    const bool work = act && sl < ngroups;                             /* group this lane predicts */

    short4v rv = short4v{ 0, 0, 0, 0 };
    if (work && res_off != OH_NO_COEFF) rv = *(const short4v *)(res_lds_base + res_lds + 4 * sl);

    int tv = 0, lv = 0, cv = 0;
    {
        const bool t_ok = i < n ? a_u : (i < 2 * n && a_ur), l_ok = i < n ? a_l : (i < 2 * n && a_bl);
        const int ti = i < n ? i : (i - n < tr_size ? i : n + tr_size - 1);
        const int li = i < n ? i : (i - n < bl_size ? i : n + bl_size - 1);
        if (t_ok) tv = M[top_off + ti];
        if (l_ok) lv = M[cm_off - 1 + __mul24(li, rs)];
        if (a_ul) cv = M[top_off - 1];
    }
    const int l_0 = __shfl(lv, base), l_n1 = __shfl(lv, base + n - 1), l_n = __shfl(lv, base + (n & 15));
    const int t_0 = __shfl(tv, base), t_n1 = __shfl(tv, base + n - 1), t_n = __shfl(tv, base + (n & 15));
    int corner, left_i, top_i;
    if (a_bl || a_l) {
        left_i = i < n ? (a_l ? lv : l_n) : (a_bl ? lv : l_n1);
        corner = a_ul ? cv : (a_l ? l_0 : l_n);
    } else {
        corner = a_ul ? cv : (a_u ? t_0 : (a_ur ? t_n : (1 << (bd - 1))));
        left_i = corner;
    }
    top_i = i < n ? (a_u ? tv : corner) : (a_ur ? tv : (a_u ? t_n1 : corner));

    if (__builtin_amdgcn_ballot_w64(act && (flags & OH_IF_FILTER)) != 0) {
        /* smoothing (:288-326); never the strong filter here (32x32 only).  All lanes run the shifts, the flag selects. */
        const int lp = row_shr1(left_i, corner), ln = row_shl1(left_i, 0);
        const int tp = row_shr1(top_i, corner), tn = row_shl1(top_i, 0);
        const int l0v = __shfl(left_i, base), t0v = __shfl(top_i, base);
        if (flags & OH_IF_FILTER) {
            if (i < 2 * n - 1) {
                left_i = (ln + 2 * left_i + lp + 2) >> 2;
                top_i  = (tn + 2 * top_i + tp + 2) >> 2;
            }
            corner = (l0v + 2 * corner + t0v + 2) >> 2;
        }
    }
    int *E = edges + slot * 36;
    const int LB = 1, TB = 18;                               /* left[k] = E[LB + k], top[k] = E[TB + k] */
    if (act) {
        E[LB + i] = left_i; E[TB + i] = top_i;               /* entries >= 2n are written too and never read */
        if (sl == 0) { E[0] = corner; E[17] = corner; }
    }
    WSYNC();

    const int g = sl, y = (4 * g) >> log2, x0 = (4 * g) & (n - 1);
    const bool edge = flags & OH_IF_EDGE;
    int v[4] = { 0, 0, 0, 0 };
    /* the DC sum needs every lane of the slot: outside the per-group predicate (and skipped when no slot is DC) */
    int dc = 0;
    if (__builtin_amdgcn_ballot_w64(act && cls == OH_IC_DC) != 0) {
        int part = i < n ? left_i + top_i : 0;
        part += __shfl_xor(part, 1); part += __shfl_xor(part, 2); part += __shfl_xor(part, 4);
        dc = (__shfl(part, base) + n) >> (log2 + 1);
    }
    if (work) {
        if (cls == OH_IC_PLANAR) {
            const int tn_ = E[TB + n], ln_ = E[LB + n], ly_ = E[LB + y];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = x0 + j;
                v[j] = ((n - 1 - x) * ly_ + (x + 1) * tn_ + (n - 1 - y) * E[TB + x] + (y + 1) * ln_ + n) >> (log2 + 1);
            }
        } else if (cls == OH_IC_DC) {
            const int l0_ = E[LB], t0_ = E[TB], ly_ = E[LB + y];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = x0 + j;
                int pv = dc;
                if (edge) {
                    if (x == 0 && y == 0) pv = (l0_ + 2 * dc + t0_ + 2) >> 2;
                    else if (y == 0)      pv = (E[TB + x] + 3 * dc + 2) >> 2;
                    else if (x == 0)      pv = (ly_ + 3 * dc + 2) >> 2;
                }
                v[j] = pv;
            }
        } else if (cls == OH_IC_PURE_V) {
            const int t0_ = E[TB], lm1 = E[LB - 1], ly_ = E[LB + y];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = E[TB + x0 + j];
            if (edge && x0 == 0) v[0] = clip_px(t0_ + ((ly_ - lm1) >> 1), bd);
        } else if (cls == OH_IC_PURE_H) {
            const int l0_ = E[LB], tm1 = E[TB - 1], ly_ = E[LB + y];
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (edge && y == 0) ? clip_px(l0_ + ((E[TB + x0 + j] - tm1) >> 1), bd) : ly_;
        } else if (cls == OH_IC_ANG_V) {
            const int id = ((y + 1) * angle) >> 5, fact = ((y + 1) * angle) & 31;
            int r[5];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                const int kk = x0 + j + id + 1;
                r[j] = E[kk >= 0 ? TB + kk - 1 : LB - 1 + ((kk * inv_a + 128) >> 8)];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = fact ? ((32 - fact) * r[j] + fact * r[j + 1] + 16) >> 5 : r[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = x0 + j;
                const int id = ((x + 1) * angle) >> 5, fact = ((x + 1) * angle) & 31;
                const int k0 = y + id + 1, k1 = k0 + 1;
                const int r0 = E[k0 >= 0 ? LB + k0 - 1 : TB - 1 + ((k0 * inv_a + 128) >> 8)];
                const int r1 = E[k1 >= 0 ? LB + k1 - 1 : TB - 1 + ((k1 * inv_a + 128) >> 8)];
                v[j] = fact ? ((32 - fact) * r0 + fact * r1 + 16) >> 5 : r0;
            }
        }
        if (res_off != OH_NO_COEFF) {
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = clip_px(v[j] + rv[j], bd);
        }
        const int gstride = c ? pr.stride[1] : pr.stride[0];
        GLOBAL PX *__restrict__ dst = G_MUT(PX, c == 0 ? pr.base[0] : (c == 1 ? pr.base[1] : pr.base[2])) + (size_t)(by + y) * gstride + bx + x0;
        put4<PX>(M + cm_off + y * rs + x0, dst, v[0], v[1], v[2], v[3]);
    }
    WSYNC();                                                 /* the wave's edge arrays are reused by its next pass */
}

template <typename PX, bool CIP, bool STAGED>
__global__ __launch_bounds__(64 * INTRA_MAX_WAVES) void intra_ctu_kernel(const OhBatch B, const OhIntraLaunch L)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    const uint32_t first_ctu = f->lvl_start[L.level];
    if (blockIdx.x >= f->lvl_start[L.level + 1] - first_ctu)
        return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint16_t *__restrict__ M = (uint16_t *)smem;                                  /* sample area, oh_ctu_areas() */
    DevIntra *__restrict__ items = (DevIntra *)(smem + L.off_items);
    uint32_t *__restrict__ sub = (uint32_t *)(smem + L.off_sub);
    uint32_t *__restrict__ small = (uint32_t *)(smem + L.off_small);              /* per sub-level: leading blocks that go four per wave */
    int16_t *__restrict__ res_l = (int16_t *)(smem + L.off_res);                  /* the CTU's residual blocks (DevIntraCtu.res_lo/res_cnt) */
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nwaves = nthr >> 6;
    IntraLds &edges = *(IntraLds *)(smem + L.off_wave + wave * OH_INTRA_WAVE_LDS);
    const DevIntraCtu ctu = gload(f->ictu + first_ctu + blockIdx.x);
    const GLOBAL uint32_t *__restrict__ ss = G_CONST(uint32_t, f->sub_start) + ctu.sub_first;
    const OhPicParams &pp = f->pp;
    const int lc = pp.log2_ctb_size, ctbw = (pp.width + (1 << lc) - 1) >> lc;
    const int cx0 = (ctu.ctu % ctbw) << lc, cy0 = (ctu.ctu / ctbw) << lc;      /* luma origin of the CTU */
    const int n_sub = min((int)ctu.n_sub, OH_MAX_CTU_BLOCKS);
    const int bd = pp.bit_depth;
    const OhCtuAreas ar = oh_ctu_areas(lc, pp.chroma_format_idc);
    PlaneRegs pr;
    pr.base[0] = (uint64_t)f->cur.p[0]; pr.base[1] = (uint64_t)f->cur.p[1]; pr.base[2] = (uint64_t)f->cur.p[2];
    pr.stride[0] = f->cur.stride[0]; pr.stride[1] = f->cur.stride[1];

    /* stage: block descriptors, sub-level table, residual blocks */
    const uint32_t item0 = ctu.item0, n_items = min(ctu.n_items, (uint32_t)OH_MAX_CTU_BLOCKS);
    {
        const GLOBAL uint4v *__restrict__ src = (const GLOBAL uint4v *)(f->intra + item0);
        uint4v *dst = (uint4v *)items;
        for (uint32_t e = tid; e < n_items * 2; e += nthr) dst[e] = src[e];
        for (int e = tid; e <= n_sub; e += nthr) sub[e] = ss[e] - item0;
        for (int e = tid; e < n_sub; e += nthr) small[e] = G_CONST(uint32_t, f->sub_small)[ctu.sub_first + e];
        /* the CTU's residual blocks: one coalesced sweep instead of a dependent load per block */
        const GLOBAL short4v *__restrict__ rsrc = (const GLOBAL short4v *)(f->res + ctu.res_lo);
        if (STAGED)
            for (uint32_t e = tid; e < ctu.res_cnt / 4; e += nthr) ((short4v *)res_l)[e] = rsrc[e];
    }
    /* stage the part of the CTU its blocks read (DevIntraCtu.bx0..by1): samples reconstructed by passes
     * 1-2 (inter), the column left of the CTU and the row above it (up to 2*wc samples: the up-right CTU) —
     * all final by the wavefront order.  Rows go as 4-sample vectors, 16 per row and step. */
    const int nplanes = pp.chroma_format_idc ? 3 : 1;
    for (int c = 0; c < nplanes; c++) {
        const int hs = hsh(pp, c), vs = vsh(pp, c);
        const int wc = (1 << lc) >> hs, hc = (1 << lc) >> vs, rs = wc + 4;
        const int x0 = cx0 >> hs, y0 = cy0 >> vs, pw = f->cur.w[c], ph = f->cur.h[c], stride = f->cur.stride[c];
        const int px0 = ctu.bx0 >> hs, px1 = (ctu.bx1 + (1 << hs) - 1) >> hs, py0 = ctu.by0 >> vs, py1 = (ctu.by1 + (1 << vs) - 1) >> vs;
        const GLOBAL PX *__restrict__ g = G_CONST(PX, f->cur.p[c]);
        uint16_t *__restrict__ Mm = M + (c == 0 ? ar.main[0] : c == 1 ? ar.main[1] : ar.main[2]);      /* selects: no indexed struct on the stack */
        uint16_t *__restrict__ Mt = M + (c == 0 ? ar.top[0] : c == 1 ? ar.top[1] : ar.top[2]);
        const int r0 = max(py0, 0), r1 = min(min(py1, hc), ph - y0);
        const int cs = max(px0, 0) & ~3, ce = min(min((px1 + 3) & ~3, wc), pw - x0);
        const int seg = tid & 15;
        if (cs + 4 * seg < ce)
            for (int row = r0 + (tid >> 4); row < r1; row += nthr >> 4)
                *(uint2v *)&Mm[row * rs + cs + 4 * seg + 4] = load4_pairs(g + (size_t)(y0 + row) * stride + x0 + cs + 4 * seg);
        if (px0 < 0 && x0 > 0)
            for (int row = r0 + tid; row < r1; row += nthr)
                Mm[row * rs + 3] = g[(size_t)(y0 + row) * stride + x0 - 1];
        if (py0 < 0 && y0 > 0) {
            const int t0 = max(px0, x0 > 0 ? -1 : 0), t1 = min(min(px1, 2 * wc), pw - x0);
            for (int xx = t0 + tid; xx < t1; xx += nthr)
                Mt[xx + 4] = g[(size_t)(y0 - 1) * stride + x0 + xx];
        }
    }
    __syncthreads();
    unsigned long long acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, t0 = 0, t1 = 0, t2 = 0, tk = 0, rt1 = 0; (void)acc; (void)t0; (void)t1; (void)t2; (void)tk; (void)rt1;
#ifdef OH_STAMPS
    STAMP(tk);
    rt1 = __builtin_amdgcn_s_memrealtime();
#endif

    for (int s = 0; s < n_sub; s++) {
        STAMP(t0);
        /* units of the sub-level: groups of up to four <=8x8 blocks (one 16-lane slot each), then the bigger blocks one
         * per wave; the slot path needs the residual staged in LDS and has no constrained-intra variant */
        const uint32_t b0 = sub[s], b1 = sub[s + 1], ns = (STAGED && !CIP) ? min(small[s], b1 - b0) : 0u;
        const uint32_t ngrp = (ns + 3) >> 2, nunits = ngrp + (b1 - b0 - ns);
        for (uint32_t u = wave; u < nunits; u += nwaves) {
            if (u < ngrp)
                intra_slots<PX>(bd, pr, items, b0 + 4 * u, (int)min(4u, ns - 4 * u), edges.E, M, res_l, lane);
            else
                intra_block<PX, CIP, STAGED>(f, bd, pr, (const uint4v *)&items[b0 + ns + (u - ngrp)], edges, M, res_l, lane, acc);
        }
        STAMP(t1);
        LDS_BARRIER();                                    /* next sub-level reads what this one wrote to LDS */
        STAMP(t2);
        ACC(0, t0, t1); ACC(1, t1, t2);
    }
#ifdef OH_STAMPS
    if (f->dbg && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && lane == 0) {
        unsigned long long te; STAMP(te);
        unsigned long long rt2 = __builtin_amdgcn_s_memrealtime();
        unsigned long long slot = atomicAdd((unsigned long long *)f->dbg, 1ull);
        if (slot < 4000) {
            unsigned long long *o = (unsigned long long *)f->dbg + 16 + slot * 16;
            o[0] = n_sub; o[1] = te - tk; o[2] = rt2 - rt1; o[3] = acc[0]; o[4] = acc[1]; o[5] = acc[2]; o[6] = acc[3]; o[7] = acc[4];
            o[8] = gridDim.x; o[9] = n_items; o[10] = tk;
        }
    }
#endif
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
 * pass 5: SAO — hevcdsp_template.c:340-567 driven per CTB by sao_filter_CTB (hevc_filter.c:197-322),
 * here one whole-picture pass from the deblocked planes (cur) into the output planes (out).
 * One lane owns 8 consecutive samples of a row (8- or 16-byte accesses); a group never straddles a
 * CTB (CTB widths are multiples of 8 samples in every plane).  Rows are 256-byte aligned and padded,
 * so whole-vector accesses past the picture width stay inside the row.
 * ======================================================================================= */
template <typename PX> struct Vec8;
template <> struct Vec8<uint8_t>  { typedef unsigned int  T __attribute__((ext_vector_type(2))); };
template <> struct Vec8<uint16_t> { typedef unsigned int  T __attribute__((ext_vector_type(4))); };

template <typename PX>
static __device__ __forceinline__ void load8(const GLOBAL PX *p, int v[8])
{
    typename Vec8<PX>::T r = *(const GLOBAL typename Vec8<PX>::T *)p;
    if (sizeof(PX) == 1) {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (r[j >> 2] >> (8 * (j & 3))) & 0xff;
    } else {
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (r[j >> 1] >> (16 * (j & 1))) & 0xffff;
    }
}
template <typename PX>
static __device__ __forceinline__ void store8(GLOBAL PX *p, const int v[8])
{
    typename Vec8<PX>::T r;
    if (sizeof(PX) == 1) {
        r[0] = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
        r[1] = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
    } else {
#pragma unroll
        for (int j = 0; j < 4; j++) r[j] = v[2 * j] | (v[2 * j + 1] << 16);
    }
    __builtin_nontemporal_store(r, (GLOBAL typename Vec8<PX>::T *)p);     /* the output is next read by another picture's MC: stream it past the L2 */
}

struct SaoEdgeCtx { int x, y, x0, y0, w, h, pw, ph, sstride, cx, cy, ctbw, ctbh, flags, bd; };

/* first neighbour a = (x+DX, y+DY), second b = (x-DX, y-DY) (pos[][] of hevcdsp_template.c:379-384) */
template <typename PX, int DX, int DY>
static __device__ __forceinline__ void sao_edge8(const GLOBAL PX *__restrict__ src, const SaoEdgeCtx &e, const int off[5], const int v[8], int r[8])
{
    int a[10], b[10];                                       /* samples x-1..x+8 of rows y+DY and y-DY */
    const int ya = min(max(e.y + DY, 0), e.ph - 1), yb = min(max(e.y - DY, 0), e.ph - 1);
    load8<PX>(src + (size_t)ya * e.sstride + e.x, a + 1);
    load8<PX>(src + (size_t)yb * e.sstride + e.x, b + 1);
    a[0] = b[0] = a[9] = b[9] = 0;
    if (DX != 0) {
        if (e.x > 0)        { a[0] = src[(size_t)ya * e.sstride + e.x - 1]; b[0] = src[(size_t)yb * e.sstride + e.x - 1]; }
        if (e.x + 8 < e.pw) { a[9] = src[(size_t)ya * e.sstride + e.x + 8]; b[9] = src[(size_t)yb * e.sstride + e.x + 8]; }
    }
    const int ly = e.y - e.y0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int lx = e.x + j - e.x0;
        bool keep = e.x + j >= e.pw;
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int nx = lx + (k ? -DX : DX), ny = ly + (k ? -DY : DY);
            const int rx = nx < 0 ? -1 : (nx >= e.w ? 1 : 0), ry = ny < 0 ? -1 : (ny >= e.h ? 1 : 0);
            if ((rx < 0 && e.cx == 0) || (rx > 0 && e.cx == e.ctbw - 1) || (ry < 0 && e.cy == 0) || (ry > 0 && e.cy == e.ctbh - 1))
                keep = true;
            else if (e.flags && (rx || ry)) {
                int bit;
                if (rx && ry) bit = 4 + (ry < 0 ? (rx < 0 ? 0 : 1) : (rx > 0 ? 2 : 3));
                else if (rx)  bit = rx > 0 ? 1 : 0;
                else          bit = ry > 0 ? 3 : 2;
                keep = keep || ((e.flags >> bit) & 1);
            }
        }
        if (!keep) {
            const int na = a[1 + j + DX], nb = b[1 + j - DX];
            const int sum = (v[j] > na) - (v[j] < na) + (v[j] > nb) - (v[j] < nb);
            const int o = sum == 0 ? off[0] : (sum == -2 ? off[1] : (sum == -1 ? off[2] : (sum == 1 ? off[3] : off[4])));
            r[j] = clip_px(v[j] + o, e.bd);
        }
    }
}

/* Lane -> sample mapping: a wave covers ONE CTB-wide strip (wc samples x 512/wc rows) so that the
 * SAO type / class is the same for all its lanes (no divergent band/edge paths); a workgroup of
 * 4 waves covers 4 such strips stacked vertically.  grid = (CTB columns, strips of rows, planes). */
template <typename PX>
__global__ __launch_bounds__(256) void sao_kernel(const OhBatch B, const int nplanes_)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.z / nplanes_];
    const OhPicParams &pp = f->pp;
    const int c = blockIdx.z % nplanes_;
    const int pw = f->cur.w[c], ph = f->cur.h[c];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int log2_gx = pp.log2_ctb_size - hsh(pp, c) - 3;            /* 8-sample groups per CTB row: 1 << log2_gx */
    const int rows_per_wave = 64 >> log2_gx;
    const int x = (blockIdx.x << (log2_gx + 3)) + ((lane & ((1 << log2_gx) - 1)) << 3);
    const int y = (blockIdx.y * 4 + wave) * rows_per_wave + (lane >> log2_gx);
    if (x >= pw || y >= ph)
        return;
    const int bd = pp.bit_depth, hs = hsh(pp, c), vs = vsh(pp, c), lc = pp.log2_ctb_size;
    const int ctbw = (pp.width + (1 << lc) - 1) >> lc, ctbh = (pp.height + (1 << lc) - 1) >> lc;
    const int sstride = f->cur.stride[c];
    const GLOBAL PX *__restrict__ src = G_CONST(PX, f->cur.p[c]);
    GLOBAL PX *__restrict__ dst = G_MUT(PX, f->out.p[c]) + (size_t)y * f->out.stride[c] + x;
    const int cx = (x << hs) >> lc, cy = (y << vs) >> lc;
    const GLOBAL OhSaoCtb *s = G_CONST(OhSaoCtb, f->sao) + cy * ctbw + cx;
    const int type = s->type_idx[c];
    int v[8], r[8];
    load8<PX>(src + (size_t)y * sstride + x, v);
#pragma unroll
    for (int j = 0; j < 8; j++) r[j] = v[j];
    if (type == 1) {                                        /* band, :340-365 */
        const int bp = s->band_position[c];
        int off[4];
#pragma unroll
        for (int k = 0; k < 4; k++) off[k] = s->offset_val[c][k + 1];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            int k = ((v[j] >> (bd - 5)) - bp) & 31;
            int o = k == 0 ? off[0] : (k == 1 ? off[1] : (k == 2 ? off[2] : off[3]));
            if (k < 4) r[j] = clip_px(v[j] + o, bd);
        }
    } else if (type == 2) {                                 /* edge, :372-567 (per-sample form) */
        const int eo = s->eo_class[c];
        const int x0 = (cx << lc) >> hs, y0 = (cy << lc) >> vs;
        const int w = min((1 << lc) >> hs, pw - x0), h = min((1 << lc) >> vs, ph - y0);
        const int flags = s->edge_flags;
        int off[5];
#pragma unroll
        for (int k = 0; k < 5; k++) off[k] = s->offset_val[c][k];
        const SaoEdgeCtx ec = { x, y, x0, y0, w, h, pw, ph, sstride, cx, cy, ctbw, ctbh, flags, bd };
        switch (eo) {                                       /* compile-time neighbour offsets: no indexed registers */
        case 0:  sao_edge8<PX, -1, 0>(src, ec, off, v, r); break;
        case 1:  sao_edge8<PX, 0, -1>(src, ec, off, v, r); break;
        case 2:  sao_edge8<PX, -1, -1>(src, ec, off, v, r); break;
        default: sao_edge8<PX, 1, -1>(src, ec, off, v, r); break;
        }
    }
    if (type && f->is_pcm && (pp.transquant_bypass_enable || pp.pcm_loop_filter_disable)) {
        /* restore_tqb_pixels (hevc_filter.c:163-193) with its geometry quirks: the min-PU range is
         * derived from the CTB's LUMA origin plus the COMPONENT's size, and the row copy length
         * is (min_pu >> hshift) BYTES whatever the sample size. */
        const int l = pp.log2_min_pu_size, mpw = pp.width >> l;
        const int X0 = cx << lc, Y0 = cy << lc;
        const int wc = min((1 << lc) >> hs, pw - (X0 >> hs)), hc = min((1 << lc) >> vs, ph - (Y0 >> vs));
        const int py = (y << vs) >> l;
        const GLOBAL uint8_t *pcm = G_CONST(uint8_t, f->is_pcm);
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int px = ((x + j) << hs) >> l;
            if (px >= (X0 >> l) && px < ((X0 + wc) >> l) && py >= (Y0 >> l) && py < ((Y0 + hc) >> l) && pcm[py * mpw + px]) {
                int sx = (px << l) >> hs;
                int len_samples = ((1 << l) >> hs) / (int)sizeof(PX);
                if (x + j - sx < len_samples)
                    r[j] = v[j];
            }
        }
    }
    store8<PX>(dst, r);
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
    {   /* mc_kernel's tap table: per fraction TAPS/2 pairs for even positions, then TAPS/2+1 pairs shifted by one tap
         * for odd positions; fraction 0 = unit filter, entry NFR = unit << (14 - bit_depth) (full-sample copy) */
        static const int8_t qpel[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
        static const int8_t epel[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
        static unsigned tab[2][5][64];
        for (int luma = 1; luma >= 0; luma--) {
            const int taps = luma ? 8 : 4, ht = taps / 2, nco = taps + 1, cs = taps + 2, nfr = luma ? 4 : 8, before = ht - 1;
            for (int bd = 8; bd <= 12; bd++)
                for (int fr = 0; fr <= nfr; fr++)
                    for (int q = 0; q < nco; q++) {
                        int k[2];
                        if (q < ht) { k[0] = 2 * q; k[1] = 2 * q + 1; } else { k[0] = 2 * (q - ht) - 1; k[1] = 2 * (q - ht); }
                        int v[2];
                        for (int j = 0; j < 2; j++) {
                            if (k[j] < 0 || k[j] >= taps) v[j] = 0;
                            else if (fr == 0) v[j] = k[j] == before;
                            else if (fr == nfr) v[j] = k[j] == before ? 1 << (14 - bd) : 0;
                            else v[j] = luma ? qpel[fr][k[j]] : epel[fr][k[j]];
                        }
                        tab[luma ? 0 : 1][bd - 8][fr * cs + q] = ((unsigned)v[0] & 0xffffu) | ((unsigned)v[1] << 16);
                    }
        }
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_mctab), tab, sizeof(tab)) != hipSuccess)
            return -1;
    }
    /* the intra kernel's LDS block is sized per launch and exceeds 64 KiB for 4:4:4 CTUs full of 4x4 blocks */
    const int max_lds = 128 * 1024;
    const void *intra_kernels[8] = {
        (const void *)intra_ctu_kernel<uint8_t, false, false>, (const void *)intra_ctu_kernel<uint8_t, false, true>,
        (const void *)intra_ctu_kernel<uint8_t, true, false>, (const void *)intra_ctu_kernel<uint8_t, true, true>,
        (const void *)intra_ctu_kernel<uint16_t, false, false>, (const void *)intra_ctu_kernel<uint16_t, false, true>,
        (const void *)intra_ctu_kernel<uint16_t, true, false>, (const void *)intra_ctu_kernel<uint16_t, true, true> };
    for (const void *k : intra_kernels)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess)
            return -1;
    return 0;
}

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

/* launchers: one launch per pass over a batch of n pictures of the geometry *p */
extern "C" void ohk_inter(const OhBatch *B, int n, const OhPicParams *p, uint32_t max_luma, uint32_t max_chroma, hipStream_t st)
{
    /* four blocks per wave; the grid is 8 contiguous slices of the block list, one per XCD */
    const unsigned gl = ((((max_luma + 3) >> 2) + 7) >> 3) * 8, gc = ((((max_chroma + 3) >> 2) + 7) >> 3) * 8;
    if (p->bit_depth == 8) {
        if (max_luma) hipLaunchKernelGGL(HIP_KERNEL_NAME(mc_kernel<uint8_t, 8>), dim3(gl, n), dim3(64), 0, st, *B);
        if (max_chroma) hipLaunchKernelGGL(HIP_KERNEL_NAME(mc_kernel<uint8_t, 4>), dim3(gc, n), dim3(64), 0, st, *B);
    } else {
        if (max_luma) hipLaunchKernelGGL(HIP_KERNEL_NAME(mc_kernel<uint16_t, 8>), dim3(gl, n), dim3(64), 0, st, *B);
        if (max_chroma) hipLaunchKernelGGL(HIP_KERNEL_NAME(mc_kernel<uint16_t, 4>), dim3(gc, n), dim3(64), 0, st, *B);
    }
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

extern "C" void ohk_intra_level(const OhBatch *B, int n, const OhPicParams *p, const OhIntraLaunch *l, uint32_t max_ctu, hipStream_t st)
{
    if (!max_ctu) return;
    dim3 g(max_ctu, n), b(64 * l->waves);
    /* instantiations: constrained intra pred carries a slow path the common one must not pay for; STAGED = every CTU of the
     * launch has its residual span in LDS (otherwise the blocks read it from HBM) */
#define INTRA_LAUNCH(PX, CIP, ST) hipLaunchKernelGGL(HIP_KERNEL_NAME(intra_ctu_kernel<PX, CIP, ST>), g, b, l->lds_bytes, st, *B, *l)
#define INTRA_BY_FLAGS(PX)                                                                     \
    do {                                                                                       \
        if (p->constrained_intra_pred) { if (l->staged) INTRA_LAUNCH(PX, true, true); else INTRA_LAUNCH(PX, true, false); }   \
        else                           { if (l->staged) INTRA_LAUNCH(PX, false, true); else INTRA_LAUNCH(PX, false, false); } \
    } while (0)
    if (p->bit_depth == 8) INTRA_BY_FLAGS(uint8_t); else INTRA_BY_FLAGS(uint16_t);
#undef INTRA_BY_FLAGS
#undef INTRA_LAUNCH
}

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

extern "C" void ohk_sao(const OhBatch *B, int n, const OhPicParams *p, hipStream_t st)
{
    /* luma geometry decides the grid; chroma planes (smaller) leave their surplus workgroups idle */
    const int ctbw = (p->width + (1 << p->log2_ctb_size) - 1) >> p->log2_ctb_size;
    const int rows_per_block = 4 * (64 >> (p->log2_ctb_size - 3));            /* luma: 4 waves x (512 / ctb) rows */
    const int np = p->chroma_format_idc ? 3 : 1;
    dim3 grid(ctbw, (p->height + rows_per_block - 1) / rows_per_block, np * n);
    if (p->bit_depth == 8) hipLaunchKernelGGL(HIP_KERNEL_NAME(sao_kernel<uint8_t>), grid, dim3(256), 0, st, *B, np);
    else                   hipLaunchKernelGGL(HIP_KERNEL_NAME(sao_kernel<uint16_t>), grid, dim3(256), 0, st, *B, np);
}
