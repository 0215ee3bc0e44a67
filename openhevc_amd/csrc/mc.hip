/*
 * mc.hip — pass 1: inter prediction (luma qpel / chroma epel, weighted, uni/bi)
 * (gfx950; overview of the passes: kernels.hip; bit-exactness: tests/test_gpu_parity.py)
 */
#include "kernels_common.h"

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
static __device__ __forceinline__ int dot2(unsigned a, unsigned b, int c)
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b), c, false);
}
/* first term of a sum: the three-operand form with the constant 0 (the compiler would clear a register and use the
 * accumulating two-operand form) */
static __device__ __forceinline__ int dot2z(unsigned a, unsigned b)
{
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
/* low halves of two registers -> one pair: one byte permute */
static __device__ __forceinline__ unsigned pack2(int lo, int hi) { return __builtin_amdgcn_perm((unsigned)hi, (unsigned)lo, 0x05040100u); }

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
                        int acc;
                        if (j & 1) {
                            acc = dot2z(P[j >> 1], co[HT]);
#pragma unroll
                            for (int q = 1; q <= HT; q++) acc = dot2(P[(j >> 1) + q], co[HT + q], acc);
                        } else {
                            acc = dot2z(P[j >> 1], co[0]);
#pragma unroll
                            for (int q = 1; q < HT; q++) acc = dot2(P[(j >> 1) + q], co[q], acc);
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
            int e0 = dot2z(T[0].x, co[0]), e1 = dot2z(T[0].y, co[0]), o0 = dot2z(T[0].x, co[HT]), o1 = dot2z(T[0].y, co[HT]);
#pragma unroll
            for (int q = 1; q < HT; q++) { e0 = dot2(T[q].x, co[q], e0); e1 = dot2(T[q].y, co[q], e1); }
#pragma unroll
            for (int q = 1; q <= HT; q++) { o0 = dot2(T[q].x, co[HT + q], o0); o1 = dot2(T[q].y, co[HT + q], o1); }
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
 * launcher
 * ======================================================================================= */
/* mc_kernel's tap table: per fraction TAPS/2 pairs for even positions, then TAPS/2+1 pairs shifted by one tap
 * for odd positions; fraction 0 = unit filter, entry NFR = unit << (14 - bit_depth) (full-sample copy) */
int ohk_init_mc(void)
{
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
    return 0;
}

/* one launch per plane kind over a batch of n pictures of the geometry *p */
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
