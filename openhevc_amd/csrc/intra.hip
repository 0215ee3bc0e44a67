/*
 * intra.hip — pass 3: intra prediction as a CTU wavefront (planar / DC / angular, constrained intra pred) + deferred residual add
 * (gfx950; overview of the passes: kernels.hip; bit-exactness: tests/test_gpu_parity.py)
 */
#include "kernels_common.h"

/* =========================================================================================
 * pass 3: intra prediction as a CTU wavefront — hevcpred_template.c:30-538
 * (constrained_intra_pred_flag == 0), each block followed by its residual (transform_add).
 *
 * One workgroup reconstructs one CTU.  The CTU's samples (with the one-sample border above and
 * to the left that intra_pred() gathers from, :164-183), the CTU's block descriptors and residual
 * blocks are staged in LDS once; the waves then take the blocks of the current SUB-LEVEL (blocks
 * of one sub-level never read each other).  Blocks of up to 8x8 samples go four per wave (slots_prepare /
 * slots_finish below: the sample-independent half runs one sub-level ahead on other waves); a bigger block has a
 * wave to itself (intra_block), which
 *   - reads its 32-byte descriptor (everything that depends only on the block's geometry and mode
 *     was resolved on the host at upload: LDS offsets, edge sizes, filter / class flags, angles),
 *   - gathers left[]/top[] from the staged CTU, one element per lane, and substitutes missing
 *     samples with wave-uniform lane reads (v_readlane) instead of the reference's serial fills,
 *   - smooths with whole-wave DPP shifts, publishes left[]/top[] in LDS once,
 *   - predicts 4 consecutive samples per lane in a mode-class specific loop, adds the residual and
 *     writes the staged CTU in LDS (for the next sub-level); the finished CTU goes to HBM in one coalesced sweep.
 * Sub-levels are separated by an LDS-only workgroup barrier, so the dependent chain inside a CTU
 * costs a handful of LDS round trips per block inside one CU — no kernel launch, no HBM round
 * trip.  CTUs of one launch are mutually independent (same wavefront level, recorder.c).
 * ======================================================================================= */
#define INTRA_MAX_WAVES 8
#define OH_TICKET_STRIDE 32u                       /* words between ticket counters: a cache line each */
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

/* four finished samples of one row -> the staged CTU (8-byte aligned by construction); the CTU goes to HBM in one
 * coalesced sweep when its last sub-level is done (intra_ctu_kernel) */
static __device__ __forceinline__ void put4(uint16_t *__restrict__ lds, int v0, int v1, int v2, int v3)
{
    *(uint2v *)lds = uint2v{ (unsigned)(v0 | (v1 << 16)), (unsigned)(v2 | (v3 << 16)) };
}

/* four samples as ONE write-through store (global_store_dword / dwordx2 sc1): what a CTU hands to another CTU of the same launch
 * leaves the XCD's L2 with the store itself, so the hand-off needs no agent-scope release (buffer_wbl2 writes back every dirty
 * line of the L2: measured, it doubled the launch) — the producer drains its stores and sets its flag, the consumer keeps its
 * acquire (CDNA4 guide, inter-workgroup visibility: valid forms) */
template <typename PX>
static __device__ __forceinline__ void store4_wt(GLOBAL PX *p, int a, int b, int c, int d)
{
    if (sizeof(PX) == 1)
        __hip_atomic_store((GLOBAL uint32_t *)p, (uint32_t)(a | (b << 8) | (c << 16) | (d << 24)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
        __hip_atomic_store((GLOBAL uint64_t *)p, (uint64_t)(uint32_t)(a | (b << 16)) | (uint64_t)(uint32_t)(c | (d << 16)) << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* Where the CTU's samples live while the pass works on it.
 *   staged  the CTU's copy in LDS (16-bit entries, layout dev_frame.h: oh_ctu_areas); a block's cm_off / top_off / rs come resolved
 *           in its descriptor; entry 0 holds 1 << (bit_depth - 1), the substitute when no neighbour exists at all (:251-257);
 *   direct  the picture itself in HBM (intra_direct_kernel: a wave per CTU, nothing staged): indices are samples from the first
 *           sample of plane 0 — the three planes of a picture half lie in one allocation (engine.hip: pic_layout) — derived from
 *           the block's position; index -1 stands for the substitute. */
template <typename PX, bool DIRECT> struct Samples;
template <typename PX> struct Samples<PX, false> {
    uint16_t *__restrict__ M;
    static constexpr int NONE = 0;
    __device__ __forceinline__ int ld(int i) const { return M[i]; }
    __device__ __forceinline__ void st4(int i, int v0, int v1, int v2, int v3) const { put4(M + i, v0, v1, v2, v3); }
    __device__ __forceinline__ void st1(int i, int v) const { M[i] = (uint16_t)v; }
};
template <typename PX> struct Samples<PX, true> {
    GLOBAL PX *g;
    int mid;
    bool wt;                                             /* wave-uniform: another CTU of this launch waits for these samples: store them write-through */
    static constexpr int NONE = -1;
    __device__ __forceinline__ int ld(int i) const { const int v = g[max(i, 0)]; return i < 0 ? mid : v; }
    __device__ __forceinline__ void st4(int i, int v0, int v1, int v2, int v3) const
    {
        if (wt) store4_wt<PX>(g + i, v0, v1, v2, v3); else store4<PX>(g + i, v0, v1, v2, v3);
    }
    __device__ __forceinline__ void st1(int i, int v) const
    {
        if (wt) __hip_atomic_store(g + i, (PX)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else g[i] = (PX)v;
    }
};
/* direct: where a block's samples are (cm = its sample (0,0), top = the row above, rs = row stride), from its position */
struct PlaneGeo { int off1, off2, st0, st1; };             /* sample offsets of planes 1, 2 from plane 0; strides of luma / chroma */
template <bool DIRECT>
static __device__ __forceinline__ void block_geo(const PlaneGeo &G, const uint32_t w0, const uint32_t w1, const uint32_t w3, const uint32_t w4, int &cm_off, int &top_off, int &rs)
{
    if (DIRECT) {
        const int bx = w0 & 0xffff, by = w0 >> 16, ci = w1 & 0xff;
        rs = ci ? G.st1 : G.st0;
        cm_off = (ci == 0 ? 0 : ci == 1 ? G.off1 : G.off2) + __mul24(by, rs) + bx;
        top_off = cm_off - rs;
    } else {
        cm_off = w3 & 0xffff; top_off = w3 >> 16; rs = w4 & 0xffff;
    }
}

/* constrained_intra_pred (hevcpred_template.c:185-286) for one block, lane-parallel.  Lane i owns left[i] and top[i] (i = 0..63) as
 * gathered with the re-derived candidate flags (unavailable entries hold the reference's memset value); lm1 / tm1 are left[-1] /
 * top[-1].  lm / tm: bit g = the 4-sample group g of the left column / top row lies in an intra CU; corner_intra likewise.
 *
 * The reference walks the arrays with four kinds of sweeps whose loop-carried value `a` only ever comes from an INTRA group, and
 * intra groups are never written by the sweep that reads them — so every sweep is a map: a non-intra group takes one element of the
 * nearest intra group before it (EXTEND_DOWN / RIGHT: its LAST element) or after it (EXTEND_UP: its FIRST element), found with
 * clz / ctz on the group mask and fetched with one lane permute; EXTEND_LEFT only ever fills the run of non-intra samples in front
 * of the first intra one.  The sweeps run in the reference's order, each on the arrays the previous one left, followed by its
 * plain fills for unavailable edges (:251-286).  All conditions are wave-uniform.  tests: every `*_cip*` picture case, golden
 * `b_10b_cip`, the random sweeps; the map form was checked against the sweep form on 200 k random edge configurations. */
static __device__ __forceinline__ int cip_fwd(const int v, const unsigned mask, const int limit, const int init, const int lane)
{
    const int g = lane >> 2;
    const unsigned below = mask & ((1u << g) - 1u);
    const int got = __shfl(v, below ? 4 * (31 - __clz((int)below)) + 3 : lane);
    return (lane < limit && !((mask >> g) & 1u)) ? (below ? got : init) : v;
}
static __device__ __forceinline__ int cip_bwd(const int v, const unsigned mask, const int limit, const int lane)
{
    const int g = lane >> 2;
    const unsigned above = ((mask >> (g + 1)) << (g + 1)) & ((1u << (limit >> 2)) - 1u);
    const int init = __builtin_amdgcn_readlane(v, (limit - 1) & 63);
    const int got = __shfl(v, above ? 4 * (__ffs((int)above) - 1) : lane);
    return (lane < limit && !((mask >> g) & 1u)) ? (above ? got : init) : v;
}
static __device__ __forceinline__ void cip_lanes(int &Lv, int &Tv, int &corner, const int lane, const int n, const int avail, unsigned lm, unsigned tm,
                                              const bool corner_intra, const int smx, const int smy, const int bl_size,
                                              const bool x_nz, const bool y_nz, const int bd, int lm1, int tm1)
{
    bool a_bl = avail & OH_AV_BOTTOM_LEFT, a_l = avail & OH_AV_LEFT, a_ul = avail & OH_AV_UP_LEFT, a_u = avail & OH_AV_UP, a_ur = avail & OH_AV_UP_RIGHT;
    lm &= 0xffffu; tm &= 0xffffu;
    if (a_bl || a_l || a_ul || a_u || a_ur) {
        /* first sample of the top row inside an intra CU, or size_max_x (:205-211, :217-219); an available edge always holds one */
        const unsigned tmx = tm & ((1u << ((smx + 3) >> 2)) - 1u);
        const int jt = min(tmx ? min(4 * (__ffs((int)tmx) - 1), smx) : smx, 63);
        const int tj = __builtin_amdgcn_readlane(Tv, jt);
        if (a_bl || a_l || a_ul) {
            const int j0 = n + bl_size - 1;
            if (!((lm & ((1u << ((j0 >> 2) + 1)) - 1u)) || corner_intra)) {       /* nothing intra in the left column or the corner: start from the top row */
                if (lane < jt) Tv = tj;
                tm1 = tj;
            }
        } else if (jt > 0) {
            if (lane < jt) Tv = tj;
            if (!x_nz || !corner_intra) tm1 = tj;
        }
        lm1 = tm1;
        if (a_bl || a_l) Lv = cip_fwd(Lv, lm, smy, lm1, lane);
        if (!a_l && lane < n) Lv = lm1;
        if (!a_bl) { const int v = __builtin_amdgcn_readlane(Lv, (n - 1) & 63); if (lane >= n && lane < 2 * n) Lv = v; }
        if (!x_nz) {
            if (lane < smy) Lv = 0;
        } else {
            Lv = cip_bwd(Lv, lm, smy, lane);
            if (y_nz && !corner_intra) lm1 = __builtin_amdgcn_readlane(Lv, 0);
        }
        tm1 = lm1;
        if (y_nz) Tv = cip_fwd(Tv, tm, smx, lm1, lane);
    }
    /* missing samples, :251-286 */
    if (!a_bl) {
        if (a_l) {
            const int v = __builtin_amdgcn_readlane(Lv, (n - 1) & 63);
            if (lane >= n && lane < 2 * n) Lv = v;
        } else if (a_ul) {
            if (lane < 2 * n) Lv = lm1;
            a_l = true;
        } else if (a_u) {
            lm1 = __builtin_amdgcn_readlane(Tv, 0);
            if (lane < 2 * n) Lv = lm1;
            a_ul = a_l = true;
        } else if (a_ur) {
            const int v = __builtin_amdgcn_readlane(Tv, n & 63);
            if (lane < n) Tv = v;
            lm1 = v;
            if (lane < 2 * n) Lv = lm1;
            a_u = a_ul = a_l = true;
        } else {
            lm1 = 1 << (bd - 1);
            if (lane < 2 * n) { Lv = lm1; Tv = lm1; }
        }
    }
    if (!a_l) { const int v = __builtin_amdgcn_readlane(Lv, n & 63); if (lane < n) Lv = v; }
    if (!a_ul) lm1 = __builtin_amdgcn_readlane(Lv, 0);
    if (!a_u && lane < n) Tv = lm1;
    if (!a_ur) { const int v = __builtin_amdgcn_readlane(Tv, (n - 1) & 63); if (lane >= n && lane < 2 * n) Tv = v; }
    corner = lm1;                                                           /* top[-1] = left[-1] */
}

template <typename PX, bool CIP, bool STAGED, bool DIRECT>
static __device__ __forceinline__ void intra_block(const DevFrame *__restrict__ f, const int bd,
                                                   const uint4v q0, const uint4v q1, IntraLds &s, const Samples<PX, DIRECT> S, const PlaneGeo &G,
                                                   const int16_t *__restrict__ res_lds_base, const int lane, unsigned long long *acc)
{
    unsigned long long ta = 0, tb = 0, tc = 0, td = 0; (void)ta; (void)tb; (void)tc; (void)td; (void)acc;
    STAMP(ta);
    /* the descriptor is the same for every lane: keep it in scalar registers */
    const uint32_t w0 = __builtin_amdgcn_readfirstlane(q0[0]), w1 = __builtin_amdgcn_readfirstlane(q0[1]);
    const uint32_t res_off = __builtin_amdgcn_readfirstlane(q0[2]), w3 = __builtin_amdgcn_readfirstlane(q0[3]);
    const uint32_t w4 = __builtin_amdgcn_readfirstlane(q1[0]), w5 = __builtin_amdgcn_readfirstlane(q1[1]);
    const uint32_t res_lds = __builtin_amdgcn_readfirstlane(q1[2]);
    const int bx = w0 & 0xffff, by = w0 >> 16, log2 = (w1 >> 8) & 0xff, avail = w1 >> 24;
    const int tr_size = (w4 >> 16) & 0xff, bl_size = w4 >> 24;
    int cm_off, top_off, rs;
    block_geo<DIRECT>(G, w0, w1, w3, w4, cm_off, top_off, rs);
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
    /* the three loads are unconditional (one wait instead of three): a value fetched for an unavailable neighbour is
     * never selected below, and every address stays inside the workgroup's LDS block (lanes >= 2n read row 0) */
    const bool t_ok = i < n ? a_u : (i < 2 * n && a_ur), l_ok = i < n ? a_l : (i < 2 * n && a_bl);
    const int ti = i < n ? i : (i < 2 * n ? (i - n < tr_size ? i : n + tr_size - 1) : 0);
    const int li = i < n ? i : (i < 2 * n ? (i - n < bl_size ? i : n + bl_size - 1) : 0);
    const int tv = S.ld(top_off + ti), lv = S.ld(cm_off - 1 + li * rs), cv = S.ld(top_off - 1);
    /* substitution (:251-286) in closed form: the reference's cascaded fills only ever copy one of
     * these wave-uniform values */
    const int l_0 = __builtin_amdgcn_readlane(lv, 0), l_n1 = __builtin_amdgcn_readlane(lv, n - 1), l_n = __builtin_amdgcn_readlane(lv, n & 63);
    const int t_0 = __builtin_amdgcn_readlane(tv, 0), t_n1 = __builtin_amdgcn_readlane(tv, n - 1), t_n = __builtin_amdgcn_readlane(tv, n & 63);
    int corner, left_i, top_i;
    if (CIP && (flags & OH_IF_CIP)) {
        /* constrained intra prediction (own kernel instantiation, so the common one carries none of this): the reference's sweeps as
         * lane-parallel maps over the edge arrays in registers (cip_lanes) */
        const int fill = sizeof(PX) == 1 ? 128 : 0x8080;                   /* memset(.., 128, ..) over 16-bit samples, :158-160 */
        left_i = l_ok ? lv : fill;
        top_i = t_ok ? tv : fill;
        cip_lanes(left_i, top_i, corner, lane, n, avail, w7 & 0xffff, w7 >> 16, (flags & OH_IF_CIP_CORNER) != 0, a_ur ? n + tr_size : n, a_bl ? n + bl_size : n,
                  a_bl ? bl_size : 0, bx != 0, by != 0, bd, a_ul ? cv : 0, a_ul ? cv : 128);
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
#define GROUP_LOOP_BEGIN                                                                  \
    _Pragma("unroll") for (int k = 0; k < 4; k++) {                                       \
        const int g = lane + 64 * k;                                                      \
        if (g >= ngroups) break;                                                          \
        const int y = (4 * g) >> log2, x0 = (4 * g) & (n - 1);                            \
        int v[4];
#define GROUP_LOOP_END                                                                    \
        if (has_res) { _Pragma("unroll") for (int j = 0; j < 4; j++) v[j] = clip_px(v[j] + rv[k][j], bd); } \
        S.st4(cm_off + y * rs + x0, v[0], v[1], v[2], v[3]);                                               \
    }
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
 * block descriptor held per lane instead of in scalar registers, split in two halves:
 *
 *   slots_prepare  everything that does not depend on sample values: descriptor decode, the LDS address each lane's
 *                  left[i] / top[i] / corner comes from with the substitution (:251-286) ALREADY resolved (the
 *                  reference's cascaded fills only ever copy one of a few fixed samples, so picking the source is
 *                  address arithmetic), the edge-array indices the lane's prediction will read, its residual;
 *   slots_finish   the dependent chain: three sample loads, smoothing (DPP row shifts, which stop at the slot
 *                  boundary by construction), publish, seven edge reads, the mode-class arithmetic, residual, store.
 *
 * A sub-level can only start when the previous one is in LDS, but its slots_prepare can run before that: the kernel
 * deals the sub-levels round-robin to K groups of waves, so a group prepares its next sub-level while others finish theirs.
 * Blocks of a sub-level are independent, so the four of a pass need no ordering.  edges: 4 x 36 ints,
 * per slot [0] = left[-1], [1..16] = left[0..15], [17] = top[-1], [18..33] = top[0..15]. */
static __device__ __forceinline__ int row_shr1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xf, 0xf, false); }
static __device__ __forceinline__ int row_shl1(int v, int fill) { return __builtin_amdgcn_update_dpp(fill, v, 0x101, 0xf, 0xf, false); }

/* a block's 32-byte descriptor from the staged copy (LDS) or from the list in HBM */
static __device__ __forceinline__ void item_load(const DevIntra *__restrict__ items, const uint32_t i, uint4v &q0, uint4v &q1)
{
    const uint4v *__restrict__ it = (const uint4v *)&items[i];
    q0 = it[0]; q1 = it[1];
}
static __device__ __forceinline__ void item_load(const GLOBAL DevIntra *__restrict__ items, const uint32_t i, uint4v &q0, uint4v &q1)
{
    const GLOBAL uint4v *__restrict__ it = (const GLOBAL uint4v *)&items[i];
    q0 = it[0]; q1 = it[1];
}

enum { SLOT_LB = 1, SLOT_TB = 18 };                          /* left[k] = E[SLOT_LB + k], top[k] = E[SLOT_TB + k] */
enum { SS_ACT = 1, SS_WORK = 2, SS_H = 4, SS_FILTER = 8, SS_EDGE = 16, SS_RES = 32, SS_SMOOTH_LANE = 64 };
struct SlotState {                                           /* per lane, carried from slots_prepare to slots_finish */
    int src_l, src_t, src_c;                                 /* M indices of the samples that become left[i], top[i], corner */
    int e[7];                                                /* indices into the slot's E read by the prediction */
    int rv[4];                                               /* residual of the lane's four samples */
    int dst, dstep;                                          /* M index of the lane's first sample; step to the next (column lines) */
    int bits;                                                /* SS_* | cls << 8 | log2 << 12 | fact << 16 */
};

template <bool STAGED, bool DIRECT, typename ITEMS>
static __device__ __forceinline__ void slots_prepare(SlotState &st, const ITEMS items, const uint32_t first, const int count, const PlaneGeo &G,
                                                     const int16_t *__restrict__ res_lds_base, const GLOBAL int16_t *__restrict__ res_pool, const int lane)
{
    /* straight-line code: every choice is a select (the compiler would turn if/else into exec-mask branches, and
     * this runs beside the dependent chain of the sub-level before) */
    const int slot = lane >> 4, sl = lane & 15;
    const bool act = slot < count;
    uint4v q0, q1;
    item_load(items, first + (act ? slot : 0), q0, q1);
    const uint32_t w1 = q0[1], res_off = q0[2], w3 = q0[3], w4 = q1[0], w5 = q1[1], res_lds = q1[2];
    const int log2 = (w1 >> 8) & 0xff, avail = w1 >> 24;
    const int tr_size = (w4 >> 16) & 0xff, bl_size = w4 >> 24;
    int cm_off, top_off, rs;
    block_geo<DIRECT>(G, q0[0], w1, w3, w4, cm_off, top_off, rs);
    const int angle = (int)(int8_t)(w5 & 0xff), flags = (w5 >> 8) & 0xff, inv_a = (int)(int16_t)(w5 >> 16);
    const int n = 1 << log2, cls = (flags >> 4) & 7, ngroups = (n * n) >> 2;
    const bool a_bl = avail & OH_AV_BOTTOM_LEFT, a_l = avail & OH_AV_LEFT, a_ul = avail & OH_AV_UP_LEFT;
    const bool a_u = avail & OH_AV_UP, a_ur = avail & OH_AV_UP_RIGHT;
    const int i = sl;                                                  /* edge element this lane owns (2n <= 16) */
    const bool work = act && sl < ngroups;                             /* line this lane predicts */

    /* lane group g = sl: four samples of one line of the block — a row (o = y, m0 = x0), or for the horizontal modes a
     * column (o = x, m0 = y0): transposed, those modes are the vertical ones with left[] and top[] swapped */
    const bool is_h = cls == OH_IC_ANG_H || cls == OH_IC_PURE_H;
    const int o = (4 * sl) >> log2, m0 = (4 * sl) & (n - 1);
    const bool has_res = res_off != OH_NO_COEFF;
    {   /* the line's residual: unconditional loads (offset 0 of the span / the pool when the block has none; SS_RES guards the use).
         * Not STAGED: straight from the residual pool in HBM — this runs a sub-level ahead of the chain that consumes the values, so
         * the round trip is hidden, and the launch needs no LDS for the CTU's residual span (more workgroups per CU) */
        const int line = is_h ? (m0 << log2) + o : 4 * sl, step = is_h ? n : 1;
        if (STAGED) {
            const int16_t *__restrict__ rp = res_lds_base + (has_res ? res_lds : 0u) + line;
#pragma unroll
            for (int j = 0; j < 4; j++) st.rv[j] = rp[j * step];
        } else {
            const GLOBAL int16_t *__restrict__ rp = res_pool + (has_res ? res_off : 0u) + line;
#pragma unroll
            for (int j = 0; j < 4; j++) st.rv[j] = rp[j * step];
        }
    }

    /* gather (:164-183) + substitution (:251-286) as source addresses.  M[0] holds 1 << (bit_depth - 1) (kernel prologue).
     * Lanes >= 2n fetch row 0 / column 0: values nobody reads. */
    {
        const int ti = i < n ? i : (i < 2 * n ? (i - n < tr_size ? i : n + tr_size - 1) : 0);
        const int li = i < n ? i : (i < 2 * n ? (i - n < bl_size ? i : n + bl_size - 1) : 0);
        const int own_t = top_off + ti, own_l = cm_off - 1 + __mul24(li, rs);
        const int l_0 = cm_off - 1, l_n1 = l_0 + __mul24(n - 1, rs), l_n = l_n1 + rs;
        const int t_0 = top_off, t_n1 = top_off + n - 1, t_n = top_off + n, c_own = top_off - 1;
        const bool any_l = a_bl || a_l;
        const int c_l = a_l ? l_0 : l_n, c_t = a_u ? t_0 : (a_ur ? t_n : (DIRECT ? -1 : 0));     /* nothing available at all: the substitute */
        const int corner = a_ul ? c_own : (any_l ? c_l : c_t);
        const int left_a = i < n ? (a_l ? own_l : l_n) : (a_bl ? own_l : l_n1);
        st.src_c = corner;
        st.src_l = any_l ? left_a : corner;
        st.src_t = i < n ? (a_u ? own_t : corner) : (a_ur ? own_t : (a_u ? t_n1 : corner));
    }

    /* the prediction's reads of the published edges (:359-538).  Angular modes incl. 10 and 26 (angle 0): one
     * (idx, fact) per line, five consecutive reference samples; ref[k] == main[k-1] for k >= 0, the projected side sample
     * for k < 0 (:447-453, :480-486).  Planar / DC: the four top samples of the line, left[o], and top[n], left[n] /
     * left[0], top[0]. */
    const int LB = SLOT_LB, TB = SLOT_TB;
    const bool ang = cls >= OH_IC_ANG_V, planar = cls == OH_IC_PLANAR;
    const int MB = is_h ? LB : TB, SB = is_h ? TB : LB;
    const int ta = (o + 1) * angle, id = ta >> 5, fact = ang ? ta & 31 : 0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const int kk = m0 + j + id + 1;
        const int ea = kk >= 0 ? MB + kk - 1 : SB - 1 + ((kk * inv_a + 128) >> 8);
        st.e[j] = ang ? ea : (j < 4 ? TB + m0 + j : LB + o);
    }
    st.e[5] = ang ? SB + o : (planar ? TB + n : LB);
    st.e[6] = ang ? SB - 1 : (planar ? LB + n : TB);
    st.dst = cm_off + (is_h ? __mul24(m0, rs) + o : __mul24(o, rs) + m0);
    st.dstep = rs;
    st.bits = (act ? SS_ACT : 0) | (work ? SS_WORK : 0) | (is_h ? SS_H : 0) | ((flags & OH_IF_FILTER) ? SS_FILTER : 0) |
              ((flags & OH_IF_EDGE) ? SS_EDGE : 0) | (has_res ? SS_RES : 0) | (i < 2 * n - 1 ? SS_SMOOTH_LANE : 0) |
              (cls << 8) | (log2 << 12) | (fact << 16);
}

template <typename PX, bool DIRECT>
static __device__ __forceinline__ void slots_finish(const SlotState &st, const int bd, int *__restrict__ edges, const Samples<PX, DIRECT> S,
                                                    const int lane, unsigned long long *acc)
{
    unsigned long long sa = 0, sb = 0, sc = 0, sd = 0, se = 0; (void)sa; (void)sb; (void)sc; (void)sd; (void)se; (void)acc;
    STAMP(sa);
    const int slot = lane >> 4, sl = lane & 15, bits = st.bits;
    const int cls = (bits >> 8) & 7, log2 = (bits >> 12) & 7, fact = bits >> 16, n = 1 << log2;
    const bool act = bits & SS_ACT, work = bits & SS_WORK;
    int left_i = S.ld(st.src_l), top_i = S.ld(st.src_t), corner = S.ld(st.src_c);

    if (__builtin_amdgcn_ballot_w64(act && (bits & SS_FILTER)) != 0) {
        /* smoothing (:288-326); never the strong filter here (32x32 only).  All lanes run the shifts, the flag selects. */
        const int lp = row_shr1(left_i, corner), ln = row_shl1(left_i, 0);
        const int tp = row_shr1(top_i, corner), tn = row_shl1(top_i, 0);
        if (bits & SS_FILTER) {
            corner = (left_i + 2 * corner + top_i + 2) >> 2;      /* only lane 0 of the slot publishes the corner: its own left[0], top[0] */
            if (bits & SS_SMOOTH_LANE) {
                left_i = (ln + 2 * left_i + lp + 2) >> 2;
                top_i  = (tn + 2 * top_i + tp + 2) >> 2;
            }
        }
    }
    STAMP(sb);
    int *E = edges + slot * 36;
    if (act) {
        E[SLOT_LB + sl] = left_i; E[SLOT_TB + sl] = top_i;   /* entries >= 2n are written too and never read */
        if (sl == 0) { E[0] = corner; E[17] = corner; }
    }
    WSYNC();
    /* the DC sum needs every lane of the slot: outside the per-line predicate (and skipped when no slot is DC) */
    int dc = 0;
    if (__builtin_amdgcn_ballot_w64(act && cls == OH_IC_DC) != 0) {
        int part = sl < n ? left_i + top_i : 0;                        /* sum over the slot's 16 lanes: DPP, no LDS round trips */
        part += __builtin_amdgcn_update_dpp(0, part, 0xb1, 0xf, 0xf, false);      /* quad_perm [1,0,3,2] */
        part += __builtin_amdgcn_update_dpp(0, part, 0x4e, 0xf, 0xf, false);      /* quad_perm [2,3,0,1] */
        part += __builtin_amdgcn_update_dpp(0, part, 0x141, 0xf, 0xf, false);     /* row_half_mirror */
        part += __builtin_amdgcn_update_dpp(0, part, 0x140, 0xf, 0xf, false);     /* row_mirror */
        dc = (part + n) >> (log2 + 1);
    }
    STAMP(sc);
    if (work) {
        const int o = (4 * sl) >> log2, m0 = (4 * sl) & (n - 1);
        int r[7], v[4];
#pragma unroll
        for (int j = 0; j < 7; j++) r[j] = E[st.e[j]];
        if (cls >= OH_IC_ANG_V) {
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = ((32 - fact) * r[j] + fact * r[j + 1] + 16) >> 5;     /* fact == 0: r[j] */
            if ((bits & SS_EDGE) && m0 == 0) v[0] = clip_px(r[0] + ((r[5] - r[6]) >> 1), bd);        /* :474-477, :501-508 */
        } else if (cls == OH_IC_PLANAR) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = m0 + j;
                v[j] = ((n - 1 - x) * r[4] + (x + 1) * r[5] + (n - 1 - o) * r[j] + (o + 1) * r[6] + n) >> (log2 + 1);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int x = m0 + j;
                int pv = dc;
                if (bits & SS_EDGE) {                         /* :410-416 */
                    if (x == 0 && o == 0) pv = (r[5] + 2 * dc + r[6] + 2) >> 2;
                    else if (o == 0)      pv = (r[j] + 3 * dc + 2) >> 2;
                    else if (x == 0)      pv = (r[4] + 3 * dc + 2) >> 2;
                }
                v[j] = pv;
            }
        }
        STAMP(sd);
        if (bits & SS_RES) {
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = clip_px(v[j] + st.rv[j], bd);
        }
        if (bits & SS_H) {
#pragma unroll
            for (int j = 0; j < 4; j++) S.st1(st.dst + j * st.dstep, v[j]);
        } else {
            S.st4(st.dst, v[0], v[1], v[2], v[3]);
        }
    }
    WSYNC();                                                 /* the wave's edge arrays are reused by its next pass */
    STAMP(se);
    ACC(5, sa, sb); ACC(6, sb, sc); ACC(7, sc, sd); ACC(8, sd, se);
}

/* ---- the schedule as a dependency graph: one launch per picture batch (intra_dag_kernel, intra_direct_kernel) ----
 * Entry k of ictu[] may start when the entries ctu_wait[4k..4k+3] (prep_intra_wait: neighbour CTUs of a lower level its blocks gather
 * from) have set ctu_done.  Hand-off = the placement-independent form of the CDNA4 guide: producer — every storing wave drains its
 * stores, workgroup barrier, ONE lane: agent-scope release, drain, relaxed agent-scope flag store; consumer — ONE wave polls
 * (relaxed agent-scope loads, s_sleep between polls), takes ONE agent-scope acquire, drains it, then the workgroup barrier in front
 * of every load of the samples.  The poll is bounded: a wave that gives up latches OH_KE_DAG_TIMEOUT in the engine's error word
 * (pinned host memory; oh_engine_sync and every other wait report it) and goes on with what is there, so the grid always drains. */
static __device__ __forceinline__ void dag_latch_error(const DevFrame *__restrict__ f, const uint32_t code, const uint32_t where)
{
    uint32_t *w = f->err_word;
    if (w && __hip_atomic_load(&w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u) {
        __hip_atomic_store(&w[1], (uint32_t)f->cur_pic_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&w[2], where, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __hip_atomic_store(&w[0], code, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
/* called by ONE wave (all 64 lanes): lanes 0..3 poll one awaited entry each */
static __device__ __forceinline__ void dag_wait_wave(const DevFrame *__restrict__ f, const uint32_t entry, const int lane, const uint32_t spin_limit, const int exp = 0)
{
    const uint32_t w = lane < 4 ? G_CONST(uint32_t, f->ctu_wait)[4 * entry + lane] : ~0u;
    uint32_t *done = f->ctu_done;
    bool ok = w == ~0u;
    uint32_t spin = 0;
    for (;;) {
        if (!ok) ok = __hip_atomic_load(&done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull)
            break;
        if (++spin >= spin_limit) {
            if (lane == 0) dag_latch_error(f, OH_KE_DAG_TIMEOUT, entry);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    if (exp & 8) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            /* the awaited CTUs' samples: not from this CU's L1 / stale L2 lines */
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              /* ... and the invalidate has completed before anyone loads */
}
/* called by ONE lane after every storing wave has drained its stores (and, with several waves, after their barrier) */
template <bool WRITE_THROUGH>
static __device__ __forceinline__ void dag_publish_lane(const DevFrame *__restrict__ f, const uint32_t entry)
{
    if (!WRITE_THROUGH)                                           /* plain stores: the XCD's L2 has to be written back */
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              /* the compiler may drop the fence's own wait (ROCm 7.2): the flag must not overtake the write-back */
    __hip_atomic_store(&f->ctu_done[entry], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* one CTU by the whole workgroup: schedule entry `entry` of picture f.  Every thread of the workgroup calls it (barriers inside).
 * DAG: the entry waits for / publishes to other entries of the same launch (above). */
template <typename PX, bool CIP, bool STAGED, bool DAG>
static __device__ __forceinline__ void intra_ctu_body(const DevFrame *__restrict__ f, const OhIntraLaunch &L, const uint32_t entry, unsigned char *smem,
                                                      const uint32_t spin_limit)
{
    uint16_t *__restrict__ M = (uint16_t *)smem;                                  /* sample area, oh_ctu_areas() */
    DevIntra *__restrict__ items = (DevIntra *)(smem + L.off_items);
    uint32_t *__restrict__ sub = (uint32_t *)(smem + L.off_sub);
    uint32_t *__restrict__ small = (uint32_t *)(smem + L.off_small);              /* per sub-level: leading blocks that go four per wave */
    int16_t *__restrict__ res_l = (int16_t *)(smem + L.off_res);                  /* STAGED: the CTU's residual blocks (DevIntraCtu.res_lo/res_cnt) */
    const GLOBAL int16_t *__restrict__ res_pool = G_CONST(int16_t, f->res);       /* else: every block fetches its own from the pool */
    const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x, nwaves = nthr >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                    /* wave-uniform: the phase bookkeeping derived from it lives in scalar registers */
    IntraLds &edges = *(IntraLds *)(smem + L.off_wave + wave * OH_INTRA_WAVE_LDS);
    const DevIntraCtu ctu = gload(f->ictu + entry);
    const GLOBAL uint32_t *__restrict__ ss = G_CONST(uint32_t, f->sub_start) + ctu.sub_first;
    const OhPicParams &pp = f->pp;
    const int lc = pp.log2_ctb_size, ctbw = (pp.width + (1 << lc) - 1) >> lc;
    const int cx0 = (ctu.ctu % ctbw) << lc, cy0 = (ctu.ctu / ctbw) << lc;      /* luma origin of the CTU */
    const int n_sub = (int)ctu.n_sub;                                             /* <= OH_MAX_CTU_BLOCKS: prep_intra_ctu rejects the list otherwise */
    const int bd = pp.bit_depth;
    const OhCtuAreas ar = oh_ctu_areas(lc, pp.chroma_format_idc);

    /* stage: block descriptors, sub-level table, residual blocks */
    const uint32_t item0 = ctu.item0, n_items = ctu.n_items & 0xffffu;       /* <= OH_MAX_CTU_BLOCKS, checked by prep_intra_ctu */
    /* write-back form, decided while the entry is at hand and kept in a scalar register: blocks (mostly-inter CTU) or rectangle */
    int blockwise;
    {
        uint32_t rect = 0;
        for (int c = 0; c < (pp.chroma_format_idc ? 3 : 1); c++) {
            const int hs = hsh(pp, c), vs = vsh(pp, c);
            rect += (uint32_t)(max(((ctu.bx1 + (1 << hs) - 1) >> hs) - max(ctu.bx0 >> hs, 0), 0) * max(((ctu.by1 + (1 << vs) - 1) >> vs) - max(ctu.by0 >> vs, 0), 0));
        }
        blockwise = __builtin_amdgcn_readfirstlane((int)((ctu.n_items >> 16) * 128u < rect));
    }
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
    uint32_t aux = 0;
    if (DAG) {
        /* descriptors and residual are on their way; now the neighbours.  Wave 0 polls and acquires, the barrier holds the others */
        aux = __builtin_amdgcn_readfirstlane(G_CONST(uint32_t, f->ctu_aux)[entry]);
        if (aux & OH_AUX_WAITS) {
            if (wave == 0) dag_wait_wave(f, entry, lane, spin_limit);
            __syncthreads();
        }
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
    if (tid == 0) M[0] = (uint16_t)(1 << (bd - 1));          /* the substitute when no neighbour exists at all (:251-257); a padding cell */
    __syncthreads();
    unsigned long long acc[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 }, t0 = 0, t1 = 0, t2 = 0, tk = 0, rt1 = 0; (void)acc; (void)t0; (void)t1; (void)t2; (void)tk; (void)rt1;
#ifdef OH_STAMPS
    STAMP(tk);
    rt1 = __builtin_amdgcn_s_memrealtime();
#endif

    /* Sub-level s belongs to the waves of phase s % K (K = L.phases): while they finish s (the dependent chain), the
     * waves that finished s-1 prepare their first four-block pass of s-1+K (everything that needs no samples) and then
     * sit out the sub-levels in between.  Units of a sub-level: groups of up to four <=8x8 blocks (one 16-lane slot
     * each), then the bigger blocks one per wave; the slot path has no
     * constrained-intra variant. */
    const int K = (int)L.phases, ph = wave % K, wi = wave / K, nwk = nwaves / K;
    const Samples<PX, false> S = { M };
    const PlaneGeo G0 = { 0, 0, 0, 0 };
    SlotState st;
    bool ready = false;                                      /* st holds the prepared unit `wi` of this wave's next sub-level */
    uint32_t b0 = 0, ns = 0, nunits = 0;                     /* this wave's next sub-level: first block, blocks in slots, units */
    auto prepare = [&](int s) {
        const uint32_t s0 = sub[s], s1 = sub[s + 1];
        b0 = s0;
        ns = !CIP ? min(small[s], s1 - s0) : 0u;
        nunits = ((ns + 3) >> 2) + (s1 - s0 - ns);
        ready = (uint32_t)wi < ((ns + 3) >> 2);
        if (ready)
            slots_prepare<STAGED, false>(st, (const DevIntra *)items, b0 + 4 * wi, (int)min(4u, ns - 4 * wi), G0, res_l, res_pool, lane);
    };
    if (ph < n_sub) prepare(ph);
    for (int s = 0, sp = 0; s < n_sub; s++, sp = sp + 1 == K ? 0 : sp + 1) {      /* sp = s % K */
        STAMP(t0);
        if (sp == ph) {
            const uint32_t ngrp = (ns + 3) >> 2;
            for (uint32_t u = wi; u < nunits; u += nwk) {
                if (u < ngrp) {
                    if (!ready)
                        slots_prepare<STAGED, false>(st, (const DevIntra *)items, b0 + 4 * u, (int)min(4u, ns - 4 * u), G0, res_l, res_pool, lane);
                    ready = false;
                    slots_finish<PX, false>(st, bd, edges.E, S, lane, acc);
                } else {
                    uint4v q0, q1;
                    item_load((const DevIntra *)items, b0 + ns + (u - ngrp), q0, q1);
                    intra_block<PX, CIP, STAGED, false>(f, bd, q0, q1, edges, S, G0, res_l, lane, acc);
                }
            }
        } else if (sp == (ph + 1 == K ? 0 : ph + 1) && s - 1 + K < n_sub) {
            prepare(s - 1 + K);
        }
        STAMP(t1);
        LDS_BARRIER();                                    /* next sub-level reads what this one wrote to LDS */
        STAMP(t2);
        ACC(0, t0, t1); ACC(1, t1, t2);
    }
    /* The last sub-level's barrier has been passed.  A CTU that is mostly INTER (B pictures: a few intra blocks scattered over a
     * rectangle that spans the CTU) stores its blocks, a quarter wave per block, instead of the rectangle: measured on the
     * >= 16 k-workgroup launches of B pictures the rectangle's write-back was 19 % of the launch (profiles/r02_intra_staging_experiment.txt) */
    const bool wt = DAG && (aux & OH_AUX_AWAITED);          /* another CTU of this launch reads these samples: write-through stores, no L2 write-back */
    if (blockwise) {
        const uint64_t gp0 = (uint64_t)f->cur.p[0], gp1 = (uint64_t)f->cur.p[1], gp2 = (uint64_t)f->cur.p[2];
        const int st0 = f->cur.stride[0], st1 = f->cur.stride[1];
        const int quarter = tid >> 4, sl = tid & 15, nq = nthr >> 4;
        for (uint32_t it = quarter; it < n_items; it += nq) {
            const uint32_t w0 = ((const uint32_t *)&items[it])[0], w1 = ((const uint32_t *)&items[it])[1];
            const uint32_t w3 = ((const uint32_t *)&items[it])[3], w4 = ((const uint32_t *)&items[it])[4];
            const int bx = w0 & 0xffff, by = w0 >> 16, ci = w1 & 0xff, log2 = (w1 >> 8) & 0xff, n = 1 << log2;
            const int cm_off = w3 & 0xffff, rs = w4 & 0xffff, stride = ci ? st1 : st0;
            GLOBAL PX *__restrict__ g = G_MUT(PX, ci == 0 ? gp0 : (ci == 1 ? gp1 : gp2)) + (size_t)by * stride + bx;
            for (int gi = sl; gi < (n * n) >> 2; gi += 16) {
                const int y = (4 * gi) >> log2, x0 = (4 * gi) & (n - 1);
                const uint2v pk = *(const uint2v *)&M[cm_off + y * rs + x0];
                GLOBAL PX *__restrict__ d = g + (size_t)y * stride + x0;
                if (wt) {
                    if (sizeof(PX) == 1) __hip_atomic_store((GLOBAL uint32_t *)d, __builtin_amdgcn_perm(pk[1], pk[0], 0x06040200), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else                 __hip_atomic_store((GLOBAL uint64_t *)d, (uint64_t)pk[0] | (uint64_t)pk[1] << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    if (sizeof(PX) == 1) *(GLOBAL uint32_t *)d = __builtin_amdgcn_perm(pk[1], pk[0], 0x06040200);
                    else                 *(GLOBAL uint2v *)d = pk;
                }
            }
        }
    } else
    /* the reconstructed CTU goes to HBM in one coalesced sweep: the staged rectangle inside the CTU (it covers every
     * block; samples of inter blocks in it are rewritten with the values they were staged with). */
    for (int c = 0; c < nplanes; c++) {
        const int hs = hsh(pp, c), vs = vsh(pp, c);
        const int wc = (1 << lc) >> hs, hc = (1 << lc) >> vs, rs = wc + 4;
        const int x0 = cx0 >> hs, y0 = cy0 >> vs, pw = f->cur.w[c], ph = f->cur.h[c], stride = f->cur.stride[c];
        const int px0 = ctu.bx0 >> hs, px1 = (ctu.bx1 + (1 << hs) - 1) >> hs, py0 = ctu.by0 >> vs, py1 = (ctu.by1 + (1 << vs) - 1) >> vs;
        GLOBAL PX *__restrict__ g = G_MUT(PX, f->cur.p[c]);
        const uint16_t *__restrict__ Mm = M + (c == 0 ? ar.main[0] : c == 1 ? ar.main[1] : ar.main[2]);
        const int r0 = max(py0, 0), r1 = min(min(py1, hc), ph - y0);
        const int cs = max(px0, 0) & ~3, ce = min(min((px1 + 3) & ~3, wc), pw - x0);
        const int seg = tid & 15;
        if (cs + 4 * seg < ce)
            for (int row = r0 + (tid >> 4); row < r1; row += nthr >> 4) {
                const uint2v pk = *(const uint2v *)&Mm[row * rs + cs + 4 * seg + 4];
                GLOBAL PX *__restrict__ d = g + (size_t)(y0 + row) * stride + x0 + cs + 4 * seg;
                if (wt) {
                    if (sizeof(PX) == 1) __hip_atomic_store((GLOBAL uint32_t *)d, __builtin_amdgcn_perm(pk[1], pk[0], 0x06040200), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else                 __hip_atomic_store((GLOBAL uint64_t *)d, (uint64_t)pk[0] | (uint64_t)pk[1] << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    if (sizeof(PX) == 1) *(GLOBAL uint32_t *)d = __builtin_amdgcn_perm(pk[1], pk[0], 0x06040200);
                    else                 *(GLOBAL uint2v *)d = pk;
                }
            }
    }
    if (DAG && (aux & OH_AUX_AWAITED)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* this wave's stores of the CTU have left */
        __syncthreads();
        if (tid == 0) dag_publish_lane<true>(f, entry);
    }
#ifdef OH_STAMPS
    if (f->dbg && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0 && lane == 0) {
        unsigned long long te; STAMP(te);
        unsigned long long rt2 = __builtin_amdgcn_s_memrealtime();
        unsigned long long slot = atomicAdd((unsigned long long *)f->dbg, 1ull);
        if (slot < 4000) {
            unsigned long long *o = (unsigned long long *)f->dbg + 16 + slot * 16;
            o[0] = n_sub; o[1] = te - tk; o[2] = rt2 - rt1; o[3] = acc[0]; o[4] = acc[1]; o[5] = acc[2]; o[6] = acc[3]; o[7] = acc[4];
            o[8] = gridDim.x; o[9] = n_items; o[10] = tk; o[11] = acc[5]; o[12] = acc[6]; o[13] = acc[7]; o[14] = acc[8];
        }
    }
#endif
}

/* waves_per_eu(6, 8): 80 VGPRs instead of 86, i.e. six waves per SIMD instead of five for the wide B-picture launches (no spills;
 * asking for seven or eight spills and loses more than the occupancy gains) */
template <typename PX, bool CIP, bool STAGED>
__global__ __launch_bounds__(64 * INTRA_MAX_WAVES) __attribute__((amdgpu_waves_per_eu(6, 8))) void intra_ctu_kernel(const OhBatch B, const OhIntraLaunch L)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    const uint32_t first_ctu = f->lvl_start[L.level];
    if (blockIdx.x >= f->lvl_start[L.level + 1] - first_ctu)
        return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    intra_ctu_body<PX, CIP, STAGED, false>(f, L, first_ctu + blockIdx.x, smem, 0u);
}


/* =========================================================================================
 * The whole picture's intra pass in ONE launch, staged form: a workgroup per schedule entry.  A workgroup does not take the entry of
 * its blockIdx: when it STARTS it draws a ticket from a counter (atomicAdd): ticket t = schedule entry t / pictures of picture
 * t % pictures, i.e. the entries in schedule order (level by level) interleaved over the pictures of the batch.  It stages its
 * entry's descriptors, waits for the entries it depends on (ctu_wait[]: lower tickets), and runs the same body as a level launch.
 * A ticket is only ever held by a workgroup that is RUNNING, and a wait only points at lower tickets — drawn earlier, by running
 * workgroups, or finished — so the lowest unfinished ticket never waits: progress depends neither on the order in which the hardware
 * dispatches workgroups nor on what other streams' kernels hold on the chip.  (With entry = blockIdx two such launches on two
 * streams deadlocked: each filled the CUs the other's not-yet-dispatched low ids needed — seen with six streams.  Persistent
 * workgroups looping over tickets are as safe but keep their CUs until the queue is empty: the other streams' kernels, which fill
 * the gaps of this latency-bound pass, lost more than the loop saved.)  No level launches, no launch as long as its slowest CTU: an
 * I picture costs its dependency chain at the CTUs' own lengths and a B picture's handful of levels overlap.
 * ======================================================================================= */
template <typename PX, bool CIP, bool STAGED>
__global__ __launch_bounds__(64 * INTRA_MAX_WAVES) __attribute__((amdgpu_waves_per_eu(6, 8))) void intra_dag_kernel(const OhBatch B, const OhIntraLaunch L, const int n_pics, const uint32_t total, uint32_t *__restrict__ ticket, const uint32_t spin_limit)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ uint32_t s_ticket;
    /* a counter per picture (waits never cross pictures), each on a cache line of its own: one word takes ~88 atomics per
     * microsecond, a B-picture batch draws 24 k tickets */
    const uint32_t pic = blockIdx.x % (uint32_t)n_pics;
    if (threadIdx.x == 0) s_ticket = atomicAdd(ticket + OH_TICKET_STRIDE * pic, 1u);
    __syncthreads();
    const uint32_t k = s_ticket;
    const DevFrame *__restrict__ f = B.f[pic];
    (void)total;
    if (k < f->n_ictu)
        intra_ctu_body<PX, CIP, STAGED, true>(f, L, k, smem, spin_limit);
}

/* =========================================================================================
 * The same dependency graph for pictures whose CTUs hold FEW intra blocks (B pictures: a third of the CTUs, a third of their
 * samples): ONE WAVE per schedule entry, nothing staged.  A block gathers its neighbours from the picture in HBM and stores its
 * samples there; the wave runs the entry's blocks in schedule order (sub-level by sub-level — any topological order would do:
 * vector memory operations of one wave reach the L1 / L2 in program order, so a block sees what the blocks before it stored),
 * four <= 8x8 blocks per pass in 16-lane slots, bigger ones alone.  No LDS beyond the wave's edge arrays, no barrier, no
 * rectangle read or written back: the chip holds as many CTUs as it holds waves, and the launch moves the blocks' samples, their
 * edges and their descriptors instead of whole staged CTUs (the staged level launches of such pictures spent 42 % of their time on
 * the rectangles and ran 5 workgroups per CU: profiles/r02_intra_staging_experiment.txt).  Hand-off between entries as above.
 * ======================================================================================= */
template <typename PX, bool CIP>
__global__ __launch_bounds__(64) void intra_direct_kernel(const OhBatch B, const int n_pics, const uint32_t total, uint32_t *__restrict__ ticket, const uint32_t spin_limit, const int exp)
{
    /* per wave: the edge arrays, and a window of 64 block descriptors (a wave reads its entry's lists front to back: tables and
     * descriptors arrive 64 at a time, so a pass waits for ONE round trip to HBM — its samples and residual — not for four) */
    __shared__ IntraLds edges;
    __shared__ __attribute__((aligned(16))) DevIntra items_l[64];
    const int lane = threadIdx.x;
    const uint32_t pic = blockIdx.x % (uint32_t)n_pics;     /* the entry is the ticket the wave draws when it starts (see intra_dag_kernel) */
    uint32_t pos = 0;
    if (lane == 0) pos = atomicAdd(ticket + OH_TICKET_STRIDE * pic, 1u);
    pos = __builtin_amdgcn_readfirstlane(pos);
    const DevFrame *__restrict__ f = B.f[pic];
    (void)total;
    if (pos >= f->n_ictu)
        return;
    const uint32_t k = (exp & 4) ? pos : G_CONST(uint32_t, f->ctu_order)[pos];      /* chains first (prep_intra_order) */
    const DevIntraCtu ctu = gload(f->ictu + k);
    const uint32_t aux = __builtin_amdgcn_readfirstlane(G_CONST(uint32_t, f->ctu_aux)[k]);
    const GLOBAL uint32_t *__restrict__ ss = G_CONST(uint32_t, f->sub_start) + ctu.sub_first;
    const GLOBAL uint32_t *__restrict__ sm = G_CONST(uint32_t, f->sub_small) + ctu.sub_first;
    const GLOBAL uint4v *__restrict__ items_g = (const GLOBAL uint4v *)f->intra;
    const GLOBAL int16_t *__restrict__ res_pool = G_CONST(int16_t, f->res);
    const int bd = f->pp.bit_depth, n_sub = (int)ctu.n_sub;
    const uint32_t item0 = ctu.item0, item_end = item0 + (ctu.n_items & 0xffffu);
    const Samples<PX, true> S = { G_MUT(PX, f->cur.p[0]), 1 << (bd - 1), (aux & OH_AUX_AWAITED) != 0u };
    const PlaneGeo G = { (int)(((const char *)f->cur.p[1] - (const char *)f->cur.p[0]) / (ptrdiff_t)sizeof(PX)),
                         (int)(((const char *)f->cur.p[2] - (const char *)f->cur.p[0]) / (ptrdiff_t)sizeof(PX)), f->cur.stride[0], f->cur.stride[1] };
    /* descriptor window [ib, ib + 64): lane i brings item ib + i (clamped to the entry's last) */
    uint32_t ib = item0;
    auto window = [&](uint32_t first) {
        ib = first;
        const uint32_t i = min(first + (uint32_t)lane, item_end - 1u);
        const uint4v q0 = items_g[2 * i], q1 = items_g[2 * i + 1];
        WSYNC();                                             /* earlier passes are done with the old window */
        ((uint4v *)items_l)[2 * lane] = q0; ((uint4v *)items_l)[2 * lane + 1] = q1;
        WSYNC();
    };
    window(item0);
    if ((aux & OH_AUX_WAITS) && !(exp & 1))
        dag_wait_wave(f, k, lane, spin_limit, exp);
    SlotState st;
    for (int cb = 0; cb < n_sub; cb += 63) {                 /* sub-level tables: 64 entries of sub_start (63 sub-levels) per round trip, one per lane */
        const uint32_t t_ss = ss[min(cb + lane, n_sub)], t_sm = sm[min(cb + lane, n_sub - 1)];
        const int ce = min(cb + 63, n_sub);
        for (int s = cb; s < ce; s++) {
            const uint32_t s0 = __builtin_amdgcn_readlane(t_ss, s - cb), s1 = __builtin_amdgcn_readlane(t_ss, s - cb + 1);
            const uint32_t ns = CIP ? 0u : min((uint32_t)__builtin_amdgcn_readlane(t_sm, s - cb), s1 - s0);
            for (uint32_t u = 0; u < ns; u += 4) {
                const uint32_t first = s0 + u, cnt = min(4u, ns - u);
                if (first + cnt > ib + 64u) window(first);
                slots_prepare<false, true>(st, (const DevIntra *)items_l, first - ib, (int)cnt, G, nullptr, res_pool, lane);
                slots_finish<PX, true>(st, bd, edges.E, S, lane, nullptr);
            }
            for (uint32_t b = s0 + ns; b < s1; b++) {
                if (b >= ib + 64u) window(b);
                uint4v q0, q1;
                item_load((const DevIntra *)items_l, b - ib, q0, q1);
                intra_block<PX, CIP, false, true>(f, bd, q0, q1, edges, S, G, nullptr, lane, nullptr);
            }
        }
    }
    if ((aux & OH_AUX_AWAITED) && !(exp & 2)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) dag_publish_lane<true>(f, k);
    }
}

/* ctu_done[] of every picture of the batch back to zero: a work list may be executed more than once */
__global__ __launch_bounds__(256) void intra_dag_reset_kernel(const OhBatch B, uint32_t *__restrict__ tickets)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < f->n_ictu) f->ctu_done[k] = 0u;
    if (blockIdx.x == 0 && threadIdx.x < 2) tickets[(threadIdx.x * OH_MAX_BATCH + blockIdx.y) * OH_TICKET_STRIDE] = 0u;      /* counter blockIdx.y of the wave-per-CTU launch / of the staged launch */
}

/* =========================================================================================
 * The same pass for pictures whose CTUs nearly all depend on their neighbours (I pictures: 126 levels for 4K): instead of one
 * launch per wavefront level — each as long as its slowest CTU, plus launch and staging overhead, 126 times — ONE launch whose
 * workgroups are the CTU ROWS (what pthread_slice.c's wavefront threads are, hevc.c:2829-2990): row y walks its CTUs left to right
 * and starts CTU x when row y-1 has finished min(x + 2, ctbw) CTUs (ff_thread_await_progress2 / the two-CTB lag of WPP, a superset
 * of the dependencies the recorder found).  A picture then costs its critical path (ctbw + 2 (ctbh - 1) CTU times at the AVERAGE
 * CTU length) instead of the sum of the levels' maxima.
 *
 * Progress is a counter per row in HBM: the finishing workgroup makes its samples visible (release fence at agent scope behind a
 * barrier) and stores x + 1; the waiting workgroup polls it (one lane, relaxed agent-scope loads) and every wave then takes an acquire fence (L1 and the XCD's non-coherent L2 lines dropped) before it stages.  Workgroup
 * ids are row-major over (row, picture), rows ascending: a waiting workgroup only ever waits for a LOWER id, and the dispatcher of
 * each XCD hands workgroups out in id order, so the lowest unfinished workgroup is always resident and never waits on a
 * non-resident one — no deadlock whatever the occupancy (in-order dispatch is what the hardware does, not what HIP promises: see
 * engine.hip where the path is chosen).  The poll is bounded all the same: a wave that gives up latches OH_KE_ROW_TIMEOUT in the
 * engine's error word — every wait on the stream then fails with picture and row — and continues, so the grid always drains.
 * ======================================================================================= */
template <typename PX, bool CIP, bool STAGED>
__global__ __launch_bounds__(64 * INTRA_MAX_WAVES) void intra_rows_kernel(const OhBatch B, const OhIntraLaunch L, const int n_pics, const uint32_t spin_limit)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int row = blockIdx.x / n_pics, pic = blockIdx.x - row * n_pics;
    const DevFrame *__restrict__ f = B.f[pic];
    const OhPicParams &pp = f->pp;
    const int lc = pp.log2_ctb_size, ctbw = (pp.width + (1 << lc) - 1) >> lc, ctbh = (pp.height + (1 << lc) - 1) >> lc;
    if (row >= ctbh)
        return;
    uint32_t *__restrict__ progress = f->row_progress;
    const GLOBAL uint32_t *__restrict__ entry_of = G_CONST(uint32_t, f->ctu_seen);
    for (int x = 0; x < ctbw; x++) {
        const uint32_t e = entry_of[row * ctbw + x];                 /* index + 1 of the CTU's schedule entry, 0: no intra block */
        if (e) {
            if (row > 0) {
                const uint32_t need = (uint32_t)min(x + 2, ctbw);
                if (threadIdx.x == 0) {
                    uint32_t spin = 0;                                   /* the default limit is ~1 s: never reached while the rows above make progress */
                    while (__hip_atomic_load(&progress[row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {   /* the acquire is the fence below */
                        if (++spin >= spin_limit) {                      /* give up, but not silently: oh_engine_sync reports picture and row */
                            dag_latch_error(f, OH_KE_ROW_TIMEOUT, (uint32_t)row);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(8);
                    }
                }
                __syncthreads();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   /* samples written by other workgroups: not from this CU's L1 */
            }
            intra_ctu_body<PX, CIP, STAGED, false>(f, L, e - 1, smem, 0u);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       /* every thread's stores of the CTU */
            __syncthreads();                                         /* ... and the LDS is free for the next CTU */
        }
        if (threadIdx.x == 0)
            __hip_atomic_store(&progress[row], (uint32_t)x + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

__global__ __launch_bounds__(64) void intra_rows_reset_kernel(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.x];
    const int lc = f->pp.log2_ctb_size, ctbh = (f->pp.height + (1 << lc) - 1) >> lc;
    for (int r = threadIdx.x; r < ctbh; r += 64) f->row_progress[r] = 0;
}

/* =========================================================================================
 * launcher
 * ======================================================================================= */
int ohk_init_intra(void)
{
    /* the intra kernel's LDS block is sized per launch and exceeds 64 KiB for 4:4:4 CTUs full of 4x4 blocks */
    const int max_lds = 128 * 1024;
    const void *intra_kernels[8] = {
        (const void *)intra_ctu_kernel<uint8_t, false, false>, (const void *)intra_ctu_kernel<uint8_t, false, true>,
        (const void *)intra_ctu_kernel<uint8_t, true, false>, (const void *)intra_ctu_kernel<uint8_t, true, true>,
        (const void *)intra_ctu_kernel<uint16_t, false, false>, (const void *)intra_ctu_kernel<uint16_t, false, true>,
        (const void *)intra_ctu_kernel<uint16_t, true, false>, (const void *)intra_ctu_kernel<uint16_t, true, true> };
    const void *row_kernels[8] = {
        (const void *)intra_rows_kernel<uint8_t, false, false>, (const void *)intra_rows_kernel<uint8_t, false, true>,
        (const void *)intra_rows_kernel<uint8_t, true, false>, (const void *)intra_rows_kernel<uint8_t, true, true>,
        (const void *)intra_rows_kernel<uint16_t, false, false>, (const void *)intra_rows_kernel<uint16_t, false, true>,
        (const void *)intra_rows_kernel<uint16_t, true, false>, (const void *)intra_rows_kernel<uint16_t, true, true> };
    for (const void *k : intra_kernels)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess)
            return -1;
    for (const void *k : row_kernels)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess)
            return -1;
    const void *dag_kernels[8] = {
        (const void *)intra_dag_kernel<uint8_t, false, false>, (const void *)intra_dag_kernel<uint8_t, false, true>,
        (const void *)intra_dag_kernel<uint8_t, true, false>, (const void *)intra_dag_kernel<uint8_t, true, true>,
        (const void *)intra_dag_kernel<uint16_t, false, false>, (const void *)intra_dag_kernel<uint16_t, false, true>,
        (const void *)intra_dag_kernel<uint16_t, true, false>, (const void *)intra_dag_kernel<uint16_t, true, true> };
    for (const void *k : dag_kernels)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, max_lds) != hipSuccess)
            return -1;
    return 0;
}

/* ctu_done[] cleared for the n pictures of B (max_ictu: the largest schedule among them) */
extern "C" void ohk_intra_dag_reset(const OhBatch *B, int n, uint32_t max_ictu, uint32_t *tickets, hipStream_t st)
{
    if (n <= 0 || !max_ictu) return;
    hipLaunchKernelGGL(intra_dag_reset_kernel, dim3((max_ictu + 255) / 256, n), dim3(256), 0, st, *B, tickets);
}

/* n pictures, each one's whole schedule, staged form: l = the LDS carve-up that fits every CTU of all of them (l->level unused) */
extern "C" void ohk_intra_dag(const OhBatch *B, int n, const OhPicParams *p, const OhIntraLaunch *l, uint32_t max_ictu, uint32_t *ticket,
                              uint32_t spin_limit, hipStream_t st)
{
    if (n <= 0 || !max_ictu) return;
    const uint32_t total = max_ictu * (uint32_t)n;
    dim3 g(total), b(64 * l->waves);
#define DAG_LAUNCH(PX, CIP, ST) hipLaunchKernelGGL(HIP_KERNEL_NAME(intra_dag_kernel<PX, CIP, ST>), g, b, l->lds_bytes, st, *B, *l, n, total, ticket, spin_limit)
#define DAG_BY_FLAGS(PX)                                                                       \
    do {                                                                                       \
        if (p->constrained_intra_pred) { if (l->staged) DAG_LAUNCH(PX, true, true); else DAG_LAUNCH(PX, true, false); }   \
        else                           { if (l->staged) DAG_LAUNCH(PX, false, true); else DAG_LAUNCH(PX, false, false); } \
    } while (0)
    if (p->bit_depth == 8) DAG_BY_FLAGS(uint8_t); else DAG_BY_FLAGS(uint16_t);
#undef DAG_BY_FLAGS
#undef DAG_LAUNCH
}

/* n pictures, each one's whole schedule, a wave per CTU straight on the picture in HBM */
extern "C" void ohk_intra_direct(const OhBatch *B, int n, const OhPicParams *p, uint32_t max_ictu, uint32_t *ticket, uint32_t spin_limit, hipStream_t st)
{
    if (n <= 0 || !max_ictu) return;
    const uint32_t total = max_ictu * (uint32_t)n;
    dim3 g(total), b(64);
    static const char *xenv = getenv("OHEVC_EXP");           /* experiments only: 1 no waits, 2 no publishing (wrong pictures) */
    const int exp = xenv ? atoi(xenv) : 0;
#define DIRECT_LAUNCH(PX, CIP) hipLaunchKernelGGL(HIP_KERNEL_NAME(intra_direct_kernel<PX, CIP>), g, b, 0, st, *B, n, total, ticket, spin_limit, exp)
    if (p->bit_depth == 8) { if (p->constrained_intra_pred) DIRECT_LAUNCH(uint8_t, true); else DIRECT_LAUNCH(uint8_t, false); }
    else                   { if (p->constrained_intra_pred) DIRECT_LAUNCH(uint16_t, true); else DIRECT_LAUNCH(uint16_t, false); }
#undef DIRECT_LAUNCH
}

/* n pictures whose intra pass runs as CTU rows (intra_rows_kernel); l: the LDS carve-up that fits every CTU of all of them */
extern "C" void ohk_intra_rows(const OhBatch *B, int n, const OhPicParams *p, const OhIntraLaunch *l, uint32_t spin_limit, hipStream_t st)
{
    if (n <= 0) return;
    const int lc = p->log2_ctb_size, ctbh = (p->height + (1 << lc) - 1) >> lc;
    hipLaunchKernelGGL(intra_rows_reset_kernel, dim3(n), dim3(64), 0, st, *B);
    dim3 g(ctbh * n), b(64 * l->waves);
#define ROWS_LAUNCH(PX, CIP, ST) hipLaunchKernelGGL(HIP_KERNEL_NAME(intra_rows_kernel<PX, CIP, ST>), g, b, l->lds_bytes, st, *B, *l, n, spin_limit)
#define ROWS_BY_FLAGS(PX)                                                                      \
    do {                                                                                       \
        if (p->constrained_intra_pred) { if (l->staged) ROWS_LAUNCH(PX, true, true); else ROWS_LAUNCH(PX, true, false); }   \
        else                           { if (l->staged) ROWS_LAUNCH(PX, false, true); else ROWS_LAUNCH(PX, false, false); } \
    } while (0)
    if (p->bit_depth == 8) ROWS_BY_FLAGS(uint8_t); else ROWS_BY_FLAGS(uint16_t);
#undef ROWS_BY_FLAGS
#undef ROWS_LAUNCH
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
