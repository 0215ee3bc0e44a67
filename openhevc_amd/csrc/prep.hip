/*
 * prep.hip — work-list preparation on the GPU: the raw lists a decoder hands over (OhPu / OhTu / OhIntra ... exactly as
 * include/ohevc_frame.h defines them) are validated and turned into what the pass kernels consume, right behind the H2D copy
 * on the engine's copy stream.  The host side of a hand-over is a memcpy and a few counting loops; nothing here depends on
 * sample values, so it overlaps the passes of earlier pictures.
 *
 *   raw list                   kernel                 product (dev_frame.h)
 *   OhPu[] (+ the host's running block counts)  prep_pu_expand   DevMcJob lists: every plane rectangle of a PU cut into <= 8x8 blocks
 *                              prep_mc_group          bi-predicted blocks first inside every run of 64 (a wave runs the second
 *                                                     list when any of its four blocks has one)
 *   OhTu[], tu_sparse, tu_cross prep_tu_mark / _scatter  DevTu buckets by transform size, DevCross list, KEEP_RES marks
 *   OhIntra[], sub_start       prep_intra_sub         <= 8x8 blocks first inside a sub-level (four of them share a wave)
 *                              prep_intra_items       DevIntra: LDS offsets, edge sizes, filter / class flags, angles, the
 *                                                     constrained-intra masks (hevcpred_template.c:116-163)
 *   OhIntraCtu[], level_start  prep_intra_ctu / _levels  DevIntraCtu (residual span, staged rectangle); the per-level launch
 *                                                     statistics the host sizes the intra launches with (DevSummary)
 *
 * Every index a pass kernel will follow is checked HERE (what engine.hip's host loop used to do): the first violation is
 * latched in DevSummary.err, the remaining preparation kernels and — because oh_frames_execute reads the summary before it
 * launches anything — all passes are skipped for that work list, so a malformed list never faults the GPU.
 */
#include <string.h>
#include "kernels_common.h"

static __device__ __forceinline__ DevSummary *summary_of(const DevFrame *f) { return (DevSummary *)f->summary; }
static __device__ __forceinline__ bool failed(const DevFrame *f) { return *(volatile uint32_t *)&summary_of(f)->err != 0; }
static __device__ __noinline__ void fail(const DevFrame *f, uint32_t code, uint32_t item)
{
    DevSummary *s = summary_of(f);
    if (atomicCAS(&s->err, 0u, code) == 0u)
        s->err_item = item;
}

/* cursors, marks and the summary start from zero (one launch for the whole batch instead of a fill per list between the copies) */
__global__ __launch_bounds__(256) void prep_clear(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    GLOBAL uint4v *z = (GLOBAL uint4v *)f->zero_ptr;           /* 256-byte aligned, a multiple of 256 bytes long */
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < f->zero_words / 4; i += gridDim.x * blockDim.x)
        z[i] = uint4v{ 0, 0, 0, 0 };
}

/* ---------------------------------------------------------------- prediction units ---- */
static __device__ __forceinline__ bool pu_ok(const DevFrame *f, const OhPu &pu)
{
    const OhPicParams &p = f->pp;
    if (pu.w < 4 || pu.h < 4 || pu.w > 64 || pu.h > 64 || (pu.w & 3) || (pu.h & 3) || pu.x + pu.w > p.width || pu.y + pu.h > p.height ||
        (pu.x & 3) || (pu.y & 3))
        return false;
    if (pu.ref[0] == OH_NO_REF && pu.ref[1] == OH_NO_REF)
        return false;
    for (int l = 0; l < 2; l++)
        if (pu.ref[l] != OH_NO_REF && (pu.ref[l] >= OH_MAX_REFS || !((f->ref_ok >> pu.ref[l]) & 1)))
            return false;
    return pu.wp == OH_NO_WP || pu.wp < f->n_wp;
}
static __device__ __forceinline__ void pu_jobs(const DevFrame *f, const OhPu &pu, uint32_t &nl, uint32_t &nc)
{
    const OhPicParams &p = f->pp;
    const int hs = hsh(p, 1), vs = vsh(p, 1);
    nl = (uint32_t)(((pu.w + 7) >> 3) * ((pu.h + 7) >> 3));
    nc = p.chroma_format_idc ? 2u * (uint32_t)((((pu.w >> hs) + 7) >> 3) * (((pu.h >> vs) + 7) >> 3)) : 0u;
}

/* one plane's rectangle of a PU cut into the <= 8x8 blocks the MC kernel works on; a DevMcJob is written as its five dwords
 * (x | y, w | h | ref[0] | ref[1], mv[0], mv[1], wp | c_idx | flags) */
static __device__ __forceinline__ uint32_t emit_jobs(DevMcJob *__restrict__ out, uint32_t o, const OhPu &pu, int c, int hs, int vs)
{
    static_assert(sizeof(DevMcJob) == 20, "DevMcJob layout");
    const int first = pu.ref[0] != OH_NO_REF ? 0 : 1;
    const uint32_t ref0 = pu.ref[first], ref1 = first == 0 ? pu.ref[1] : (uint32_t)OH_NO_REF;
    const uint32_t w2 = (uint32_t)(uint16_t)pu.mv[first][0] | ((uint32_t)(uint16_t)pu.mv[first][1] << 16);
    const uint32_t w3 = (uint32_t)(uint16_t)pu.mv[1][0] | ((uint32_t)(uint16_t)pu.mv[1][1] << 16);
    const uint32_t w4 = (uint32_t)pu.wp | ((uint32_t)c << 16) | ((first ? (uint32_t)OH_MCF_FROM_L1 : 0u) << 24);
    const int x0 = pu.x >> hs, y0 = pu.y >> vs, w = pu.w >> hs, h = pu.h >> vs;
    for (int oy = 0; oy < h; oy += 8)
        for (int ox = 0; ox < w; ox += 8) {
            const uint32_t bw = (uint32_t)(w - ox < 8 ? w - ox : 8), bh = (uint32_t)(h - oy < 8 ? h - oy : 8);
            GLOBAL uint32_t *d = (GLOBAL uint32_t *)(out + o++);
            d[0] = (uint32_t)(x0 + ox) | ((uint32_t)(y0 + oy) << 16);
            d[1] = bw | (bh << 8) | (ref0 << 16) | (ref1 << 24);
            d[2] = w2; d[3] = w3; d[4] = w4;
        }
    return o;
}

__global__ __launch_bounds__(256) void prep_pu_expand(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f->n_pu || failed(f))
        return;
    const OhPu pu = gload(f->pu + i);
    const OhPicParams &p = f->pp;
    /* the host counted the blocks of every PU with pu_jobs()' arithmetic while sizing the lists and hands the running sums
     * over (pu_off): a PU writes inside its own range whatever its fields hold, and an invalid one writes nothing */
    uint32_t nl, nc;
    pu_jobs(f, pu, nl, nc);
    if (!pu_ok(f, pu) || f->pu_off[i + 1] - f->pu_off[i] != nl || f->pu_off[f->n_pu + 2 + i] - f->pu_off[f->n_pu + 1 + i] != nc ||
        f->pu_off[i + 1] > f->n_mc_luma || f->pu_off[f->n_pu + 2 + i] > f->n_mc_chroma) {
        fail(f, OH_PE_PU, i);
        return;
    }
    emit_jobs((DevMcJob *)f->mc_luma, f->pu_off[i], pu, 0, 0, 0);
    if (p.chroma_format_idc) {
        uint32_t o = f->pu_off[f->n_pu + 1 + i];
        o = emit_jobs((DevMcJob *)f->mc_chroma, o, pu, 1, hsh(p, 1), vsh(p, 1));
        emit_jobs((DevMcJob *)f->mc_chroma, o, pu, 2, hsh(p, 1), vsh(p, 1));
    }
}

/* stable partition of every run of 64 blocks: the bi-predicted ones first.  A wave loads its run, then stores it permuted
 * (all loads of the wave precede its stores in program order, so the run is rearranged in place). */
__global__ __launch_bounds__(64) void prep_mc_group(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    if (failed(f))
        return;
    const uint32_t runs_l = (f->n_mc_luma + 63) >> 6;
    const bool luma = blockIdx.x < runs_l;
    const uint32_t run = luma ? blockIdx.x : blockIdx.x - runs_l, n = luma ? f->n_mc_luma : f->n_mc_chroma;
    if ((uint64_t)run * 64 >= n)
        return;
    DevMcJob *__restrict__ jobs = (DevMcJob *)(luma ? f->mc_luma : f->mc_chroma) + (size_t)run * 64;
    const int lane = threadIdx.x, cnt = (int)min(64u, n - run * 64);
    const bool live = lane < cnt;
    uint32_t w[5] = { 0, 0, 0, 0, 0 };
    if (live) {
        const GLOBAL uint32_t *s = (const GLOBAL uint32_t *)(jobs + lane);
#pragma unroll
        for (int q = 0; q < 5; q++) w[q] = s[q];
    }
    const bool bi = live && ((w[1] >> 24) & 0xff) != OH_NO_REF;          /* DevMcJob.ref[1]: byte 7 */
    const unsigned long long mb = __builtin_amdgcn_ballot_w64(bi), ml = __builtin_amdgcn_ballot_w64(live && !bi);
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int dst = bi ? __builtin_popcountll(mb & below) : __builtin_popcountll(mb) + __builtin_popcountll(ml & below);
    __builtin_amdgcn_s_waitcnt(0);
    if (live) {
        GLOBAL uint32_t *d = (GLOBAL uint32_t *)(jobs + dst);
#pragma unroll
        for (int q = 0; q < 5; q++) d[q] = w[q];
    }
}

/* ---------------------------------------------------------------- transform blocks ---- */
/* everything about one transform block except the positions inside its sparse record (sparse_ok_*) */
static __device__ __forceinline__ bool tu_ok(const DevFrame *f, const OhTu &t, uint32_t i, uint32_t *rec_first = nullptr, uint32_t *rec_cnt = nullptr)
{
    const OhPicParams &p = f->pp;
    const int nplanes = p.chroma_format_idc ? 3 : 1;
    if (t.c_idx >= nplanes || t.log2_size < 2 || t.log2_size > 5 || t.kind > OH_TU_PCM)
        return false;
    const int n = 1 << t.log2_size;
    if (t.x + n > f->cur.w[t.c_idx] || t.y + n > f->cur.h[t.c_idx])
        return false;
    if ((uint64_t)t.coeff_off + (uint64_t)n * n > f->n_coeff || (t.coeff_off & 3) || (t.x & 3) || (t.y & 3))
        return false;
    if ((t.flags & OH_TUF_ROTATE) && t.log2_size != 2)
        return false;
    if (t.flags & OH_TUF_CROSS) {
        const uint32_t cw = f->tu_cross ? f->tu_cross[i] : OH_NO_COEFF, ty = cw & 0xffffff;
        if (cw == OH_NO_COEFF || p.chroma_format_idc != 3 || t.c_idx == 0 || ty >= f->n_tu || t.kind == OH_TU_PCM)
            return false;
        const OhTu y = gload(f->tu_raw + ty);
        if (y.c_idx != 0 || y.log2_size != t.log2_size)
            return false;
    }
    if (t.flags & OH_TUF_SPARSE) {
        if (!f->sparse || !f->tu_sparse || t.kind == OH_TU_BYPASS || t.kind == OH_TU_PCM)
            return false;
        const uint64_t so = f->tu_sparse[i];
        if (so >= f->n_sparse)
            return false;
        const uint32_t w0 = f->sparse[so], cnt = w0 & 0xffff, qp = (w0 >> 16) & 0xff, mid = w0 >> 24;
        if (cnt > (uint32_t)(n * n) || so + 1 + cnt > f->n_sparse || qp > 75 || (mid != OH_FLAT_MATRIX && (mid > 5 || !f->scaling)))
            return false;
        if (rec_first) { *rec_first = (uint32_t)so + 1; *rec_cnt = cnt; }
    } else if (!f->coeffs_present) {
        return false;                                               /* dense block, but no pool came with the list */
    }
    return true;
}

/* only for pictures with cross-component prediction (4:4:4 range extension): the luma blocks those chroma blocks read keep
 * their residual — marked before the buckets are written */
__global__ __launch_bounds__(256) void prep_tu_mark(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f->n_tu || failed(f))
        return;
    const OhTu t = gload(f->tu_raw + i);
    if (!(t.flags & OH_TUF_CROSS))
        return;
    if (tu_ok(f, t, i)) ((uint8_t *)f->tu_keep)[f->tu_cross[i] & 0xffffff] = 1;
    else fail(f, OH_PE_TU, i);
}

/* every block is validated and goes to its size bucket.  The bucket ranges are known (the host counted the sizes while it
 * looked for dense blocks; prep_finish compares); a workgroup reserves its share of each bucket with ONE atomic per size, its
 * waves take theirs in order, so blocks stay in list order inside a workgroup's share. */
__global__ __launch_bounds__(256) void prep_tu_scatter(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    __shared__ uint32_t wcnt[4][4], wbase[4];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < f->n_tu && !failed(f);               /* no early return: every wave reaches the two barriers */
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    OhTu t;
    int k = -1;
    uint32_t rec = 0, cnt = 0;
    if (live) {
        t = gload(f->tu_raw + i);
        if (tu_ok(f, t, i, &rec, &cnt)) k = t.log2_size - 2;
        else fail(f, OH_PE_TU, i);
    }
    /* positions of the sparse records (every one must lie inside its block: residual_kernel scatters to them).  A lane checks a
     * short record itself, loads issued back to back; the long ones (up to n * n entries) are taken by the whole wave, one at a time */
    {
        const GLOBAL uint32_t *__restrict__ sp = G_CONST(uint32_t, f->sparse);
        const uint32_t n2 = k >= 0 ? 16u << (2 * k) : 0u;
        uint32_t bad = 0;
        if (k >= 0 && cnt <= 16)
            for (uint32_t q = 0; q < cnt; q++) bad |= (sp[rec + q] & 0xffff) >= n2;
        unsigned long long big = __builtin_amdgcn_ballot_w64(k >= 0 && cnt > 16);
        while (big) {
            const int src = __builtin_ctzll(big);
            big &= big - 1;
            const uint32_t r0 = __shfl(rec, src), rc = __shfl(cnt, src), rn2 = __shfl(n2, src);
            uint32_t b = 0;
            for (uint32_t q = lane; q < rc; q += 64) b |= (sp[r0 + q] & 0xffff) >= rn2;
            if (__builtin_amdgcn_ballot_w64(b != 0) && lane == src) bad = 1;
        }
        if (bad) { fail(f, OH_PE_TU, i); k = -1; }
    }
    unsigned long long m[4];
#pragma unroll
    for (int s = 0; s < 4; s++) {
        m[s] = __builtin_amdgcn_ballot_w64(k == s);
        if (lane == 0) wcnt[s][wave] = (uint32_t)__builtin_popcountll(m[s]);
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        uint32_t tot = 0;
        for (int w = 0; w < nw; w++) tot += wcnt[threadIdx.x][w];
        wbase[threadIdx.x] = tot ? atomicAdd(&f->tu_cursor[threadIdx.x], tot) : 0u;
    }
    __syncthreads();
    if (k < 0)
        return;
    uint32_t pos = f->tu_first[k] + wbase[k];
    for (int w = 0; w < wave; w++) pos += wcnt[k][w];
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    const unsigned long long mk = k == 0 ? m[0] : k == 1 ? m[1] : k == 2 ? m[2] : m[3];
    pos += (uint32_t)__builtin_popcountll(mk & below);
    if (pos >= f->n_tu) { fail(f, OH_PE_TU, i); return; }                     /* more blocks of a size than the host counted */
    alignas(16) DevTu d;
    d.t = t;
    if (f->tu_keep[i]) d.t.flags |= OH_TUF_KEEP_RES;
    d.sparse_off = (t.flags & OH_TUF_SPARSE) ? f->tu_sparse[i] : OH_NO_COEFF;
    uint4v q;
    __builtin_memcpy(&q, &d, 16);
    *(GLOBAL uint4v *)((DevTu *)f->tu + pos) = q;
    if (t.flags & OH_TUF_CROSS) {
        /* cross-component prediction: the chroma block is finished by cross_kernel once every inverse transform of the picture is done */
        const uint32_t cw = f->tu_cross[i];
        const OhTu ty = gload(f->tu_raw + (cw & 0xffffff));
        alignas(16) DevCross dc;
        dc.x = t.x; dc.y = t.y; dc.c_idx = t.c_idx; dc.log2_size = t.log2_size; dc.flags = t.flags;
        dc.scale = (int8_t)(cw >> 24); dc.res_c = t.coeff_off; dc.res_y = ty.coeff_off;
        const uint32_t o = atomicAdd(&f->tu_cursor[9], 1u);
        if (o >= f->n_cross) { fail(f, OH_PE_TU, i); return; }
        __builtin_memcpy(&q, &dc, 16);
        *(GLOBAL uint4v *)((DevCross *)f->cross + o) = q;
    }
}

/* ---------------------------------------------------------------- intra blocks ---- */
/* inside a sub-level the blocks are independent: the <= 8x8 ones go first — the intra kernel runs four of them per wave —
 * and sub_small says how many there are (none with constrained intra pred, which needs the one-block path) */
__global__ __launch_bounds__(256) void prep_intra_sub(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= f->n_sub || failed(f))
        return;
    const uint32_t b0 = f->sub_start[j], b1 = f->sub_start[j + 1];
    if (b0 > b1 || b1 > f->n_intra || (j == 0 && b0 != 0) || (j + 1 == f->n_sub && b1 != f->n_intra)) {
        fail(f, OH_PE_INTRA_TABLES, j);
        return;
    }
    uint32_t ns = 0;
    for (uint32_t b = b0; b < b1; b++)
        ns += gload(f->intra_raw + b).log2_size <= 3;
    uint32_t ws = b0, wb = b0 + ns;
    for (uint32_t b = b0; b < b1; b++)
        f->intra_perm[b] = gload(f->intra_raw + b).log2_size <= 3 ? ws++ : wb++;
    f->sub_small_w[j] = f->pp.constrained_intra_pred ? 0u : ns;
}

__constant__ int8_t c_angle[33] = { 32, 26, 21, 17, 13, 9, 5, 2, 0, -2, -5, -9, -13, -17, -21, -26, -32,
                                    -26, -21, -17, -13, -9, -5, -2, 0, 2, 5, 9, 13, 17, 21, 26, 32 };      /* intraPredAngle, H.265 table 8-4 */
__constant__ int16_t c_inv_angle[15] = { -4096, -1638, -910, -630, -482, -390, -315, -256, -315, -390, -482, -630, -910, -1638, -4096 };

/* one block: OhIntra -> DevIntra, everything that depends only on geometry and mode resolved (what intra_pred() derives per call,
 * hevcpred_template.c:73-163, 288-294) */
__global__ __launch_bounds__(256) void prep_intra_items(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= f->n_intra || failed(f))
        return;
    const OhPicParams &p = f->pp;
    const OhIntra it = gload(f->intra_raw + i);
    const int nplanes = p.chroma_format_idc ? 3 : 1;
    if (it.c_idx >= nplanes || it.log2_size < 2 || it.log2_size > 5 || it.mode > 34) { fail(f, OH_PE_INTRA, i); return; }
    const int c = it.c_idx, hs = hsh(p, c), vs = vsh(p, c), log2 = it.log2_size, n = 1 << log2, mode = it.mode;
    const int pw = f->cur.w[c], ph = f->cur.h[c];
    if (it.x + n > pw || it.y + n > ph ||
        ((it.avail & (OH_AV_LEFT | OH_AV_BOTTOM_LEFT | OH_AV_UP_LEFT)) && it.x == 0) ||
        ((it.avail & (OH_AV_UP | OH_AV_UP_RIGHT | OH_AV_UP_LEFT)) && it.y == 0) ||
        ((it.avail & OH_AV_UP_RIGHT) && it.x + n >= pw) || ((it.avail & OH_AV_BOTTOM_LEFT) && it.y + n >= ph)) { fail(f, OH_PE_INTRA, i); return; }
    alignas(16) DevIntra d;
    __builtin_memset(&d, 0, sizeof(d));
    d.res_off = OH_NO_COEFF;
    if (it.tu != OH_NO_COEFF) {
        if (it.tu >= f->n_tu) { fail(f, OH_PE_INTRA, i); return; }
        const OhTu t = gload(f->tu_raw + it.tu);
        if (t.c_idx != it.c_idx || t.x != it.x || t.y != it.y || t.log2_size != it.log2_size || (t.flags & OH_TUF_ADD_NOW)) { fail(f, OH_PE_INTRA, i); return; }
        d.res_off = t.coeff_off;
    }
    const int lc = p.log2_ctb_size, rs = ((1 << lc) >> hs) + 4;
    const OhCtuAreas areas = oh_ctu_areas(lc, p.chroma_format_idc);
    const int lx = it.x - ((((it.x << hs) >> lc) << lc) >> hs), ly = it.y - ((((it.y << vs) >> lc) << lc) >> vs);
    const uint32_t a_main = c == 0 ? areas.main[0] : c == 1 ? areas.main[1] : areas.main[2];
    const uint32_t a_top = c == 0 ? areas.top[0] : c == 1 ? areas.top[1] : areas.top[2];
    d.x = it.x; d.y = it.y; d.c_idx = it.c_idx; d.log2_size = it.log2_size; d.mode = it.mode; d.avail = it.avail;
    d.rs = (uint16_t)rs;
    d.cm_off = (uint16_t)(a_main + ly * rs + lx + 4);
    d.top_off = (uint16_t)(ly == 0 ? a_top + lx + 4 : d.cm_off - rs);
    const int tr = (it.x + 2 * n < pw ? it.x + 2 * n : pw) - (it.x + n);
    const int bl = (it.y + 2 * n < ph ? it.y + 2 * n : ph) - (it.y + n);
    d.tr_size = (uint8_t)(tr < 0 ? 0 : tr); d.bl_size = (uint8_t)(bl < 0 ? 0 : bl);
    int flags = 0, cls;
    if (!p.intra_smoothing_disabled && (c == 0 || p.chroma_format_idc == 3) && mode != 1 && n != 4) {   /* :288-294 */
        const int d26 = mode > 26 ? mode - 26 : 26 - mode, d10 = mode > 10 ? mode - 10 : 10 - mode;
        const int thresh = log2 == 3 ? 7 : log2 == 4 ? 1 : 0;
        if ((d26 < d10 ? d26 : d10) > thresh) {
            flags |= OH_IF_FILTER;
            if (p.strong_intra_smoothing && c == 0 && log2 == 5) flags |= OH_IF_STRONG_CAND;
        }
    }
    if (c == 0 && n < 32 && (mode == 1 || mode == 10 || mode == 26)) flags |= OH_IF_EDGE;     /* :410-416, :474-477, :501-508 */
    if (mode == 0) cls = OH_IC_PLANAR;
    else if (mode == 1) cls = OH_IC_DC;
    else if (mode == 26) cls = OH_IC_PURE_V;
    else if (mode == 10) cls = OH_IC_PURE_H;
    else cls = mode >= 18 ? OH_IC_ANG_V : OH_IC_ANG_H;
    if (mode >= 2) {
        d.angle = c_angle[mode - 2];
        if (d.angle < 0 && ((n * d.angle) >> 5) < -1) d.inv_angle = c_inv_angle[mode - 11];
    }
    if (p.constrained_intra_pred) {
        /* hevcpred_template.c:116-163: candidates that lie in inter CUs do not count; the kernel's slow path then
         * patches the gathered edges from the per-group intra masks (:185-249) */
        const GLOBAL uint8_t *map = G_CONST(uint8_t, f->is_intra);
        const int lpu = p.log2_min_pu_size, mpw = p.width >> lpu, mph = p.height >> lpu;
        const int X0 = it.x << hs, Y0 = it.y << vs, sl_h = n << hs, sl_v = n << vs;
        auto cell = [&](int px, int py) { return px >= 0 && py >= 0 && px < mpw && py < mph && map[px + py * mpw] != 0; };
        auto isi = [&](int dx, int dy) { return cell((X0 + dx * (1 << hs)) >> lpu, (Y0 + dy * (1 << vs)) >> lpu); };
        int pu_v = sl_v >> lpu, pu_h = sl_h >> lpu, av = it.avail;
        const bool on_x = !(X0 & ((1 << lpu) - 1)), on_y = !(Y0 & ((1 << lpu) - 1));
        if (!pu_h) pu_h++;
        auto any2 = [&](int px, int py, int dx, int dy, int cnt) { bool r = false; for (int q = 0; q < cnt; q += 2) r |= cell(px + q * dx, py + q * dy); return r; };
        if ((av & OH_AV_BOTTOM_LEFT) && on_x) {
            const int yb = (Y0 + sl_v) >> lpu;
            if (!any2((X0 - 1) >> lpu, yb, 0, 1, min(pu_v, mph - yb))) av &= ~OH_AV_BOTTOM_LEFT;
        }
        if ((av & OH_AV_LEFT) && on_x) {
            const int yl = Y0 >> lpu;
            if (!any2((X0 - 1) >> lpu, yl, 0, 1, min(pu_v, mph - yl))) av &= ~OH_AV_LEFT;
        }
        if ((av & OH_AV_UP_LEFT) && !cell((X0 - 1) >> lpu, (Y0 - 1) >> lpu)) av &= ~OH_AV_UP_LEFT;
        if ((av & OH_AV_UP) && on_y) {
            const int xt = X0 >> lpu;
            if (!any2(xt, (Y0 - 1) >> lpu, 1, 0, min(pu_h, mpw - xt))) av &= ~OH_AV_UP;
        }
        if ((av & OH_AV_UP_RIGHT) && on_y) {
            const int xr = (X0 + sl_h) >> lpu;
            if (!any2(xr, (Y0 - 1) >> lpu, 1, 0, min(pu_h, mpw - xr))) av &= ~OH_AV_UP_RIGHT;
        }
        d.avail = (uint8_t)av;
        unsigned lm = 0, tm = 0;
        for (int k = 0; 4 * k < 2 * n; k++) {
            if (it.x > 0 && isi(-1, 4 * k)) lm |= 1u << k;
            if (it.y > 0 && isi(4 * k, -1)) tm |= 1u << k;
        }
        d.cip_left = (uint16_t)lm; d.cip_top = (uint16_t)tm;
        flags |= OH_IF_CIP;
        if (it.x > 0 && it.y > 0 && isi(-1, -1)) flags |= OH_IF_CIP_CORNER;
    }
    d.flags = (uint8_t)(flags | (cls << 4));
    GLOBAL uint4v *o = (GLOBAL uint4v *)((DevIntra *)f->intra + f->intra_perm[i]);
    uint4v q[2];
    static_assert(sizeof(DevIntra) == 32, "DevIntra layout");
    __builtin_memcpy(q, &d, 32);
    o[0] = q[0]; o[1] = q[1];
}

/* One wave per entry of the schedule (a CTU that holds intra blocks): the entry is checked (the level -> ictu[] -> sub_start[]
 * -> intra[] nesting the intra kernel follows blindly), its residual span and the rectangle to stage are reduced over its
 * blocks, lanes taking the blocks in turn; ctu_aux[k] keeps what the level statistics need of it. */
__global__ __launch_bounds__(64) void prep_intra_ctu(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    const uint32_t k = blockIdx.x;
    if (k >= f->n_ictu || failed(f))
        return;
    const OhPicParams &p = f->pp;
    const int lane = threadIdx.x;
    const int lc = p.log2_ctb_size, ctbw = (p.width + (1 << lc) - 1) >> lc, ctbh = (p.height + (1 << lc) - 1) >> lc;
    DevIntra *__restrict__ di = (DevIntra *)f->intra;
    const OhIntraCtu c = gload(f->ictu_raw + k);
    uint32_t expect = 0;
    if (k) { const OhIntraCtu pc = gload(f->ictu_raw + k - 1); expect = pc.sub_first + pc.n_sub; }
    if (c.sub_first != expect || !c.n_sub || (uint64_t)c.sub_first + c.n_sub > f->n_sub || (k + 1 == f->n_ictu && c.sub_first + c.n_sub != f->n_sub) ||
        c.ctu >= (uint32_t)(ctbw * ctbh) || c.n_sub > OH_MAX_CTU_BLOCKS) { if (!lane) fail(f, OH_PE_INTRA_TABLES, k); return; }
    const uint32_t b0 = f->sub_start[c.sub_first], b1 = f->sub_start[c.sub_first + c.n_sub];
    if (b1 < b0 || b1 - b0 > OH_MAX_CTU_BLOCKS) { if (!lane) fail(f, OH_PE_INTRA_TABLES, k); return; }
    if (!lane && atomicExch(&f->ctu_seen[c.ctu], k + 1) != 0u) fail(f, OH_PE_INTRA_TABLES, k);   /* a CTU heads ONE entry: its level is one number */
    unsigned long long lo = ~0ull, hi = 0;
    int bx0 = 1 << 14, bx1 = -(1 << 14), by0 = 1 << 14, by1 = -(1 << 14);
    int any_res = 0, bad = 0, dep = 0;                        /* dep: neighbour CTUs a block gathers from, bit 0 left, 1 up-left, 2 up, 3 up-right */
    uint32_t area = 0;                                        /* samples the CTU's blocks cover, all planes */
    for (uint32_t b = b0 + lane; b < b1; b += 64) {
        const uint4v q0 = *(const GLOBAL uint4v *)(di + b);
        const int x = q0[0] & 0xffff, y = q0[0] >> 16, ci = q0[1] & 0xff, log2 = (q0[1] >> 8) & 0xff, n = 1 << log2, av = q0[1] >> 24;
        const uint32_t res_off = q0[2];
        area += 1u << (2 * log2);
        const int hs = hsh(p, ci), vs = vsh(p, ci);
        if ((uint32_t)((((y << vs) >> lc) * ctbw) + ((x << hs) >> lc)) != c.ctu) bad = 1;       /* the CTU is written back from c.ctu's origin */
        const int lx = x - ((((x << hs) >> lc) << lc) >> hs), ly = y - ((((y << vs) >> lc) << lc) >> vs);
        /* which neighbour CTUs hold the samples the block gathers (hevcpred_template.c:164-183): a superset by geometry (the
         * recorder's levels count intra samples only; an entry never waits for a CTU of its own or a higher level) */
        if (lx == 0 && ((av & (OH_AV_LEFT | OH_AV_BOTTOM_LEFT)) || ((av & OH_AV_UP_LEFT) && ly > 0))) dep |= 1;
        if (lx == 0 && ly == 0 && (av & OH_AV_UP_LEFT)) dep |= 2;
        if (ly == 0 && ((av & OH_AV_UP) || ((av & OH_AV_UP_LEFT) && lx > 0) || ((av & OH_AV_UP_RIGHT) && lx + n < ((1 << lc) >> hs)))) dep |= 4;
        if (ly == 0 && (av & OH_AV_UP_RIGHT) && lx + n >= ((1 << lc) >> hs)) dep |= 8;
        bx0 = min(bx0, (lx - 1) * (1 << hs)); bx1 = max(bx1, (lx + 2 * n) << hs);
        by0 = min(by0, (ly - 1) * (1 << vs)); by1 = max(by1, (ly + 2 * n) << vs);
        if (res_off == OH_NO_COEFF)
            continue;
        any_res = 1;
        lo = min(lo, (unsigned long long)res_off); hi = max(hi, (unsigned long long)res_off + (1u << (2 * log2)));
    }
    uint32_t slot_passes = 0;                                  /* wave passes of the CTU: groups of four <= 8x8 blocks + the bigger blocks */
    for (uint32_t j = c.sub_first + lane; j < c.sub_first + c.n_sub; j += 64) {
        const uint32_t sm = f->sub_small_w[j];
        slot_passes += (sm + 3) / 4 + (f->sub_start[j + 1] - f->sub_start[j] - sm);
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) {
        lo = min(lo, (unsigned long long)__shfl_xor(lo, d)); hi = max(hi, (unsigned long long)__shfl_xor(hi, d));
        bx0 = min(bx0, __shfl_xor(bx0, d)); bx1 = max(bx1, __shfl_xor(bx1, d));
        by0 = min(by0, __shfl_xor(by0, d)); by1 = max(by1, __shfl_xor(by1, d));
        any_res |= __shfl_xor(any_res, d); bad |= __shfl_xor(bad, d); dep |= __shfl_xor(dep, d);
        slot_passes += __shfl_xor((int)slot_passes, d);
        area += (uint32_t)__shfl_xor((int)area, d);
    }
    if (bad) { if (!lane) fail(f, OH_PE_INTRA_TABLES, k); return; }
    uint32_t res_lo = 0, res_cnt = 0;
    if (hi > lo && (lo & 3) == 0 && hi - lo <= 3u * 64 * 64) {
        /* the CTU's residual blocks lie together in the pool (a recorder appends them CTU by CTU): stageable in LDS */
        res_lo = (uint32_t)lo;
        res_cnt = (uint32_t)((hi - lo + 3) & ~3ull);
        if ((unsigned long long)res_lo + res_cnt > f->n_coeff) res_cnt = (uint32_t)(hi - lo) & ~3u;
    }
    if (res_cnt)
        for (uint32_t b = b0 + lane; b < b1; b += 64) {
            const uint32_t ro = *((const GLOBAL uint32_t *)(di + b) + 2);
            if (ro != OH_NO_COEFF) *((GLOBAL uint32_t *)(di + b) + 6) = ro - res_lo;       /* DevIntra.res_lds */
        }
    if (!lane) {
        alignas(16) DevIntraCtu d;
        d.sub_first = c.sub_first; d.n_sub = c.n_sub; d.ctu = c.ctu; d.item0 = b0; d.n_items = (b1 - b0) | min((area + 63) >> 6, 0xffffu) << 16;
        d.res_lo = res_lo; d.res_cnt = res_cnt;
        d.bx0 = (int16_t)bx0; d.bx1 = (int16_t)bx1; d.by0 = (int16_t)by0; d.by1 = (int16_t)by1;
        uint4v q[2];
        static_assert(sizeof(DevIntraCtu) == 32, "DevIntraCtu layout");
        __builtin_memcpy(q, &d, 32);
        GLOBAL uint4v *o = (GLOBAL uint4v *)((DevIntraCtu *)f->ictu + k);
        o[0] = q[0]; o[1] = q[1];
        f->ctu_aux[k] = min(slot_passes, (uint32_t)OH_AUX_PASSES) | (uint32_t)dep << OH_AUX_DEP_SHIFT | (any_res && !res_cnt ? (uint32_t)OH_AUX_RES_SCATTERED : 0u);
        DevSummary *sm = summary_of(f);
        atomicAdd(&sm->intra_area64, (area + 63) >> 6);
        atomicMax(&sm->max_passes, slot_passes);
    }
}

/* one wave per wavefront level: the launch statistics of the level, reduced over its CTUs into the summary */
__global__ __launch_bounds__(64) void prep_intra_levels(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    const uint32_t l = blockIdx.x;
    if (l >= f->n_levels || failed(f))
        return;
    const uint32_t k0 = f->lvl_start[l], k1 = f->lvl_start[l + 1];         /* the host checked the level table (it needs it) */
    uint32_t max_items = 1, max_sub = 1, max_res = 0, staged = 1;
    unsigned long long sum_items = 0, sum_sub = 0;
    for (uint32_t k = k0 + threadIdx.x; k < k1; k += 64) {
        const DevIntraCtu d = gload(f->ictu + k);
        const uint32_t aux = f->ctu_aux[k];
        f->ctu_lvl[k] = l;
        if (aux & OH_AUX_RES_SCATTERED) staged = 0;
        max_items = max(max_items, min(d.n_items & 0xffffu, (uint32_t)OH_MAX_CTU_BLOCKS));
        max_sub = max(max_sub, min((uint32_t)d.n_sub, (uint32_t)OH_MAX_CTU_BLOCKS));
        max_res = max(max_res, d.res_cnt);
        sum_items += aux & OH_AUX_PASSES;
        sum_sub += d.n_sub;
    }
#pragma unroll
    for (int d = 32; d; d >>= 1) {
        max_items = max(max_items, (uint32_t)__shfl_xor((int)max_items, d));
        max_sub = max(max_sub, (uint32_t)__shfl_xor((int)max_sub, d));
        max_res = max(max_res, (uint32_t)__shfl_xor((int)max_res, d));
        staged = min(staged, (uint32_t)__shfl_xor((int)staged, d));
        sum_items += __shfl_xor(sum_items, d);
        sum_sub += __shfl_xor(sum_sub, d);
    }
    if (threadIdx.x == 0) {
        DevLevelStat s;
        s.n_ctu = k1 - k0; s.max_items = max_items; s.max_sub = max_sub; s.max_res = max_res; s.staged = staged; s.pad = 0;
        s.sum_items = sum_items; s.sum_sub = sum_sub;
        *((DevLevelStat *)(summary_of(f) + 1) + l) = s;
    }
}

/* The schedule as a dependency graph, for the one-launch forms of the intra pass (intra.hip): one thread per entry looks up the
 * entries of the neighbour CTUs its blocks gather from; it waits for those of a LOWER level (the level table is the contract: a
 * block never reads intra samples of a CTU of its own or a higher level), which sit at lower indices of ictu[] — the one-launch
 * kernels dispatch in index order, so a wait always points at a workgroup dispatched earlier.  The awaited entry learns that it
 * has to publish (OH_AUX_AWAITED). */
__global__ __launch_bounds__(256) void prep_intra_wait(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.y];
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= f->n_ictu || failed(f))
        return;
    const OhPicParams &p = f->pp;
    const int lc = p.log2_ctb_size, ctbw = (p.width + (1 << lc) - 1) >> lc;
    const uint32_t aux = f->ctu_aux[k], my = f->ctu_lvl[k];
    const int ctu = (int)gload(f->ictu_raw + k).ctu, x = ctu % ctbw, y = ctu / ctbw;
    const int dep = (aux >> OH_AUX_DEP_SHIFT) & 15;
    uint4v w = { ~0u, ~0u, ~0u, ~0u };
    bool any = false;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int nx = x + (j == 3 ? 1 : j == 2 ? 0 : -1), ny = y - (j != 0);
        if (!(dep >> j & 1) || nx < 0 || ny < 0 || nx >= ctbw)
            continue;
        const uint32_t e = f->ctu_seen[ny * ctbw + nx];
        if (!e || f->ctu_lvl[e - 1] >= my)
            continue;
        w[j] = e - 1;
        any = true;
        atomicOr(&f->ctu_aux[e - 1], (uint32_t)OH_AUX_AWAITED);
    }
    *((GLOBAL uint4v *)f->ctu_wait + k) = w;
    if (any) atomicOr(&f->ctu_aux[k], (uint32_t)OH_AUX_WAITS);
}

/* Dispatch order of the wave-per-CTU form (intra_direct_kernel): the entries that wait or are awaited first, in schedule order (a
 * wait still points at a lower position), then the entries nothing depends on.  The chains — a B picture's few levels, each as long
 * as one CTU's latency — then start with the launch and the independent CTUs, the bulk, fill the chip around them; in plain schedule
 * order a chain's second link is not even dispatched before every independent CTU of the batch has been.  One workgroup per
 * picture: a stable partition by a block-wide scan. */
__global__ __launch_bounds__(256) void prep_intra_order(const OhBatch B)
{
    const DevFrame *__restrict__ f = B.f[blockIdx.x];
    const uint32_t n = f->n_ictu;
    if (!n || failed(f))
        return;
    __shared__ uint32_t cnt[256];
    const uint32_t t = threadIdx.x, per = (n + 255u) / 256u, b0 = min(t * per, n), b1 = min(b0 + per, n);
    uint32_t c = 0;
    for (uint32_t k = b0; k < b1; k++)
        c += (f->ctu_aux[k] & (OH_AUX_WAITS | OH_AUX_AWAITED)) != 0u;
    cnt[t] = c;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {                  /* inclusive scan */
        const uint32_t v = t >= d ? cnt[t - d] : 0u;
        __syncthreads();
        cnt[t] += v;
        __syncthreads();
    }
    const uint32_t n_chain = cnt[255];
    uint32_t pc = cnt[t] - c, pi = n_chain + (b0 - pc);       /* chain entries before this segment; independent ones likewise */
    for (uint32_t k = b0; k < b1; k++) {
        if (f->ctu_aux[k] & (OH_AUX_WAITS | OH_AUX_AWAITED)) f->ctu_order[pc++] = k;
        else f->ctu_order[pi++] = k;
    }
}

/* last: the counts are compared with the host's (the bucket ranges were laid out for them) and the summary — error, counts,
 * level statistics — goes to the list's pinned host block with plain stores: visible to the host once the event behind this
 * kernel has completed */
__global__ __launch_bounds__(64) void prep_finish(const OhBatch B)
{
    DevFrame *f = (DevFrame *)B.f[blockIdx.x];
    DevSummary *s = summary_of(f);
    if (threadIdx.x == 0 && !failed(f)) {
        for (int k = 0; k < 4; k++) {
            s->tu_cnt[k] = f->tu_cursor[k];
            if (f->tu_cursor[k] != f->tu_cnt[k]) fail(f, OH_PE_TU, 0xffffffffu);
        }
        s->n_cross = f->tu_cursor[9];
        if (f->tu_cursor[9] != f->n_cross) fail(f, OH_PE_TU, 0xffffffffu);
    }
    __syncthreads();
    const uint32_t words = (uint32_t)((sizeof(DevSummary) + (size_t)f->n_levels * sizeof(DevLevelStat)) / 4);
    const uint32_t *src = (const uint32_t *)s;
    uint32_t *dst = (uint32_t *)f->summary_host;
    for (uint32_t i = threadIdx.x; i < words; i += 64)
        dst[i] = *(volatile const uint32_t *)(src + i);
}

/* =========================================================================================
 * launcher: everything for ONE work list, enqueued on `st` (the engine's copy stream, behind the list's H2D copy)
 * ======================================================================================= */
/* ---- the hand-over of a work list that lies in page-locked host memory (OH_FRAME_PINNED): the GPU PULLS it.  A table of segments
 * (host source, device destination, bytes) stands in the engine's pinned staging block; the workgroups copy them over PCIe in 16-byte
 * units (a kernel reads pinned host memory at the copy engines' rate: 55-57 GB/s on this box, tools/pull_rate.hip) — one launch
 * instead of one DMA request per array, and no staging copy on the host.  MEASURED: the host's share of the hand-over falls from 0.30 to
 * 0.10 ms per 4K picture, but beside the passes of the other batches the pull's workgroups get few wave slots and move ~30 GB/s where
 * the copy engines move 41: the bench decodes 58-62 Gpixels/s this way against 83 with the staged copy (fifteen DMA requests per
 * picture: 47).  The staged copy stays the default; this is the path of OH_FRAME_PINNED lists. ---- */
typedef unsigned int pull_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void prep_pull(const OhPullSeg *__restrict__ segs, const int nseg)
{
    for (int s = blockIdx.y; s < nseg; s += gridDim.y) {
        const OhPullSeg g = segs[s];
        const unsigned stride = gridDim.x * 256u, first = blockIdx.x * 256u + threadIdx.x;
        if ((((uintptr_t)g.src | (uintptr_t)g.dst) & 15) == 0) {
            const pull_u4 *__restrict__ src = (const pull_u4 *)g.src;
            pull_u4 *__restrict__ dst = (pull_u4 *)g.dst;
            const unsigned n = (unsigned)(g.bytes >> 4);
            /* four loads in flight per lane: the link's latency (microseconds) is what has to be covered, by few resident waves when the
             * passes of other batches hold most of the wave slots */
            unsigned i = first;
            for (; i + 3 * stride < n; i += 4 * stride) {
                const pull_u4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
                const pull_u4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
                dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
            }
            for (; i < n; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
            for (unsigned k = (n << 4) + first; k < g.bytes; k += stride) ((uint8_t *)g.dst)[k] = ((const uint8_t *)g.src)[k];
        } else if ((((uintptr_t)g.src | (uintptr_t)g.dst) & 3) == 0) {
            const uint32_t *__restrict__ src = (const uint32_t *)g.src;
            uint32_t *__restrict__ dst = (uint32_t *)g.dst;
            const unsigned n = (unsigned)(g.bytes >> 2);
            for (unsigned i = first; i < n; i += stride) dst[i] = __builtin_nontemporal_load(src + i);
            for (unsigned i = (n << 2) + first; i < g.bytes; i += stride) ((uint8_t *)g.dst)[i] = ((const uint8_t *)g.src)[i];
        } else {
            for (unsigned i = first; i < g.bytes; i += stride) ((uint8_t *)g.dst)[i] = ((const uint8_t *)g.src)[i];
        }
    }
}
extern "C" void ohk_pull(const OhPullSeg *segs, int nseg, size_t total_bytes, hipStream_t st)
{
    if (nseg <= 0)
        return;
    /* 64 workgroups: alone on the chip even 8 x 32 workgroups fill the link (tools/pull_rate.hip: 55-57 GB/s); beside the passes of the
     * batches in flight the pull gets ~30 GB/s whatever the grid (4 / 16 / 64 columns: 61.7 / 58.2 / 53.7 Gpixels/s in the bench) */
    const unsigned gx = 4, gy = nseg < 16 ? (unsigned)nseg : 16;
    (void)total_bytes;
    hipLaunchKernelGGL(prep_pull, dim3(gx, gy), dim3(256), 0, st, segs, nseg);
}

extern "C" void ohk_prepare(const OhBatch *B, int nb, const OhPrepCounts *n, uint32_t max_mc_runs, uint32_t max_cross, hipStream_t st)
{
    /* n: the LARGEST count of the batch per kind (every kernel bounds its index by the list's own count) */
    hipLaunchKernelGGL(prep_clear, dim3(16, nb), dim3(256), 0, st, *B);
    if (n->n_pu) {
        hipLaunchKernelGGL(prep_pu_expand, dim3((n->n_pu + 255) / 256, nb), dim3(256), 0, st, *B);
        if (max_mc_runs) hipLaunchKernelGGL(prep_mc_group, dim3(max_mc_runs, nb), dim3(64), 0, st, *B);
    }
    if (n->n_tu) {
        if (max_cross) hipLaunchKernelGGL(prep_tu_mark, dim3((n->n_tu + 255) / 256, nb), dim3(256), 0, st, *B);
        hipLaunchKernelGGL(prep_tu_scatter, dim3((n->n_tu + 255) / 256, nb), dim3(256), 0, st, *B);
    }
    if (n->n_intra) {
        hipLaunchKernelGGL(prep_intra_sub, dim3((n->n_sub + 255) / 256, nb), dim3(256), 0, st, *B);
        hipLaunchKernelGGL(prep_intra_items, dim3((n->n_intra + 255) / 256, nb), dim3(256), 0, st, *B);
        hipLaunchKernelGGL(prep_intra_ctu, dim3(n->n_ictu, nb), dim3(64), 0, st, *B);
        hipLaunchKernelGGL(prep_intra_levels, dim3(n->n_levels, nb), dim3(64), 0, st, *B);
        hipLaunchKernelGGL(prep_intra_wait, dim3((n->n_ictu + 255) / 256, nb), dim3(256), 0, st, *B);
        hipLaunchKernelGGL(prep_intra_order, dim3(nb), dim3(256), 0, st, *B);
    }
    hipLaunchKernelGGL(prep_finish, dim3(nb), dim3(64), 0, st, *B);
}
