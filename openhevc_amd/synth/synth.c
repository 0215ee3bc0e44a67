/*
 * synth.c — synthetic decoded-syntax generator (see include/ohevc_synth.h).
 *
 * Walks every CTB in raster order and every coding quadtree in z-scan order — the order the
 * reference's hls_coding_quadtree / hls_transform_tree visit blocks (hevc.c:2508, :1443) — and
 * records what the reference's CTU loop would hand to the DSP tables.  Boundary strengths follow
 * the H.265 8.7.2.4 rules the reference implements in hevc_filter.c:584-941 (that derivation
 * stays on the host).
 */
#include <stdlib.h>
#include <string.h>
#include "../../include/ohevc_synth.h"

typedef struct Cell {            /* one 4x4 luma block */
    uint8_t intra, cbf, edges;   /* edges: 1 TU-left 2 TU-top 4 PU-left 8 PU-top */
    uint8_t call;                /* log2 size of the block ff_hevc_deblocking_boundary_strengths is called for here (hevc.c:1578, 1607, 2400, 2484) */
    int8_t  ref[2];              /* picture ids, -1 = list unused */
    int16_t mv[2][2];
} Cell;

typedef struct Gen {
    OhRecorder *rec;
    const OhSynthParams *sp;
    OhPicParams p;
    uint64_t s;
    Cell *cells; int cw, ch;
    int32_t ref_pics[OH_MAX_REFS]; int n_ref;
    int8_t *qp; uint8_t *is_pcm, *is_intra;
    const OhCtbMaps *maps;           /* slices / tiles, NULL = one slice, one tile */
    int16_t blk[32 * 32];
} Gen;

/* splitmix64 */
static uint64_t rnd64(Gen *g)
{
    uint64_t z = (g->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int rnd(Gen *g, int n) { return n <= 1 ? 0 : (int)(rnd64(g) % (uint64_t)n); }
static int rnd_range(Gen *g, int lo, int hi) { return lo + rnd(g, hi - lo + 1); }
static int pct(Gen *g, int p) { return rnd(g, 100) < p; }
static int clipi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

void oh_synth_defaults(OhSynthParams *sp, int slice_type, uint64_t seed)
{
    memset(sp, 0, sizeof(*sp));
    sp->seed = seed; sp->slice_type = slice_type; sp->n_refs = slice_type ? 2 : 0;
    sp->intra_pct = 12; sp->skip_pct = 35; sp->bi_pct = 45; sp->frac_mv_pct = 75; sp->mv_range = 160;
    sp->cbf_pct = 60; sp->weighted_pct = 0; sp->split_pct = 45; sp->qp_base = 30; sp->qp_var = 5;
    sp->sao_pct = 50; sp->tskip_pct = 0; sp->pcm_pct = 0; sp->bypass_pct = 0; sp->vary_deblock_offsets = 0;
}

/* ---- coefficient blocks ---- */
static int laplace(Gen *g, int scale)
{
    /* two-sided geometric ~ Laplacian */
    int m = 0, k = 1 + rnd(g, 6);
    for (int i = 0; i < k; i++)
        m += rnd(g, scale + 1);
    m = m / 2;
    return rnd(g, 2) ? m : -m;
}

static void gen_coeffs(Gen *g, int log2, int kind)
{
    int n = 1 << log2, n2 = n * n;
    memset(g->blk, 0, sizeof(int16_t) * (size_t)n2);
    if (kind == OH_TU_SKIP || kind == OH_TU_BYPASS) {
        for (int i = 0; i < n2; i++)
            g->blk[i] = (int16_t)(pct(g, 60) ? laplace(g, kind == OH_TU_SKIP ? 400 : 24) : 0);
        return;
    }
    if (kind == OH_TU_PCM) {
        int mx = (1 << g->p.bit_depth) - 1;
        for (int i = 0; i < n2; i++)
            g->blk[i] = (int16_t)rnd(g, mx + 1);
        return;
    }
    int shape = rnd(g, 10);
    if (shape < 3) {                               /* DC only (idct_dc in the reference) */
        g->blk[0] = (int16_t)clipi(laplace(g, 6000), -32768, 32767);
        if (!g->blk[0]) g->blk[0] = 64;
        return;
    }
    int nz = 1 + rnd(g, shape < 8 ? (n2 / 8 > 1 ? n2 / 8 : 2) : n2 / 2);
    int lim = shape < 8 ? (n > 8 ? n / 2 : n) : n;  /* low-frequency corner or whole block */
    for (int i = 0; i < nz; i++) {
        int x = rnd(g, lim), y = rnd(g, lim);
        if (rnd(g, 3)) { x = rnd(g, x + 1); y = rnd(g, y + 1); }
        int sc = (x + y == 0) ? 5000 : 1800 / (1 + (x + y) / 2);
        g->blk[y * n + x] = (int16_t)clipi(laplace(g, sc), -32768, 32767);
    }
    if (rnd(g, 40) == 0)                            /* rare saturating block */
        g->blk[rnd(g, n2)] = (int16_t)(rnd(g, 2) ? 32767 : -32768);
}

/* ---- per-cell bookkeeping ---- */
static Cell *cell(Gen *g, int x, int y) { return &g->cells[(y >> 2) * g->cw + (x >> 2)]; }

static void mark_edges(Gen *g, int x, int y, int w, int h, int tu, int pu)
{
    for (int yy = y; yy < y + h; yy += 4) {
        Cell *c = cell(g, x, yy);
        c->edges |= (uint8_t)((tu ? 1 : 0) | (pu ? 4 : 0));
    }
    for (int xx = x; xx < x + w; xx += 4) {
        Cell *c = cell(g, xx, y);
        c->edges |= (uint8_t)((tu ? 2 : 0) | (pu ? 8 : 0));
    }
}

static void fill_cells(Gen *g, int x, int y, int w, int h, int intra, const int8_t ref[2], int16_t mv[2][2])
{
    for (int yy = y; yy < y + h; yy += 4)
        for (int xx = x; xx < x + w; xx += 4) {
            Cell *c = cell(g, xx, yy);
            c->intra = (uint8_t)intra;
            c->ref[0] = ref ? ref[0] : -1; c->ref[1] = ref ? ref[1] : -1;
            if (mv) memcpy(c->mv, mv, sizeof(c->mv)); else memset(c->mv, 0, sizeof(c->mv));
        }
}

static void set_call(Gen *g, int x, int y, int log2)
{
    for (int yy = y; yy < y + (1 << log2) && yy < g->p.height; yy += 4)
        for (int xx = x; xx < x + (1 << log2) && xx < g->p.width; xx += 4)
            cell(g, xx, yy)->call = (uint8_t)log2;
}

static void set_cbf(Gen *g, int x, int y, int n)
{
    for (int yy = y; yy < y + n; yy += 4)
        for (int xx = x; xx < x + n; xx += 4)
            cell(g, xx, yy)->cbf = 1;
}

/* ---- transform trees ---- */
typedef struct CuInfo {
    int intra, bypass, x, y, log2;
    int mode[4], mode_c[4];          /* intra modes per PU (NxN: 4) */
    int nxn;
} CuInfo;

static int chroma_log2(const Gen *g, int log2) { return g->p.chroma_format_idc == 3 ? log2 : log2 - 1; }

static int pick_kind(Gen *g, const CuInfo *cu, int c_idx, int log2)
{
    if (cu->bypass) return OH_TU_BYPASS;
    if (log2 == 2 && pct(g, g->sp->tskip_pct)) return OH_TU_SKIP;
    if (cu->intra && c_idx == 0 && log2 == 2) return OH_TU_DST4;
    return OH_TU_IDCT;
}

static int pick_flags(Gen *g, const CuInfo *cu, int kind)
{
    int f = cu->intra ? 0 : OH_TUF_ADD_NOW;
    if ((kind == OH_TU_SKIP || kind == OH_TU_BYPASS) && rnd(g, 4) == 0) {
        f |= OH_TUF_RDPCM;
        if (rnd(g, 2)) f |= OH_TUF_RDPCM_VER;
    }
    if (kind == OH_TU_SKIP && cu->intra && rnd(g, 8) == 0)
        f |= OH_TUF_ROTATE;
    return f;
}

/* one transform block of one plane: optional residual item, and the intra item for intra CUs */
/* returns the index of the recorded transform block, OH_NO_COEFF when cbf == 0 */
static uint32_t emit_tb(Gen *g, const CuInfo *cu, int c_idx, int xl, int yl, int log2, int mode, int avail, int cbf, int zero_block)
{
    int hs = oh_hshift(&g->p, c_idx), vs = oh_vshift(&g->p, c_idx);
    int x = xl >> hs, y = yl >> vs;
    uint32_t tu = OH_NO_COEFF;
    if (!cbf && zero_block) {                          /* cross-component prediction with cbf 0: a block of zeros carries it, hevc.c:1315-1331 */
        memset(g->blk, 0, sizeof(int16_t) << (2 * log2));
        tu = oh_rec_tu(g->rec, c_idx, x, y, log2, OH_TU_BYPASS, cu->intra ? 0 : OH_TUF_ADD_NOW, g->blk);
    }
    if (cbf) {
        int kind = pick_kind(g, cu, c_idx, log2);
        int flags = pick_flags(g, cu, kind);
        if (kind != OH_TU_BYPASS && kind != OH_TU_PCM && g->sp->sparse_pct > 0 && pct(g, g->sp->sparse_pct)) {   /* no draw when off: the
                                                                   dense streams of the golden fixtures stay what they were */
            /* what residual_coding parses: a few non-zero levels, the block's QP, its scaling matrix (hevc_cabac.c:1478-1494) */
            uint32_t pairs[1024];
            int n = 1 << log2, shape = rnd(g, 10), cnt = 0;
            int nz = shape < 3 ? 1 : 1 + rnd(g, shape < 8 ? (n * n / 8 > 1 ? n * n / 8 : 2) : n * n / 2);
            int lim = shape < 8 ? (n > 8 ? n / 2 : n) : n;
            memset(g->blk, 0, sizeof(int16_t) * (size_t)(n * n));        /* marks the positions already taken */
            for (int i = 0; i < nz; i++) {
                int px = shape < 3 ? 0 : rnd(g, lim), py = shape < 3 ? 0 : rnd(g, lim);
                int lvl = clipi(laplace(g, px + py == 0 ? 60 : 12), -32768, 32767);
                if (!lvl) lvl = 1;
                if (rnd(g, 60) == 0) lvl = rnd(g, 2) ? 32767 : -32768;                  /* rare saturating level */
                if (g->blk[py * n + px]) continue;
                g->blk[py * n + px] = 1;
                pairs[cnt++] = (uint32_t)(py * n + px) | ((uint32_t)(uint16_t)(int16_t)lvl << 16);
            }
            int qp = clipi(g->sp->qp_base + rnd_range(g, -g->sp->qp_var, g->sp->qp_var), 0, 51) + 6 * (g->p.bit_depth - 8);
            int matrix = (g->sp->scaling_list && !(kind == OH_TU_SKIP && log2 > 2)) ? 3 * !cu->intra + c_idx : (int)OH_FLAT_MATRIX;
            tu = oh_rec_tu_sparse(g->rec, c_idx, x, y, log2, kind, flags, qp, matrix, cnt, pairs);
        } else {
            gen_coeffs(g, log2, kind);
            tu = oh_rec_tu(g->rec, c_idx, x, y, log2, kind, flags, g->blk);
        }
    }
    if (cu->intra)
        oh_rec_intra(g->rec, c_idx, x, y, log2, mode, avail, tu);
    return tu;
}

static void gen_tu(Gen *g, const CuInfo *cu, int x, int y, int log2, int blk_idx, int xb, int yb)
{
    int nch = g->p.chroma_format_idc;
    int pu = cu->nxn ? blk_idx : 0;
    int avail = cu->intra ? oh_rec_avail(g->rec, x, y, 1 << log2, 1 << log2) : 0;
    int cbf_y = pct(g, g->sp->cbf_pct) || cu->bypass;
    mark_edges(g, x, y, 1 << log2, 1 << log2, 1, 0);
    set_call(g, x, y, log2);                                /* one call per transform unit, hevc.c:1578 */
    uint32_t tu_y = emit_tb(g, cu, 0, x, y, log2, cu->mode[pu], avail, cbf_y, 0);
    /* cross_pf is only parsed when the luma block is coded (hevc.c:1286-1290); 4:4:4 only */
    int cross = nch == 3 && cbf_y && !cu->bypass && g->sp->ccp_pct > 0 && pct(g, g->sp->ccp_pct);
    if (cbf_y) set_cbf(g, x, y, 1 << log2);
    /* the luma block is reconstructed before its chroma blocks: from here on it counts as decoded
     * (the second chroma block of a 4:2:2 pair has the first one above it, hevc.c:1297-1345) */
    if (cu->intra) oh_rec_mark_decoded(g->rec, x, y, 1 << log2, 1 << log2);
    if (nch == 3 || (nch && log2 > 2)) {
        int lc = chroma_log2(g, log2), nblk = nch == 2 ? 2 : 1;
        for (int c = 1; c < 3; c++)
            for (int i = 0; i < nblk; i++) {
                int yy = y + (i << lc);               /* 4:2:2: two square blocks stacked vertically */
                int av = nch == 2 ? (cu->intra ? oh_rec_avail(g->rec, x, yy, 2 << lc, 1 << lc) : 0) : avail;
                int scale = cross ? (rnd(g, 2) ? 1 : -1) * (1 << rnd(g, 4)) : 0;          /* log2_res_scale_abs_plus1 - 1 in 0..3, sign */
                uint32_t tu_c = emit_tb(g, cu, c, x, yy, lc, cu->mode_c[pu], av, pct(g, g->sp->cbf_pct / 2) || cu->bypass, cross && scale);
                if (cross && scale && tu_c != OH_NO_COEFF) oh_rec_tu_cross(g->rec, tu_c, tu_y, scale);
            }
    } else if (nch && blk_idx == 3) {               /* chroma of four 4x4 luma blocks, hevc.c:1395-1420 */
        int nblk = nch == 2 ? 2 : 1;
        for (int c = 1; c < 3; c++)
            for (int i = 0; i < nblk; i++) {
                int yy = yb + 4 * i;
                int av = cu->intra ? oh_rec_avail(g->rec, xb, yy, 8, nch == 2 ? 4 : 8) : 0;
                emit_tb(g, cu, c, xb, yy, 2, cu->mode_c[0], av, pct(g, g->sp->cbf_pct / 2) || cu->bypass, 0);
            }
    }
}

static void gen_tt(Gen *g, const CuInfo *cu, int x, int y, int log2, int depth, int blk_idx, int xb, int yb)
{
    int split;
    if (log2 > 5) split = 1;
    else if (log2 <= g->p.log2_min_tb_size) split = 0;
    else if (cu->nxn && depth == 0) split = 1;      /* interSplit / NxN forces one split */
    else split = depth < 3 && pct(g, g->sp->split_pct);
    if (split) {
        int h = 1 << (log2 - 1);
        gen_tt(g, cu, x, y, log2 - 1, depth + 1, 0, x, y);
        gen_tt(g, cu, x + h, y, log2 - 1, depth + 1, 1, x, y);
        gen_tt(g, cu, x, y + h, log2 - 1, depth + 1, 2, x, y);
        gen_tt(g, cu, x + h, y + h, log2 - 1, depth + 1, 3, x, y);
    } else {
        gen_tu(g, cu, x, y, log2, blk_idx, xb, yb);
    }
}

/* ---- prediction units ---- */
static void gen_mv(Gen *g, int16_t mv[2])
{
    int r = g->sp->mv_range;
    for (int k = 0; k < 2; k++) {
        int v = rnd_range(g, -r, r);
        if (rnd(g, 60) == 0)                          /* far outside the picture: exercises clamping */
            v = (rnd(g, 2) ? 1 : -1) * (4 * (k ? g->p.height : g->p.width) + rnd(g, 400));
        if (!pct(g, g->sp->frac_mv_pct))
            v &= ~3;
        mv[k] = (int16_t)clipi(v, -32768, 32767);
    }
}

static void gen_pu(Gen *g, int x, int y, int w, int h)
{
    int8_t ref[2] = { -1, -1 };
    int slot[2] = { -1, -1 };
    int16_t mv[2][2] = { { 0, 0 }, { 0, 0 } };
    int can_bi = g->sp->slice_type == 2 && w + h != 12;   /* 8x4 / 4x8 are uni-predicted only */
    int lists = can_bi && pct(g, g->sp->bi_pct) ? 3 : (g->sp->slice_type == 2 && rnd(g, 2) ? 2 : 1);
    for (int l = 0; l < 2; l++) {
        if (!(lists & (1 << l))) continue;
        slot[l] = rnd(g, g->n_ref);
        ref[l] = (int8_t)g->ref_pics[slot[l]];
        gen_mv(g, mv[l]);
    }
    OhWeights wp, *pwp = NULL;
    if (pct(g, g->sp->weighted_pct)) {
        memset(&wp, 0, sizeof(wp));
        wp.log2_denom[0] = (uint8_t)rnd(g, 8);
        wp.log2_denom[1] = (uint8_t)clipi(wp.log2_denom[0] + rnd_range(g, -1, 1), 0, 7);
        for (int l = 0; l < 2; l++)
            for (int c = 0; c < 3; c++) {
                wp.w[l][c] = (int16_t)((1 << wp.log2_denom[c ? 1 : 0]) + rnd_range(g, -24, 24));
                wp.o[l][c] = (int16_t)rnd_range(g, -40, 40);
            }
        pwp = &wp;
    }
    oh_rec_pu(g->rec, x, y, w, h, slot[0], mv[0][0], mv[0][1], slot[1], mv[1][0], mv[1][1], pwp);
    fill_cells(g, x, y, w, h, 0, ref, mv);
    mark_edges(g, x, y, w, h, 0, 1);
}

/* ---- coding units ---- */
static void gen_cu(Gen *g, int x, int y, int log2)
{
    const OhSynthParams *sp = g->sp;
    int n = 1 << log2;
    CuInfo cu;
    memset(&cu, 0, sizeof(cu));
    cu.x = x; cu.y = y; cu.log2 = log2;
    int qp = clipi(sp->qp_base + rnd_range(g, -sp->qp_var, sp->qp_var), 0, 51);
    int mcw = oh_min_cb_width(&g->p), l = g->p.log2_min_cb_size;
    for (int yy = y >> l; yy < (y + n) >> l; yy++)
        for (int xx = x >> l; xx < (x + n) >> l; xx++)
            g->qp[yy * mcw + xx] = (int8_t)qp;
    mark_edges(g, x, y, n, n, 1, 1);

    cu.intra = sp->slice_type == 0 || pct(g, sp->intra_pct);
    cu.bypass = g->p.transquant_bypass_enable && pct(g, sp->bypass_pct);
    int pcm = cu.intra && log2 >= 3 && log2 <= 5 && pct(g, sp->pcm_pct);
    if (cu.intra) {                                     /* tab_mvf[].pred_flag = PF_INTRA for every PU of the CU, hevc.c:2380-2395 */
        int lp = g->p.log2_min_pu_size, mpw = oh_min_pu_width(&g->p);
        for (int yy = y >> lp; yy < (y + n) >> lp; yy++)
            for (int xx = x >> lp; xx < (x + n) >> lp; xx++)
                g->is_intra[yy * mpw + xx] = 1;
    }
    if ((cu.bypass) || (pcm && g->p.pcm_loop_filter_disable)) {   /* set_deblocking_bypass, hevc.c:1428-1441 */
        int lp = g->p.log2_min_pu_size, mpw = oh_min_pu_width(&g->p);
        for (int yy = y >> lp; yy < (y + n) >> lp; yy++)
            for (int xx = x >> lp; xx < (x + n) >> lp; xx++)
                g->is_pcm[yy * mpw + xx] = 2;
    }
    if (pcm) {                                          /* hls_pcm_sample, hevc.c:1587-1640 */
        fill_cells(g, x, y, n, n, 1, NULL, NULL);
        set_call(g, x, y, log2);                        /* hevc.c:1607 */
        gen_coeffs(g, log2, OH_TU_PCM);
        oh_rec_tu(g->rec, 0, x, y, log2, OH_TU_PCM, OH_TUF_ADD_NOW, g->blk);
        if (g->p.chroma_format_idc) {
            int lc = chroma_log2(g, log2);
            for (int c = 1; c < 3; c++)
                for (int i = 0; i < (g->p.chroma_format_idc == 2 ? 2 : 1); i++) {      /* 4:2:2: w/2 x h samples as two squares */
                    gen_coeffs(g, lc, OH_TU_PCM);
                    oh_rec_tu(g->rec, c, x >> oh_hshift(&g->p, c), (y >> oh_vshift(&g->p, c)) + (i << lc), lc, OH_TU_PCM,
                              OH_TUF_ADD_NOW, g->blk);
                }
        }
        oh_rec_mark_decoded(g->rec, x, y, n, n);
        return;
    }
    if (cu.intra) {
        fill_cells(g, x, y, n, n, 1, NULL, NULL);
        cu.nxn = log2 == g->p.log2_min_cb_size && log2 == 3 && pct(g, 40);
        for (int k = 0; k < 4; k++) {
            cu.mode[k] = rnd(g, 4) == 0 ? rnd(g, 2) : rnd(g, 35);        /* planar/DC more frequent */
            int pick = rnd(g, 5);                                         /* intra_chroma_pred_mode, hevc.c:2280-2300 */
            static const int tab[4] = { 0, 26, 10, 1 };
            cu.mode_c[k] = pick == 4 ? cu.mode[k] : (tab[pick] == cu.mode[k] ? 34 : tab[pick]);
        }
        if (!cu.nxn)
            for (int k = 1; k < 4; k++) { cu.mode[k] = cu.mode[0]; cu.mode_c[k] = cu.mode_c[0]; }
        if (g->p.chroma_format_idc != 3)
            for (int k = 1; k < 4; k++) cu.mode_c[k] = cu.mode_c[0];
        gen_tt(g, &cu, x, y, log2, 0, 0, x, y);
        return;
    }
    /* inter: partition (hevc.c:2437-2476) */
    int part = rnd(g, 10);
    int h2 = n / 2, q = n / 4;
    if (part < 5)                    gen_pu(g, x, y, n, n);
    else if (part == 5)            { gen_pu(g, x, y, n, h2); gen_pu(g, x, y + h2, n, h2); }
    else if (part == 6)            { gen_pu(g, x, y, h2, n); gen_pu(g, x + h2, y, h2, n); }
    else if (part == 7 && log2 > 3) { gen_pu(g, x, y, n, q); gen_pu(g, x, y + q, n, n - q); }             /* 2NxnU */
    else if (part == 8 && log2 > 3) { gen_pu(g, x, y, n - q, n); gen_pu(g, x + n - q, y, q, n); }         /* nRx2N */
    else if (part == 9 && log2 > 3) { gen_pu(g, x, y, n, n - q); gen_pu(g, x, y + n - q, n, q); }         /* 2NxnD */
    else if (log2 == g->p.log2_min_cb_size && log2 > 3) {                                                  /* NxN   */
        gen_pu(g, x, y, h2, h2); gen_pu(g, x + h2, y, h2, h2); gen_pu(g, x, y + h2, h2, h2); gen_pu(g, x + h2, y + h2, h2, h2);
    } else                           gen_pu(g, x, y, n, n);
    oh_rec_mark_decoded(g->rec, x, y, n, n);
    if (!pct(g, sp->skip_pct) || cu.bypass)
        gen_tt(g, &cu, x, y, log2, 0, 0, x, y);
    else
        set_call(g, x, y, log2);                        /* skipped / no residual: one call for the coding block, hevc.c:2400, 2484 */
}

static void gen_cqt(Gen *g, int x, int y, int log2)
{
    int n = 1 << log2;
    int inside = x + n <= g->p.width && y + n <= g->p.height;
    int split = !inside || (log2 > g->p.log2_min_cb_size && pct(g, g->sp->split_pct));
    if (log2 <= g->p.log2_min_cb_size) split = 0;
    if (split) {
        int h = n / 2;
        for (int k = 0; k < 4; k++) {
            int xx = x + (k & 1) * h, yy = y + (k >> 1) * h;
            if (xx < g->p.width && yy < g->p.height)
                gen_cqt(g, xx, yy, log2 - 1);
        }
    } else {
        gen_cu(g, x, y, log2);
    }
}

/* ---- boundary strength, H.265 8.7.2.4 (hevc_filter.c:584-700) ---- */
static int motion_bs(const Cell *p, const Cell *q)
{
    int np = (p->ref[0] >= 0) + (p->ref[1] >= 0), nq = (q->ref[0] >= 0) + (q->ref[1] >= 0);
#define FAR(a, b) (abs((a)[0] - (b)[0]) >= 4 || abs((a)[1] - (b)[1]) >= 4)
    if (np != nq)
        return 1;
    if (np == 1) {
        int lp = p->ref[0] >= 0 ? 0 : 1, lq = q->ref[0] >= 0 ? 0 : 1;
        if (p->ref[lp] != q->ref[lq])
            return 1;
        return FAR(p->mv[lp], q->mv[lq]);
    }
    if (!((p->ref[0] == q->ref[0] && p->ref[1] == q->ref[1]) || (p->ref[0] == q->ref[1] && p->ref[1] == q->ref[0])))
        return 1;
    if (p->ref[0] != p->ref[1]) {
        if (p->ref[0] == q->ref[0])
            return FAR(p->mv[0], q->mv[0]) || FAR(p->mv[1], q->mv[1]);
        return FAR(p->mv[0], q->mv[1]) || FAR(p->mv[1], q->mv[0]);
    }
    return (FAR(p->mv[0], q->mv[0]) || FAR(p->mv[1], q->mv[1])) &&
           (FAR(p->mv[0], q->mv[1]) || FAR(p->mv[1], q->mv[0]));
#undef FAR
}

/* slices / tiles: the block at (x, y) belongs to a slice without deblocking (no strengths are derived for its edges, hevc.c:1577),
 * or the edge lies on a CTB boundary that is a slice / tile boundary the loop filter must not cross (hevc_filter.c:819-824, 857-862) */
static int edge_off(const Gen *g, int x, int y, int vertical)
{
    if (!g->maps)
        return 0;
    const int lc = g->p.log2_ctb_size, ctbw = oh_ctb_width(&g->p), rs = (y >> lc) * ctbw + (x >> lc);
    if (g->maps->deblock_disabled[rs])
        return 1;
    if ((vertical ? x : y) & ((1 << lc) - 1))
        return 0;
    const int fl = oh_ctb_bs_flags(g->maps, ctbw, rs);
    const int slice_b = vertical ? fl & OH_BSF_LEFT_SLICE : fl & OH_BSF_UP_SLICE, tile_b = vertical ? fl & OH_BSF_LEFT_TILE : fl & OH_BSF_UP_TILE;
    const int bd_slice = (fl & OH_BSF_ACROSS_SLICES) || !slice_b, bd_tiles = g->maps->loop_filter_across_tiles || !tile_b;
    return !(bd_slice && bd_tiles);
}

static void derive_bs(Gen *g)
{
    uint8_t *vbs = oh_rec_vertical_bs(g->rec), *hbs = oh_rec_horizontal_bs(g->rec);
    int bsw = g->p.width >> 2;
    for (int y = 0; y < g->p.height; y += 4)
        for (int x = 0; x < g->p.width; x += 4) {
            const Cell *q = cell(g, x, y);
            if (edge_off(g, x, y, 1) && edge_off(g, x, y, 0))
                continue;
            if (x && !(x & 7) && (q->edges & 5) && !edge_off(g, x, y, 1)) {
                const Cell *p = cell(g, x - 4, y);
                int bs = (p->intra || q->intra) ? 2 : (((q->edges & 1) && (p->cbf || q->cbf)) ? 1 : motion_bs(p, q));
                vbs[(x + y * bsw) >> 2] = (uint8_t)bs;
            }
            if (y && !(y & 7) && (q->edges & 10) && !edge_off(g, x, y, 0)) {
                const Cell *p = cell(g, x, y - 4);
                int bs = (p->intra || q->intra) ? 2 : (((q->edges & 2) && (p->cbf || q->cbf)) ? 1 : motion_bs(p, q));
                hbs[(x + y * bsw) >> 2] = (uint8_t)bs;
            }
        }
}

/* the same picture as maps for the engine's boundary-strength pass (OhBsInputs): the reference's tab_mvf / cbf_luma as this
 * generator's cells hold them (POC = picture id of the reference), the call sizes, single slice and tile */
static void emit_bs_maps(Gen *g)
{
    OhBsInputs *in = oh_rec_bs_maps(g->rec);
    if (!in)
        return;
    const int lpu = g->p.log2_min_pu_size, ltu = g->p.log2_min_tb_size, mpw = oh_min_pu_width(&g->p), mtw = g->p.width >> ltu;
    OhMvField *mvf = (OhMvField *)in->mvf;
    uint8_t *cbf = (uint8_t *)in->cbf_luma, *call = (uint8_t *)in->call_log2;
    for (int y = 0; y < g->p.height; y += 4)
        for (int x = 0; x < g->p.width; x += 4) {
            const Cell *c = cell(g, x, y);
            if (!(x & ((1 << lpu) - 1)) && !(y & ((1 << lpu) - 1))) {
                OhMvField *m = &mvf[(y >> lpu) * mpw + (x >> lpu)];
                memset(m, 0, sizeof(*m));
                for (int l = 0; l < 2; l++)
                    if (!c->intra && c->ref[l] >= 0) {
                        m->pred_flag |= 1u << l;
                        m->mv[l][0] = c->mv[l][0]; m->mv[l][1] = c->mv[l][1];
                        m->poc[l] = c->ref[l]; m->ref_idx[l] = (uint8_t)c->ref[l];
                    }
            }
            if (!(x & ((1 << ltu) - 1)) && !(y & ((1 << ltu) - 1))) {
                cbf[(y >> ltu) * mtw + (x >> ltu)] = c->cbf;
                call[(y >> ltu) * mtw + (x >> ltu)] = g->maps && g->maps->deblock_disabled[(y >> g->p.log2_ctb_size) * oh_ctb_width(&g->p) +
                                                      (x >> g->p.log2_ctb_size)] ? 0 : c->call;
            }
        }
    in->loop_filter_across_tiles = g->maps ? g->maps->loop_filter_across_tiles : 1;      /* ctb_flags: oh_rec_finish() from the CTB maps */
}

static void gen_sao(Gen *g)
{
    OhSaoCtb *sao = oh_rec_sao(g->rec);
    int n = oh_ctb_width(&g->p) * oh_ctb_height(&g->p);
    int sc = g->p.bit_depth - (g->p.bit_depth < 10 ? g->p.bit_depth : 10);
    for (int i = 0; i < n; i++) {
        OhSaoCtb *s = &sao[i];
        memset(s, 0, sizeof(*s));
        if (!g->p.sao_enabled)
            continue;
        for (int c = 0; c < (g->p.chroma_format_idc ? 3 : 1); c++) {
            if (c == 2) {                            /* Cr shares type and class with Cb, hevc.c:1137-1150 */
                s->type_idx[2] = s->type_idx[1];
                s->eo_class[2] = s->eo_class[1];
            } else {
                s->type_idx[c] = pct(g, g->sp->sao_pct) ? (uint8_t)(1 + rnd(g, 2)) : 0;
                s->eo_class[c] = (uint8_t)rnd(g, 4);
            }
            if (!s->type_idx[c])
                continue;
            s->band_position[c] = (uint8_t)rnd(g, 32);
            for (int k = 1; k < 5; k++) {
                int mag = rnd(g, 8) << sc;
                if (s->type_idx[c] == 1) s->offset_val[c][k] = (int16_t)(rnd(g, 2) ? mag : -mag);
                else                     s->offset_val[c][k] = (int16_t)(k <= 2 ? mag : -mag);
            }
        }
    }
}

const OhFrame *oh_synth_picture(OhRecorder *rec, const OhSynthParams *sp, int cur_pic,
                                const int32_t *ref_pics, int n_ref_pics)
{
    Gen g;
    memset(&g, 0, sizeof(g));
    g.rec = rec; g.sp = sp; g.p = *oh_rec_params(rec); g.s = sp->seed * 0x2545F4914F6CDD1Dull + 0x1234567;
    g.cw = (g.p.width + 3) >> 2; g.ch = (g.p.height + 3) >> 2;
    g.cells = (Cell *)calloc((size_t)g.cw * g.ch, sizeof(Cell));
    g.n_ref = sp->n_refs < n_ref_pics ? sp->n_refs : n_ref_pics;
    if (sp->slice_type && g.n_ref < 1) { free(g.cells); return NULL; }
    for (int i = 0; i < g.n_ref; i++) g.ref_pics[i] = ref_pics[i];

    oh_rec_begin(rec, cur_pic, ref_pics, n_ref_pics);
    g.qp = oh_rec_qp_y_tab(rec);
    g.is_pcm = oh_rec_is_pcm(rec);
    g.is_intra = oh_rec_is_intra(rec);
    if (sp->scaling_list) {                              /* any legal list: entries 1..255 */
        OhScalingList *sl = oh_rec_scaling_list(rec);
        for (size_t i = 0; i < sizeof(sl->sl); i++) ((uint8_t *)sl->sl)[i] = (uint8_t)(1 + rnd(&g, 64 + (int)(i & 63)));
        for (size_t i = 0; i < sizeof(sl->sl_dc); i++) ((uint8_t *)sl->sl_dc)[i] = (uint8_t)(1 + rnd(&g, 200));
    }
    memset(g.qp, sp->qp_base, oh_qp_tab_size(&g.p));

    int ctb = 1 << g.p.log2_ctb_size;
    OhDeblockCtb *db = oh_rec_deblock(rec);
    int beta = 2 * rnd_range(&g, -3, 3), tc = 2 * rnd_range(&g, -3, 3), i = 0;
    const int tiles = sp->tile_cols > 1 || sp->tile_rows > 1;
    int8_t *slice_beta = NULL, *slice_tc = NULL;
    if (sp->n_slices > 1 || tiles) {
        /* slices: contiguous CTB ranges in raster scan starting at random addresses (no tiles), or one per tile / one for all
         * tiles; per slice: loop filtering across its boundaries, deblocking on / off, its own beta / tc offsets */
        OhCtbMaps *m = oh_rec_ctb_maps(rec);
        const int W = oh_ctb_width(&g.p), H = oh_ctb_height(&g.p), n = W * H;
        if (!m) { free(g.cells); return NULL; }
        slice_beta = (int8_t *)calloc((size_t)n, 2); slice_tc = slice_beta ? slice_beta + n : NULL;
        if (!slice_beta) { free(g.cells); return NULL; }
        if (tiles) {
            const int tc_ = sp->tile_cols > 1 ? (sp->tile_cols < W ? sp->tile_cols : W) : 1, tr_ = sp->tile_rows > 1 ? (sp->tile_rows < H ? sp->tile_rows : H) : 1;
            m->tiles_enabled = 1;
            m->loop_filter_across_tiles = !(sp->slice_knobs & OH_SYNTH_NO_LF_ACROSS_TILES);
            for (int cy = 0; cy < H; cy++)
                for (int cx = 0; cx < W; cx++) {
                    int tx = 0, ty = 0;                            /* uniform spacing, hevc_ps.c: column_width = ((i+1)*W)/cols - (i*W)/cols */
                    while (((tx + 1) * W) / tc_ <= cx) tx++;
                    while (((ty + 1) * H) / tr_ <= cy) ty++;
                    m->tile_id[cy * W + cx] = ty * tc_ + tx;
                    m->slice_addr[cy * W + cx] = (sp->slice_knobs & OH_SYNTH_SLICE_PER_TILE) ? ((ty * H) / tr_) * W + (tx * W) / tc_ : 0;
                }
        } else {
            int addr = 0;
            for (int k = 0; k < n; k++) {
                if (k && rnd(&g, n) < sp->n_slices - 1) addr = k;      /* on average n_slices - 1 further slice starts */
                m->slice_addr[k] = addr;
            }
        }
        for (int k = 0; k < n; k++) {
            const int a = m->slice_addr[k];
            if (a == k || (tiles && !(sp->slice_knobs & OH_SYNTH_SLICE_PER_TILE) && k == 0)) {     /* first CTB of a slice: draw its header */
                m->filter_slice_edges[k] = !((sp->slice_knobs & OH_SYNTH_NO_LF_ACROSS_SLICES) && rnd(&g, 2));
                m->deblock_disabled[k] = (sp->slice_knobs & OH_SYNTH_DEBLOCK_OFF_SLICES) && rnd(&g, 3) == 0;
                if (sp->slice_knobs & OH_SYNTH_SLICE_OFFSETS) { beta = 2 * rnd_range(&g, -6, 6); tc = 2 * rnd_range(&g, -6, 6); }
                slice_beta[k] = (int8_t)beta; slice_tc[k] = (int8_t)tc;
            } else {
                m->filter_slice_edges[k] = m->filter_slice_edges[a]; m->deblock_disabled[k] = m->deblock_disabled[a];
                slice_beta[k] = slice_beta[a]; slice_tc[k] = slice_tc[a];
            }
        }
        g.maps = m;
    }
    for (int y = 0; y < g.p.height; y += ctb)
        for (int x = 0; x < g.p.width; x += ctb, i++) {
            if (slice_beta) { beta = slice_beta[i]; tc = slice_tc[i]; }
            if (sp->vary_deblock_offsets && rnd(&g, 4) == 0) {
                beta = 2 * rnd_range(&g, -6, 6); tc = 2 * rnd_range(&g, -6, 6);
            }
            db[i].beta_offset = (int8_t)beta; db[i].tc_offset = (int8_t)tc;
            gen_cqt(&g, x, y, g.p.log2_ctb_size);
        }
    if (sp->bs_from_motion && g.p.deblock_enabled)
        emit_bs_maps(&g);
    else
        derive_bs(&g);
    gen_sao(&g);
    free(g.cells);
    free(slice_beta);
    return oh_rec_finish(rec);
}
