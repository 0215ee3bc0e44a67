/*
 * stream.c — synthetic HEVC Annex-B stream writer (see include/ohevc_stream.h).
 *
 * Written from the syntax tables of H.265 clause 7.3 and the CABAC clauses 9.3.2-9.3.4; the reference decoder parses what
 * this file writes (hevc_ps.c, hevc.c:514-1110 slice header, hevc.c:1112-2700 slice data, hevc_cabac.c), and decoding
 * these streams with it is the test of this file (tests/test_streams.py).
 */
#include <stdlib.h>
#include <string.h>
#include "../../include/ohevc_stream.h"

/* ================================================================================================= bits */
typedef struct Bits { uint8_t *buf; size_t cap, n; } Bits;            /* n = bits written */

static void bits_reserve(Bits *b, size_t more_bits)
{
    size_t need = (b->n + more_bits + 7) / 8 + 16;
    if (need > b->cap) {
        size_t c = b->cap ? b->cap : 4096;
        while (c < need) c *= 2;
        b->buf = (uint8_t *)realloc(b->buf, c);
        memset(b->buf + b->cap, 0, c - b->cap);
        b->cap = c;
    }
}
static void put_bit(Bits *b, int v)
{
    bits_reserve(b, 1);
    if (v) b->buf[b->n >> 3] |= (uint8_t)(0x80 >> (b->n & 7));
    b->n++;
}
static void bits_drain(Bits *to, Bits *from)           /* appends the (byte-aligned) content of `from` to `to` and empties it */
{
    const size_t nb = from->n / 8;
    for (size_t k = 0; k < nb; k++) { bits_reserve(to, 8); to->buf[to->n >> 3] = from->buf[k]; to->n += 8; }
    if (from->buf) memset(from->buf, 0, nb + 1 <= from->cap ? nb + 1 : from->cap);
    from->n = 0;
}
static void put_bits(Bits *b, uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) put_bit(b, (v >> i) & 1); }
static void put_ue(Bits *b, uint32_t v)
{
    uint32_t x = v + 1;
    int len = 0;
    while ((x >> len) > 1) len++;
    put_bits(b, 0, len);
    put_bits(b, x, len + 1);
}
static void put_se(Bits *b, int v) { put_ue(b, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
static void byte_align_zero(Bits *b) { while (b->n & 7) put_bit(b, 0); }
static void rbsp_trailing(Bits *b) { put_bit(b, 1); byte_align_zero(b); }

/* one NAL unit: start code, two header bytes, payload with emulation prevention (7.4.2) */
static int g_layer;                                     /* nuh_layer_id of the NAL units being written (two-layer streams: 0 base, 1 enhancement) */
static void emit_nal(Bits *out, int type, const uint8_t *rbsp, size_t n)
{
    byte_align_zero(out);
    put_bits(out, 0, 24); put_bits(out, 1, 8);
    put_bits(out, (uint32_t)(type << 1) | (uint32_t)(g_layer >> 5), 8);   /* forbidden_zero_bit, nal_unit_type, nuh_layer_id high bit */
    put_bits(out, (uint32_t)((g_layer & 31) << 3) | 1u, 8);               /* nuh_layer_id low bits, nuh_temporal_id_plus1 = 1 */
    int zeros = 0;
    for (size_t i = 0; i < n; i++) {
        if (zeros >= 2 && rbsp[i] <= 3) { put_bits(out, 3, 8); zeros = 0; }
        put_bits(out, rbsp[i], 8);
        zeros = rbsp[i] == 0 ? zeros + 1 : 0;
    }
}

/* ================================================================================================= CABAC encoder (9.3.4) */
static const uint8_t k_range_lps[64][4] = {
    { 128, 176, 208, 240 }, { 128, 167, 197, 227 }, { 128, 158, 187, 216 }, { 123, 150, 178, 205 }, { 116, 142, 169, 195 }, { 111, 135, 160, 185 },
    { 105, 128, 152, 175 }, { 100, 122, 144, 166 }, { 95, 116, 137, 158 }, { 90, 110, 130, 150 }, { 85, 104, 123, 142 }, { 81, 99, 117, 135 },
    { 77, 94, 111, 128 }, { 73, 89, 105, 122 }, { 69, 85, 100, 116 }, { 66, 80, 95, 110 }, { 62, 76, 90, 104 }, { 59, 72, 86, 99 },
    { 56, 69, 81, 94 }, { 53, 65, 77, 89 }, { 51, 62, 73, 85 }, { 48, 59, 69, 80 }, { 46, 56, 66, 76 }, { 43, 53, 63, 72 },
    { 41, 50, 59, 69 }, { 39, 48, 56, 65 }, { 37, 45, 54, 62 }, { 35, 43, 51, 59 }, { 33, 41, 48, 56 }, { 32, 39, 46, 53 },
    { 30, 37, 43, 50 }, { 29, 35, 41, 48 }, { 27, 33, 39, 45 }, { 26, 31, 37, 43 }, { 24, 30, 35, 41 }, { 23, 28, 33, 39 },
    { 22, 27, 32, 37 }, { 21, 26, 30, 35 }, { 20, 24, 29, 33 }, { 19, 23, 27, 31 }, { 18, 22, 26, 30 }, { 17, 21, 25, 28 },
    { 16, 20, 23, 27 }, { 15, 19, 22, 25 }, { 14, 18, 21, 24 }, { 14, 17, 20, 23 }, { 13, 16, 19, 22 }, { 12, 15, 18, 21 },
    { 12, 14, 17, 20 }, { 11, 14, 16, 19 }, { 11, 13, 15, 18 }, { 10, 12, 15, 17 }, { 10, 12, 14, 16 }, { 9, 11, 13, 15 },
    { 9, 11, 12, 14 }, { 8, 10, 12, 14 }, { 8, 9, 11, 13 }, { 7, 9, 11, 12 }, { 7, 9, 10, 12 }, { 7, 8, 10, 11 },
    { 6, 8, 9, 11 }, { 6, 7, 9, 10 }, { 6, 7, 8, 9 }, { 2, 2, 2, 2 } };
static const uint8_t k_next_lps[64] = { 0, 0, 1, 2, 2, 4, 4, 5, 6, 7, 8, 9, 9, 11, 11, 12, 13, 13, 15, 15, 16, 16, 18, 18, 19, 19, 21, 21, 22, 22, 23, 24,
                                        24, 25, 26, 26, 27, 27, 28, 29, 29, 30, 30, 30, 31, 32, 32, 33, 33, 33, 34, 34, 35, 35, 35, 36, 36, 36, 37, 37, 37, 38, 38, 63 };

/* contexts, grouped per syntax element (tables 9-5 .. 9-37); three initialisation types */
enum {
    C_SAO_MERGE = 0, C_SAO_TYPE = 1, C_SPLIT_CU = 2, C_BYPASS_FLAG = 5, C_SKIP = 6, C_QP_DELTA = 9, C_PRED_MODE = 11, C_PART_MODE = 12,
    C_PREV_INTRA = 16, C_CHROMA_MODE = 17, C_MERGE_FLAG = 18, C_MERGE_IDX = 19, C_INTER_DIR = 20, C_REF_IDX = 25, C_MVD_GT0 = 27, C_MVD_GT1 = 28,
    C_MVP = 29, C_ROOT_CBF = 30, C_SPLIT_TU = 31, C_CBF_LUMA = 34, C_CBF_CHROMA = 36, C_TSKIP = 40, C_LAST_X = 42, C_LAST_Y = 60, C_CSBF = 78,
    C_SIG = 82, C_GT1 = 124, C_GT2 = 148, C_RES_SCALE = 154, C_RES_SIGN = 162, C_SIG_TS = 164, C_RDPCM = 166, C_RDPCM_DIR = 168, N_CTX = 170
};
#define X 154                                        /* unused in this initialisation type */
static const uint8_t k_init[3][N_CTX] = {
    {   /* I slices */
        153, 200, 139, 141, 157, 154, X, X, X, 154, 154, X, 184, X, X, X, 184, 63, X, X, X, X, X, X, X, X, X, X, X, X, X,
        153, 138, 138, 111, 141, 94, 138, 182, 154, 139, 139,
        110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
        110, 110, 124, 125, 140, 153, 125, 127, 140, 109, 111, 143, 127, 111, 79, 108, 123, 63,
        91, 171, 134, 141,
        111, 111, 125, 110, 110, 94, 124, 108, 124, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125, 107, 125, 141, 179, 153, 125,
        140, 139, 182, 182, 152, 136, 152, 136, 153, 136, 139, 111, 136, 139, 111,
        140, 92, 137, 138, 140, 152, 138, 139, 153, 74, 149, 92, 139, 107, 122, 152, 140, 179, 166, 182, 140, 227, 122, 197,
        138, 153, 136, 167, 152, 152,
        154, 154, 154, 154, 154, 154, 154, 154, 154, 154,              /* log2_res_scale_abs_plus1 (8), res_scale_sign_flag (2): range extension */
        141, 111, 139, 139, 139, 139 },                                /* sig_coeff_flag of skip / bypass blocks (luma, chroma), explicit_rdpcm_flag (2), _dir_flag (2) */
    {   /* initType 1 */
        153, 185, 107, 139, 126, 154, 197, 185, 201, 154, 154, 149, 154, 139, 154, 154, 154, 152, 110, 122, 95, 79, 63, 31, 31, 153, 153, 140, 198, 168, 79,
        124, 138, 94, 153, 111, 149, 107, 167, 154, 139, 139,
        125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
        125, 110, 94, 110, 95, 79, 125, 111, 110, 78, 110, 111, 111, 95, 94, 108, 123, 108,
        121, 140, 61, 154,
        155, 154, 139, 153, 139, 123, 123, 63, 153, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
        170, 153, 123, 123, 107, 121, 107, 121, 167, 151, 183, 140, 151, 183, 140,
        154, 196, 196, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 137, 169, 194, 166, 167, 154, 167, 137, 182,
        107, 167, 91, 122, 107, 167,
        154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
        140, 140, 139, 139, 139, 139 },
    {   /* initType 2 */
        153, 160, 107, 139, 126, 154, 197, 185, 201, 154, 154, 134, 154, 139, 154, 154, 183, 152, 154, 137, 95, 79, 63, 31, 31, 153, 153, 169, 198, 168, 79,
        224, 167, 122, 153, 111, 149, 92, 167, 154, 139, 139,
        125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
        125, 110, 124, 110, 95, 94, 125, 111, 111, 79, 125, 126, 111, 111, 79, 108, 123, 93,
        121, 140, 61, 154,
        170, 154, 139, 153, 139, 123, 123, 63, 124, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154, 166, 183, 140, 136, 153, 154,
        170, 153, 138, 138, 122, 121, 122, 121, 167, 151, 183, 140, 151, 183, 140,
        154, 196, 167, 167, 154, 152, 167, 182, 182, 134, 149, 136, 153, 121, 136, 122, 169, 208, 166, 167, 154, 152, 167, 182,
        107, 167, 91, 107, 107, 167,
        154, 154, 154, 154, 154, 154, 154, 154, 154, 154,
        140, 140, 139, 139, 139, 139 } };
#undef X
/* intra_chroma_pred_mode has one context; the second value of the element's table row belongs to it in no initialisation type */

typedef struct Cabac {
    Bits *out;
    uint32_t low, range;
    int outstanding, first;
    uint8_t state[N_CTX];                             /* pStateIdx << 1 | valMps */
} Cabac;

static void cabac_start(Cabac *c, Bits *out) { c->out = out; c->low = 0; c->range = 510; c->outstanding = 0; c->first = 1; }
static void cabac_init_contexts(Cabac *c, int init_type, int qp)
{
    if (qp < 0) qp = 0;
    if (qp > 51) qp = 51;
    for (int i = 0; i < N_CTX; i++) {
        const int v = k_init[init_type][i], m = (v >> 4) * 5 - 45, n = ((v & 15) << 3) - 16;
        int pre = ((m * qp) >> 4) + n;
        pre = pre < 1 ? 1 : (pre > 126 ? 126 : pre);
        c->state[i] = pre <= 63 ? (uint8_t)((63 - pre) << 1) : (uint8_t)(((pre - 64) << 1) | 1);
    }
}
static void cabac_put(Cabac *c, int b)
{
    if (c->first) c->first = 0; else put_bit(c->out, b);
    for (; c->outstanding > 0; c->outstanding--) put_bit(c->out, !b);
}
static void cabac_renorm(Cabac *c)
{
    while (c->range < 256) {
        if (c->low < 256) cabac_put(c, 0);
        else if (c->low >= 512) { c->low -= 512; cabac_put(c, 1); }
        else { c->low -= 256; c->outstanding++; }
        c->range <<= 1; c->low <<= 1;
    }
}
static void enc_bin(Cabac *c, int ctx, int bin)
{
    uint8_t *st = &c->state[ctx];
    const int p = *st >> 1, mps = *st & 1;
    const uint32_t lps = k_range_lps[p][(c->range >> 6) & 3];
    c->range -= lps;
    if (bin != mps) {
        c->low += c->range; c->range = lps;
        *st = (uint8_t)((k_next_lps[p] << 1) | (p == 0 ? !mps : mps));
    } else {
        *st = (uint8_t)(((p < 62 ? p + 1 : 62) << 1) | mps);
    }
    cabac_renorm(c);
}
static void enc_bypass(Cabac *c, int bin)
{
    c->low <<= 1;
    if (bin) c->low += c->range;
    if (c->low >= 1024) { cabac_put(c, 1); c->low -= 1024; }
    else if (c->low < 512) cabac_put(c, 0);
    else { c->low -= 512; c->outstanding++; }
}
static void enc_bypass_bits(Cabac *c, uint32_t v, int n) { for (int i = n - 1; i >= 0; i--) enc_bypass(c, (v >> i) & 1); }
/* terminate bin; when it is 1 the engine is flushed (9.3.4.5): the last bit written is the stop / alignment one bit */
static void enc_terminate(Cabac *c, int bin)
{
    c->range -= 2;
    if (bin) {
        c->low += c->range; c->range = 2;
        cabac_renorm(c);
        cabac_put(c, (c->low >> 9) & 1);
        put_bits(c->out, ((c->low >> 7) & 3) | 1, 2);
    } else {
        cabac_renorm(c);
    }
}

/* ================================================================================================= trace (tests) */
/* what was written, element by element (OhStreamParams.trace): compared with what the reference decoder parsed */
static int32_t *g_trace;
static size_t g_trace_n, g_trace_cap;
static int g_trace_on;
static void tr(int id, int v)
{
    if (!g_trace_on) return;
    if (g_trace_n + 2 > g_trace_cap) { g_trace_cap = g_trace_cap ? 2 * g_trace_cap : 1 << 16; g_trace = (int32_t *)realloc(g_trace, g_trace_cap * sizeof(int32_t)); }
    g_trace[g_trace_n++] = id; g_trace[g_trace_n++] = v;
}
size_t oh_stream_trace(const int32_t **recs) { *recs = g_trace; return g_trace_n / 2; }

/* the quantised levels of every residual block written, for the sparse hand-over (OhStreamParams.levels, include/ohevc_stream.h) */
static uint32_t *g_lev;
static size_t g_lev_n, g_lev_cap;
static int g_lev_on;
static void lev_put(uint32_t v)
{
    if (g_lev_n + 1 > g_lev_cap) { g_lev_cap = g_lev_cap ? 2 * g_lev_cap : 1 << 16; g_lev = (uint32_t *)realloc(g_lev, g_lev_cap * sizeof(uint32_t)); }
    g_lev[g_lev_n++] = v;
}
size_t oh_stream_levels(const uint32_t **words) { *words = g_lev; return g_lev_n; }

/* chroma QP of a block (8.6.1: qPi -> QpC for ChromaArrayType 1); the PPS carries the cb / cr offsets (+1 / -2 unless OhStreamParams says otherwise), no slice offsets */
static int g_cb_off = 1, g_cr_off = -2;               /* pps_cb_qp_offset / pps_cr_qp_offset of the stream being written */
static int chroma_qp(int qp_y, int c_idx, int bit_depth, int chroma_format_idc)
{
    static const uint8_t tab[14] = { 29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37 };     /* qPi 30..43 */
    const int bd_off = 6 * (bit_depth - 8);
    int qpi = qp_y + (c_idx == 1 ? g_cb_off : g_cr_off);
    qpi = qpi < -bd_off ? -bd_off : qpi > 57 ? 57 : qpi;
    const int qpc = chroma_format_idc != 1 ? (qpi < 51 ? qpi : 51) : qpi < 30 ? qpi : qpi >= 43 ? qpi - 6 : tab[qpi - 30];
    return qpc + bd_off;
}

/* ================================================================================================= generator state */
typedef struct Rng { uint64_t s; } Rng;
static uint64_t rnd64(Rng *g)
{
    uint64_t z = (g->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static int rnd(Rng *g, int n) { return n <= 1 ? 0 : (int)(rnd64(g) % (uint64_t)n); }
static int pct(Rng *g, int p) { return rnd(g, 100) < p; }

enum { SLICE_B = 0, SLICE_P = 1, SLICE_I = 2 };
enum { PART_2Nx2N = 0, PART_2NxN, PART_Nx2N, PART_NxN, PART_2NxnU, PART_2NxnD, PART_nLx2N, PART_nRx2N };

typedef struct Pic {                                   /* per-picture maps, 4x4 luma granularity unless noted */
    int w4, h4;
    uint8_t *skip, *depth, *intra, *ipm, *pcm;        /* cu_skip_flag, ct_depth, pred mode intra, IntraPredModeY, pcm / outside */
} Pic;

typedef struct Slice {
    int type, addr, qp, n_ref[2], max_merge, cabac_init_flag, tmvp, mvd_l1_zero, deblock_disabled, lf_across, dependent;
    int sao_luma, sao_chroma;
} Slice;

typedef struct W {
    const OhStreamParams *p;
    Rng g;
    Bits out;                                          /* the stream */
    Cabac c;
    Pic pic;
    int ctb, lc, ctbw, ctbh, n_ctb, min_cb_log2;
    int *slice_of;                                     /* per CTB (raster): slice address it belongs to */
    int *tile_of, *rs_of_ts, *ts_of_rs;
    int col_bd[64], row_bd[64], tcols, trows;
    Slice sl;
    int qp_delta_pending, cu_bypass;
    int cross_pf, res_scale;                           /* cross-component prediction of the chroma block being coded */
    int tt_pu;                                         /* transform tree: index of the depth-1 block being coded = the partition of an intra NxN coding block */
    int stat_coeff[4];                                 /* StatCoeff (9.3.3.11): reset with the contexts */
    int el;                                            /* enhancement layer of a two-layer stream: parameter-set ids 1, inter-layer prediction only */
    int bl_w, bl_h;                                    /* two-layer stream: the base layer's size (VPS rep_format 0) */
    uint8_t wpp_ctx[N_CTX]; int have_wpp;              /* the context states after the second CTB of the last row that had two (wavefront synchronisation) */
    /* SAO parameters of the CTBs (for merge candidates we only need to know that they exist) */
} W;

/* ---- neighbourhood ---- */
static int ctb_of(const W *w, int x, int y) { return (y >> w->lc) * w->ctbw + (x >> w->lc); }
/* z-scan availability (6.4.1) of the luma position (xn, yn) seen from (x, y): inside the picture, already coded, same slice, same tile */
static int avail(const W *w, int x, int y, int xn, int yn)
{
    if (xn < 0 || yn < 0 || xn >= w->p->width || yn >= w->p->height)
        return 0;
    const int a = ctb_of(w, x, y), b = ctb_of(w, xn, yn);
    if (w->slice_of[a] != w->slice_of[b] || w->tile_of[a] != w->tile_of[b])
        return 0;
    return w->pic.depth[(yn >> 2) * w->pic.w4 + (xn >> 2)] != 0xff;      /* 0xff: not coded yet */
}
static uint8_t *cell(uint8_t *m, const W *w, int x, int y) { return &m[(y >> 2) * w->pic.w4 + (x >> 2)]; }
static void fill(uint8_t *m, const W *w, int x, int y, int n, int v)
{
    for (int yy = y; yy < y + n && yy < w->p->height; yy += 4)
        for (int xx = x; xx < x + n && xx < w->p->width; xx += 4)
            *cell(m, w, xx, yy) = (uint8_t)v;
}

/* ================================================================================================= parameter sets */
static int rext_profile(const OhStreamParams *p)
{
    return p->chroma_format_idc >= 2 || p->bit_depth > 10 || p->tskip_rotation || p->tskip_context || p->implicit_rdpcm || p->explicit_rdpcm || p->intra_smoothing_disabled ||
           p->persistent_rice || p->log2_max_tskip_size > 2;
}

static int max_tskip(const OhStreamParams *p) { return p->log2_max_tskip_size > 2 ? p->log2_max_tskip_size : 2; }

static void write_ptl(Bits *b, const OhStreamParams *p)
{
    const int profile = rext_profile(p) ? 4 : p->bit_depth > 8 ? 2 : 1;                  /* Main / Main 10 / format range extensions */
    put_bits(b, 0, 2); put_bit(b, 0); put_bits(b, (uint32_t)profile, 5);                  /* profile space, tier, profile */
    for (int i = 0; i < 32; i++) put_bit(b, i == profile || (profile < 4 && i == 2));     /* compatibility flags */
    put_bit(b, 1); put_bit(b, 0); put_bit(b, 0); put_bit(b, 1);                           /* progressive, interlaced, non-packed, frame-only */
    put_bits(b, 0, 16); put_bits(b, 0, 16); put_bits(b, 0, 12);                           /* reserved 44 bits */
    put_bits(b, 186, 8);                                                                  /* level 6.2 */
}

/* The video parameter set of a TWO-LAYER (SHVC spatial scalability) stream, in the syntax the reference parses (hevc_ps.c:714-1097:
 * the SHM 4.1 draft its decoder was written against — not the final annex F): base layer 0, enhancement layer 1 directly dependent on
 * it (sample and motion prediction), one layer set with both, representation formats in the VPS (the EL's SPS carries neither size
 * nor bit depth: update_rep_format_flag 0).  Every value is the plainest the syntax allows; the comments name the element read at
 * each step of parse_vps_extension. */
static void write_vps_two_layers(W *w, int el_w, int el_h)
{
    const OhStreamParams *p = w->p;
    Bits b = { 0 };
    put_bits(&b, 0, 4); put_bits(&b, 3, 2); put_bits(&b, 1, 6); put_bits(&b, 0, 3); put_bit(&b, 1); put_bits(&b, 0xffff, 16);   /* id, reserved, max_layers_minus1 = 1, sub-layers, nesting, extension offset */
    write_ptl(&b, p);
    put_bit(&b, 1);                                        /* sub_layer_ordering_info_present */
    put_ue(&b, (uint32_t)p->n_refs + 1); put_ue(&b, 0); put_ue(&b, 0);
    put_bits(&b, 1, 6); put_ue(&b, 1);                     /* vps_max_layer_id = 1, vps_num_layer_sets_minus1 = 1 */
    put_bit(&b, 1); put_bit(&b, 1);                        /* layer_id_included_flag[1][0..1] */
    put_bit(&b, 0);                                        /* timing info */
    put_bit(&b, 1);                                        /* vps_extension_flag */
    while (b.n & 7) put_bit(&b, 1);                        /* vps_extension_alignment_bit_equal_to_one */
    /* ---- vps_extension ---- */
    put_bit(&b, 0); put_bit(&b, 0);                        /* avc_base_layer_flag, splitting_flag */
    for (int i = 0; i < 16; i++) put_bit(&b, i == 2);      /* scalability_mask: bit 2 = spatial / quality scalability (SCALABILITY_ID, hevc.h:133-137) */
    put_bits(&b, 0, 3);                                    /* dimension_id_len_minus1[0]: one bit */
    put_bit(&b, 0);                                        /* vps_nuh_layer_id_present_flag: layer_id_in_nuh[1] = 1 */
    put_bit(&b, 1);                                        /* dimension_id[1][0] = 1 */
    put_bits(&b, 0, 4);                                    /* view_id_len_minus1 */
    put_bit(&b, 0);                                        /* view_id_val[0] (the parser counts one view) */
    put_bit(&b, 1);                                        /* direct_dependency_flag[1][0] */
    put_bit(&b, 0);                                        /* vps_sub_layers_max_minus1_present_flag */
    put_bit(&b, 0);                                        /* max_tid_ref_present_flag */
    put_bit(&b, 1);                                        /* all_ref_layers_active_flag */
    put_bits(&b, 1, 10);                                   /* vps_number_layer_sets_minus1 (must equal vps_num_layer_sets - 1) */
    put_bits(&b, 0, 6);                                    /* vps_num_profile_tier_level_minus1 */
    put_bit(&b, 0);                                        /* more_output_layer_sets_than_default_flag */
    put_bit(&b, 1);                                        /* default_one_target_output_layer_flag (numOutputLayerSets = 2 > 1) */
    put_bit(&b, 0);                                        /* profile_level_tier_idx[1]: one bit */
    put_bit(&b, 0);                                        /* alt_output_layer_flag */
    put_bit(&b, 0);                                        /* rep_format_idx_present_flag: a format per layer, vps_rep_format_idx[i] = i */
    for (int l = 0; l < 2; l++) {                          /* rep_format(): hevc_ps.c:411-467 */
        put_bit(&b, 1);                                    /* chroma_and_bit_depth_vps_present_flag */
        put_bits(&b, (uint32_t)(l ? el_w : p->width), 16); put_bits(&b, (uint32_t)(l ? el_h : p->height), 16);
        put_bits(&b, 1, 2);                                /* chroma_format_idc 4:2:0 */
        put_bits(&b, (uint32_t)p->bit_depth - 8, 4); put_bits(&b, (uint32_t)p->bit_depth - 8, 4);
    }
    put_bit(&b, 1);                                        /* max_one_active_ref_layer_flag */
    /* poc_lsb_not_present_flag: only for layers without a reference layer — none */
    put_bit(&b, 0);                                        /* cross_layer_phase_alignment_flag: zero-position-aligned */
    put_bit(&b, 0);                                        /* sub_layer_flag_info_present_flag[1] */
    put_ue(&b, (uint32_t)p->n_refs + 1);                   /* max_vps_dec_pic_buffering_minus1[1][0][0] (one sub-DPB) */
    put_ue(&b, 0); put_ue(&b, 0);                          /* max_vps_num_reorder_pics, max_vps_latency_increase_plus1 */
    put_ue(&b, 0);                                         /* direct_dep_type_len_minus2 */
    put_bit(&b, 1); put_bits(&b, 2, 2);                    /* default_direct_dependency_type_flag, type 2: samples and motion */
    put_bit(&b, 0); put_bit(&b, 0);                        /* single_layer_for_non_irap_flag, higher_layer_irap_skip_flag */
    put_bit(&b, 0);                                        /* vps_vui_present_flag */
    rbsp_trailing(&b);
    emit_nal(&w->out, 32, b.buf, b.n / 8);
    free(b.buf);
}

static void write_vps(W *w)
{
    Bits b = { 0 };
    put_bits(&b, 0, 4); put_bits(&b, 3, 2); put_bits(&b, 0, 6); put_bits(&b, 0, 3); put_bit(&b, 1); put_bits(&b, 0xffff, 16);
    write_ptl(&b, w->p);
    put_bit(&b, 1);                                        /* sub_layer_ordering_info_present */
    put_ue(&b, (uint32_t)w->p->n_refs + 1); put_ue(&b, w->p->gop == 3 ? 2 : 0); put_ue(&b, 0);   /* max_dec_pic_buffering_minus1, num_reorder, max_latency_plus1 */
    put_bits(&b, 0, 6); put_ue(&b, 0);                     /* max_layer_id, num_layer_sets_minus1 */
    put_bit(&b, 0);                                        /* timing info */
    put_bit(&b, 0);                                        /* extension */
    rbsp_trailing(&b);
    emit_nal(&w->out, 32, b.buf, b.n / 8);
    free(b.buf);
}

static void write_sps(W *w)
{
    const OhStreamParams *p = w->p;
    Bits b = { 0 };
    if (w->el) {
        /* the sequence parameter set of the enhancement layer (hevc_ps.c:1520-1760 with nuh_layer_id > 0): no sub-layer count, no
         * profile / tier / level, no size, no bit depth — the representation format comes from the VPS */
        put_bits(&b, 0, 4);                                /* vps id */
        put_ue(&b, 1);                                     /* sps id 1 (the decoder of layer 1 looks for the base layer's SPS at id 0, hevc.c:453) */
        put_bit(&b, 0);                                    /* update_rep_format_flag */
    } else {
    put_bits(&b, 0, 4); put_bits(&b, 0, 3); put_bit(&b, 1);                      /* vps id, max_sub_layers_minus1, temporal_id_nesting */
    write_ptl(&b, p);
    put_ue(&b, 0);                                         /* sps id */
    put_ue(&b, (uint32_t)p->chroma_format_idc);            /* 1 = 4:2:0, 2 = 4:2:2, 3 = 4:4:4 */
    if (p->chroma_format_idc == 3) put_bit(&b, 0);         /* separate_colour_plane_flag */
    put_ue(&b, (uint32_t)p->width); put_ue(&b, (uint32_t)p->height);
    }
    {   /* conformance window, offsets in chroma sample units (4:2:0: two luma samples) */
        const int any = p->conf_win_left | p->conf_win_right | p->conf_win_top | p->conf_win_bottom;
        put_bit(&b, any != 0);
        if (any) {
            put_ue(&b, (uint32_t)p->conf_win_left / 2); put_ue(&b, (uint32_t)p->conf_win_right / 2);
            put_ue(&b, (uint32_t)p->conf_win_top / 2); put_ue(&b, (uint32_t)p->conf_win_bottom / 2);
        }
    }
    if (!w->el) { put_ue(&b, (uint32_t)p->bit_depth - 8); put_ue(&b, (uint32_t)p->bit_depth - 8); }
    put_ue(&b, 4);                                         /* log2_max_poc_lsb = 8 */
    put_bit(&b, 1);
    put_ue(&b, (uint32_t)p->n_refs + 1); put_ue(&b, p->gop == 3 ? 2 : 0); put_ue(&b, 0);
    put_ue(&b, (uint32_t)w->min_cb_log2 - 3);              /* log2_min_luma_coding_block_size_minus3 */
    put_ue(&b, (uint32_t)(p->log2_ctb_size - w->min_cb_log2));
    put_ue(&b, (uint32_t)p->log2_min_tb_size - 2);
    put_ue(&b, (uint32_t)(p->log2_max_tb_size - p->log2_min_tb_size));
    put_ue(&b, (uint32_t)p->max_th_depth_inter); put_ue(&b, (uint32_t)p->max_th_depth_intra);
    put_bit(&b, p->scaling_list != 0);
    if (p->scaling_list) {
        put_bit(&b, p->scaling_list > 1);                  /* sps_scaling_list_data_present_flag: 0 = the default lists */
        if (p->scaling_list > 1) {                         /* scaling_list_data() 7.3.4 with random lists */
            Rng sg = { p->seed * 0x9E3779B97F4A7C15ull + 4242 };
            for (int size_id = 0; size_id < 4; size_id++)
                for (int matrix_id = 0; matrix_id < 6; matrix_id += size_id == 3 ? 3 : 1) {
                    const int kind = rnd(&sg, 4);          /* 0: the default list, 1: copy of the previous matrix, else explicit */
                    if (kind < 2) {
                        put_bit(&b, 0);
                        put_ue(&b, kind == 1 && matrix_id && size_id != 3 ? 1 : 0);   /* scaling_list_pred_matrix_id_delta; 0 = the default list */
                        continue;
                    }
                    put_bit(&b, 1);
                    int next = 8;
                    if (size_id > 1) { const int dc = 1 + rnd(&sg, 64); put_se(&b, dc - 8); next = dc; }
                    const int coef_num = size_id == 0 ? 16 : 64;
                    for (int i = 0; i < coef_num; i++) {
                        int v = next + rnd(&sg, 33) - 16;
                        if (rnd(&sg, 16) == 0) v = 1 + rnd(&sg, 255);
                        v = v < 1 ? 1 : v > 255 ? 255 : v;
                        int d = v - next;
                        if (d > 127) d -= 256;
                        if (d < -128) d += 256;
                        put_se(&b, d);
                        next = v;
                    }
                }
        }
    }
    put_bit(&b, p->amp != 0);
    put_bit(&b, p->sao != 0);
    put_bit(&b, p->pcm != 0);
    if (p->pcm) {
        put_bits(&b, (uint32_t)p->bit_depth - 1, 4); put_bits(&b, (uint32_t)p->bit_depth - 1, 4);     /* PCM samples at full depth */
        put_ue(&b, (uint32_t)w->min_cb_log2 - 3);          /* log2_min_pcm: the smallest coding block */
        put_ue(&b, (uint32_t)((p->log2_ctb_size < 5 ? p->log2_ctb_size : 5) - w->min_cb_log2));
        put_bit(&b, p->pcm_loop_filter == 0);              /* pcm_loop_filter_disabled_flag */
    }
    put_ue(&b, 0);                                         /* no short-term RPS in the SPS: slices carry theirs */
    put_bit(&b, 0);                                        /* long-term refs */
    put_bit(&b, p->tmvp != 0);
    put_bit(&b, p->strong_intra_smoothing != 0);
    put_bit(&b, 0);                                        /* VUI */
    if (rext_profile(p)) {                                 /* sps_range_extension (7.3.2.2.2) */
        put_bit(&b, 1); put_bit(&b, 1); put_bits(&b, 0, 7);
        put_bit(&b, p->tskip_rotation != 0); put_bit(&b, p->tskip_context != 0);
        put_bit(&b, p->implicit_rdpcm != 0); put_bit(&b, p->explicit_rdpcm != 0);
        put_bit(&b, 0);                                    /* extended_precision_processing: not in the reference's kernels */
        put_bit(&b, p->intra_smoothing_disabled != 0);
        put_bit(&b, 0);                                    /* high_precision_offsets */
        put_bit(&b, p->persistent_rice != 0);
        put_bit(&b, 0);                                    /* cabac_bypass_alignment */
    } else {
        put_bit(&b, 0);                                    /* extension */
    }
    rbsp_trailing(&b);
    emit_nal(&w->out, 33, b.buf, b.n / 8);
    free(b.buf);
}

static void write_pps(W *w)
{
    const OhStreamParams *p = w->p;
    Bits b = { 0 };
    put_ue(&b, (uint32_t)w->el); put_ue(&b, (uint32_t)w->el);      /* pps id, sps id: 0 / 0, the enhancement layer's 1 / 1 */
    put_bit(&b, p->dependent_slices != 0);
    put_bit(&b, 0); put_bits(&b, 0, 3);                    /* output_flag_present, extra slice header bits */
    put_bit(&b, p->sign_data_hiding != 0);
    put_bit(&b, p->cabac_init_present != 0);
    put_ue(&b, (uint32_t)p->n_refs - 1); put_ue(&b, (uint32_t)p->n_refs - 1);
    put_se(&b, 0);                                         /* init_qp 26 */
    put_bit(&b, p->constrained_intra_pred != 0);
    put_bit(&b, p->transform_skip != 0);
    put_bit(&b, p->cu_qp_delta != 0);
    if (p->cu_qp_delta) put_ue(&b, (uint32_t)(p->log2_ctb_size > w->min_cb_log2));   /* diff_cu_qp_delta_depth: quantisation groups of half a CTB (a whole one when that is the smallest coding block) */
    put_se(&b, g_cb_off); put_se(&b, g_cr_off);            /* pps_cb_qp_offset / pps_cr_qp_offset */
    put_bit(&b, 0);                                        /* slice-level chroma qp offsets */
    put_bit(&b, p->weighted_pred != 0); put_bit(&b, p->weighted_pred != 0);
    put_bit(&b, p->transquant_bypass != 0);
    const int tiles = w->tcols > 1 || w->trows > 1;
    put_bit(&b, tiles);
    put_bit(&b, p->wpp != 0);
    if (tiles) {
        put_ue(&b, (uint32_t)w->tcols - 1); put_ue(&b, (uint32_t)w->trows - 1);
        put_bit(&b, 1);                                    /* uniform spacing */
        put_bit(&b, p->lf_across_tiles != 0);
    }
    put_bit(&b, p->lf_across_slices != 0);
    put_bit(&b, 1);                                        /* deblocking_filter_control_present */
    put_bit(&b, p->deblocking_override != 0);
    put_bit(&b, 0);                                        /* pps_deblocking_filter_disabled */
    put_se(&b, 1); put_se(&b, -1);                         /* beta_offset_div2, tc_offset_div2 */
    if (w->el) put_bit(&b, 0);                             /* pps_infer_scaling_list_flag: a layer-1 PPS carries it (hevc_ps.c:2380-2385) */
    put_bit(&b, 0);                                        /* scaling list data */
    put_bit(&b, 0);                                        /* lists modification */
    put_ue(&b, 0);                                         /* log2_parallel_merge_level 2 */
    put_bit(&b, 0);                                        /* slice header extension */
    if (rext_profile(p)) {                                 /* pps_range_extension (7.3.2.3.2) */
        put_bit(&b, 1); put_bit(&b, 1); put_bits(&b, 0, 7);
        if (p->transform_skip) put_ue(&b, (uint32_t)(max_tskip(p) - 2));   /* log2_max_transform_skip_block_size_minus2 */
        put_bit(&b, p->cross_component_pred != 0);
        put_bit(&b, 0);                                    /* chroma_qp_offset_list_enabled_flag */
        put_ue(&b, (uint32_t)p->sao_offset_scale_luma); put_ue(&b, (uint32_t)p->sao_offset_scale_chroma);   /* log2_sao_offset_scale_luma / chroma (0 .. bit depth - 10) */
    } else {
        put_bit(&b, 0);                                    /* pps extension */
    }
    rbsp_trailing(&b);
    emit_nal(&w->out, 34, b.buf, b.n / 8);
    free(b.buf);
}

/* ================================================================================================= slice data */
/* ---- SAO (7.3.8.3) ---- */
static void code_sao(W *w, int rx, int ry)
{
    Cabac *c = &w->c;
    const int x = rx << w->lc, y = ry << w->lc;
    if (!w->sl.sao_luma && !w->sl.sao_chroma)
        return;
    int merge = 0;
    if (rx > 0 && avail(w, x, y, x - 1, y)) { merge = pct(&w->g, 25); enc_bin(c, C_SAO_MERGE, merge); tr(OH_SE_SAO_MERGE, merge); }
    if (!merge && ry > 0 && avail(w, x, y, x, y - 1)) { merge = pct(&w->g, 25); enc_bin(c, C_SAO_MERGE, merge); tr(OH_SE_SAO_MERGE, merge); }
    if (merge)
        return;
    const int max_off = (1 << ((w->p->bit_depth < 10 ? w->p->bit_depth : 10) - 5)) - 1;
    int type_c = 0;
    for (int ci = 0; ci < 3; ci++) {
        if ((ci == 0 && !w->sl.sao_luma) || (ci > 0 && !w->sl.sao_chroma))
            continue;
        int type;
        if (ci == 2) {
            type = type_c;
        } else {
            type = pct(&w->g, w->p->sao_pct) ? 1 + rnd(&w->g, 2) : 0;           /* 1 band, 2 edge */
            enc_bin(c, C_SAO_TYPE, type != 0);
            if (type) enc_bypass(c, type == 2);
            tr(OH_SE_SAO_TYPE, type);
            if (ci == 1) type_c = type;
        }
        if (!type)
            continue;
        int off[4];
        for (int k = 0; k < 4; k++) {
            off[k] = rnd(&w->g, max_off + 1);
            for (int u = 0; u < off[k]; u++) enc_bypass(c, 1);
            if (off[k] < max_off) enc_bypass(c, 0);
            tr(OH_SE_SAO_OFFSET_ABS, off[k]);
        }
        if (type == 1) {
            for (int k = 0; k < 4; k++)
                if (off[k]) { const int sg = rnd(&w->g, 2); enc_bypass(c, sg); tr(OH_SE_SAO_OFFSET_SIGN, sg); }
            const int bp = rnd(&w->g, 32);
            enc_bypass_bits(c, (uint32_t)bp, 5); tr(OH_SE_SAO_BAND_POS, bp);
        } else if (ci != 2) {
            const int eo = rnd(&w->g, 4);
            enc_bypass_bits(c, (uint32_t)eo, 2); tr(OH_SE_SAO_EO_CLASS, eo);
        }
    }
}

/* ---- residual_coding (7.3.8.11) ---- */
static const uint8_t k_sig_map4[16] = { 0, 1, 4, 5, 2, 3, 4, 5, 6, 6, 8, 8, 7, 7, 8, 8 };

/* position k of the 4x4 (or sub-block) scans: scan 0 up-right diagonal, 1 horizontal, 2 vertical (6.5.3-6.5.5) */
static void scan_pos(int scan, int n_log2, int k, int *x, int *y)
{
    const int n = 1 << n_log2;
    if (scan == 1) { *x = k & (n - 1); *y = k >> n_log2; return; }
    if (scan == 2) { *y = k & (n - 1); *x = k >> n_log2; return; }
    int i = 0, xx = 0, yy = 0;
    for (;;) {                                             /* up-right diagonal: from bottom-left to top-right, diagonal after diagonal */
        while (yy >= 0) {
            if (xx < n && yy < n) { if (i == k) { *x = xx; *y = yy; return; } i++; }
            yy--; xx++;
        }
        yy = xx; xx = 0;
    }
}

static void code_last_prefix(Cabac *c, int base, int v, int log2, int c_idx)
{
    int off, shift;
    if (c_idx == 0) { off = 3 * (log2 - 2) + ((log2 - 1) >> 2); shift = (log2 + 1) >> 2; }
    else { off = 15; shift = log2 - 2; }
    const int cmax = (log2 << 1) - 1;
    for (int i = 0; i < v; i++) enc_bin(c, base + off + (i >> shift), 1);
    if (v < cmax) enc_bin(c, base + off + (v >> shift), 0);
}

static void code_remaining(Cabac *c, int value, int rice)
{
    const int prefix = value >> rice;
    if (prefix < 4) {
        for (int i = 0; i < prefix; i++) enc_bypass(c, 1);
        enc_bypass(c, 0);
        enc_bypass_bits(c, (uint32_t)(value - (prefix << rice)), rice);
    } else {
        int v = value - (4 << rice), len = 0, base = 0;    /* escape: four ones, then exp-Golomb of order rice + 1 */
        while (v >= base + (1 << (rice + 1 + len))) { base += 1 << (rice + 1 + len); len++; }
        for (int i = 0; i < 4 + len; i++) enc_bypass(c, 1);
        enc_bypass(c, 0);
        enc_bypass_bits(c, (uint32_t)(v - base), rice + 1 + len);
    }
}

/* a chroma block without coefficients whose residual is the scaled luma residual (cross-component prediction with cbf = 0,
 * hevc.c:1315-1330): no syntax, but a block of the levels log */
static void log_cross_only(W *w, int log2, int c_idx)
{
    if (!g_lev_on) return;
    lev_put((uint32_t)log2 | (uint32_t)c_idx << 4 | (uint32_t)(w->cu_bypass != 0) << 9 | 1u << 11 | 1u << 12);   /* bit 12: no residual_coding() */
    lev_put((uint32_t)(w->res_scale & 0xff) << 24);
}

/* codes one transform block with random coefficients; scan: 0 diagonal, 1 horizontal, 2 vertical */
static void code_residual(W *w, int log2, int c_idx, int scan, int cu_intra, int mode)      /* mode: the block's intra prediction mode (-1: inter) */
{
    Cabac *c = &w->c;
    Rng *g = &w->g;
    const int n = 1 << log2, n_sb_log2 = log2 - 2;
    int16_t lev[32 * 32];
    memset(lev, 0, sizeof(int16_t) * (size_t)(n * n));
    /* random levels: low-frequency corner mostly, a few larger values, rarely saturating ones */
    {
        const int shape = rnd(g, 10), dens = w->p->coeff_density;
        int nz = shape < 3 ? 1 : 1 + rnd(g, (shape < 8 ? (n * n / 8 > 1 ? n * n / 8 : 2) : n * n / 2) * dens / 100 + 1);
        const int lim = shape < 8 ? (n > 8 ? n / 2 : n) : n;
        for (int i = 0; i < nz; i++) {
            int x = shape < 3 ? 0 : rnd(g, lim), y = shape < 3 ? 0 : rnd(g, lim);
            int m = 1 + rnd(g, 3) * rnd(g, 4) * rnd(g, (x + y == 0) ? 12 : 3);
            if (rnd(g, 200) == 0) m = 1 + rnd(g, 32767);
            lev[y * n + x] = (int16_t)(rnd(g, 2) ? m : -m);
        }
    }
    tr(OH_SE_RESIDUAL, log2 | (c_idx << 4) | (scan << 8));
    int tskip = 0, rdpcm = 0, rdpcm_dir = 0;
    if (w->p->transform_skip && !w->cu_bypass && log2 <= max_tskip(w->p)) {
        tskip = pct(g, w->p->tskip_pct);
        enc_bin(c, C_TSKIP + (c_idx ? 1 : 0), tskip);
    }
    if (!cu_intra && w->p->explicit_rdpcm && (tskip || w->cu_bypass)) {
        rdpcm = rnd(g, 2);
        enc_bin(c, C_RDPCM + (c_idx ? 1 : 0), rdpcm);
        if (rdpcm) { rdpcm_dir = rnd(g, 2); enc_bin(c, C_RDPCM_DIR + (c_idx ? 1 : 0), rdpcm_dir); }
    }
    const int ts_ctx = w->p->tskip_context && (tskip || w->cu_bypass);       /* one sig_coeff_flag context for the whole block */
    /* last significant coefficient in scan order */
    int last_sb = -1, last_pos = -1, lx = 0, ly = 0;
    const int n_sb = 1 << (2 * n_sb_log2);
    for (int i = n_sb - 1; i >= 0 && last_sb < 0; i--) {
        int sx, sy;
        scan_pos(scan, n_sb_log2, i, &sx, &sy);
        for (int k = 15; k >= 0; k--) {
            int px, py;
            scan_pos(scan, 2, k, &px, &py);
            if (lev[(sy * 4 + py) * n + sx * 4 + px]) { last_sb = i; last_pos = k; lx = sx * 4 + px; ly = sy * 4 + py; break; }
        }
    }
    if (last_sb < 0) { lev[0] = 1; last_sb = 0; last_pos = 0; lx = ly = 0; }
    /* sign data hiding (9.3.4.3 / hevc_cabac.c:1745-1760, 1810-1814): in a sub-block whose first and last non-zero positions are at
     * least four scan positions apart the sign of the FIRST one is not coded, it is the parity of the sub-block's sum of magnitudes —
     * the writer owns the content, so it gives that coefficient the sign the parity says */
    const int sdh = w->p->sign_data_hiding && !w->cu_bypass && !rdpcm &&
                    !(cu_intra && w->p->implicit_rdpcm && tskip && (mode == 10 || mode == 26));
    if (sdh)
        for (int i = 0; i < n_sb; i++) {
            int sx, sy, first = -1, last = -1, sum = 0, fx = 0, fy = 0;
            scan_pos(scan, n_sb_log2, i, &sx, &sy);
            for (int k = 0; k < 16; k++) {
                int px, py;
                scan_pos(scan, 2, k, &px, &py);
                const int v = lev[(sy * 4 + py) * n + sx * 4 + px];
                if (!v) continue;
                if (first < 0) { first = k; fx = sx * 4 + px; fy = sy * 4 + py; }
                last = k;
                sum += abs(v);
            }
            if (first >= 0 && last - first >= 4) {
                const int a = abs(lev[fy * n + fx]);
                lev[fy * n + fx] = (int16_t)((sum & 1) ? -a : a);
            }
        }
    if (g_lev_on) {
        /* header: log2 | c_idx << 4 | transform_skip << 8 | cu_transquant_bypass << 9 | intra CU << 10 | cross-component prediction << 11 |
         * QP (with QpBdOffset; valid while cu_qp_delta is off: the slice QP) << 16; then the number of levels | res_scale_val << 24;
         * then pos | level << 16 in raster order */
        const int bd_off = 6 * (w->p->bit_depth - 8);
        const int qp = c_idx ? chroma_qp(w->sl.qp, c_idx, w->p->bit_depth, w->p->chroma_format_idc) : w->sl.qp + bd_off;
        lev_put((uint32_t)log2 | (uint32_t)c_idx << 4 | (uint32_t)tskip << 8 | (uint32_t)(w->cu_bypass != 0) << 9 | (uint32_t)(cu_intra != 0) << 10 |
                (uint32_t)(w->cross_pf != 0 && c_idx != 0) << 11 | (uint32_t)rdpcm << 13 | (uint32_t)rdpcm_dir << 14 | (uint32_t)qp << 16);
        uint32_t cnt = 0;
        for (int i = 0; i < n * n; i++) cnt += lev[i] != 0;
        lev_put(cnt | (uint32_t)(c_idx ? w->res_scale & 0xff : 0) << 24);
        for (int i = 0; i < n * n; i++)
            if (lev[i]) lev_put((uint32_t)i | (uint32_t)(uint16_t)lev[i] << 16);
    }
    {
        int cx = lx, cy = ly;
        if (scan == 2) { cx = ly; cy = lx; }               /* coded swapped for the vertical scan */
        int px, py, sx = 0, sy = 0;
        if (cx < 4) px = cx; else { int l = 0; while ((cx >> (l + 1)) > 1) l++; px = 2 * (l + 1) + ((cx >> l) & 1); sx = l; }
        if (cy < 4) py = cy; else { int l = 0; while ((cy >> (l + 1)) > 1) l++; py = 2 * (l + 1) + ((cy >> l) & 1); sy = l; }
        code_last_prefix(c, C_LAST_X, px, log2, c_idx);
        code_last_prefix(c, C_LAST_Y, py, log2, c_idx);
        if (cx >= 4) enc_bypass_bits(c, (uint32_t)(cx & ((1 << sx) - 1)), sx);
        if (cy >= 4) enc_bypass_bits(c, (uint32_t)(cy & ((1 << sy) - 1)), sy);
    }
    uint8_t csbf[8][8];
    memset(csbf, 0, sizeof(csbf));
    int greater1_ctx = 1, first_sb_done = 0;
    for (int i = last_sb; i >= 0; i--) {
        int sx, sy;
        scan_pos(scan, n_sb_log2, i, &sx, &sy);
        const int right = sx + 1 < (1 << n_sb_log2) ? csbf[sy][sx + 1] : 0, below = sy + 1 < (1 << n_sb_log2) ? csbf[sy + 1][sx] : 0;
        int any = 0;
        for (int k = 0; k < 16; k++) {
            int px, py;
            scan_pos(scan, 2, k, &px, &py);
            any |= lev[(sy * 4 + py) * n + sx * 4 + px] != 0;
        }
        int infer_dc = 0;
        if (i < last_sb && i > 0) {
            enc_bin(c, C_CSBF + (c_idx ? 2 : 0) + (right | below), any);
            infer_dc = 1;
        } else {
            any = 1;                                       /* first and last sub-block: inferred coded */
        }
        csbf[sy][sx] = (uint8_t)any;
        if (!any)
            continue;
        /* significant_coeff_flag */
        int sig[16], n_sig = 0;
        const int start = i == last_sb ? last_pos - 1 : 15;
        if (i == last_sb) sig[n_sig++] = last_pos;
        const int prev = right | (below << 1);
        for (int k = start; k >= 0; k--) {
            int px, py;
            scan_pos(scan, 2, k, &px, &py);
            const int xc = sx * 4 + px, yc = sy * 4 + py, s = lev[yc * n + xc] != 0;
            if (k == 0 && infer_dc && n_sig == 0) {        /* the DC position of a coded sub-block with nothing else in it is inferred */
                sig[n_sig++] = 0;
                continue;
            }
            int sc;
            if (log2 == 2) sc = k_sig_map4[(yc << 2) + xc];
            else if (xc + yc == 0) sc = 0;
            else {
                if (prev == 0) sc = (px + py == 0) ? 2 : (px + py < 3) ? 1 : 0;
                else if (prev == 1) sc = py == 0 ? 2 : (py == 1 ? 1 : 0);
                else if (prev == 2) sc = px == 0 ? 2 : (px == 1 ? 1 : 0);
                else sc = 2;
                if (c_idx == 0) { if (sx || sy) sc += 3; sc += log2 == 3 ? (scan == 0 ? 9 : 15) : 21; }
                else sc += log2 == 3 ? 9 : 12;
            }
            enc_bin(c, ts_ctx ? C_SIG_TS + (c_idx ? 1 : 0) : C_SIG + (c_idx ? 27 : 0) + sc, s);
            if (s) { sig[n_sig++] = k; infer_dc = 0; }
        }
        if (!n_sig)
            continue;                                      /* cannot happen: a coded sub-block holds a coefficient */
        /* greater1 / greater2 flags */
        int ctx_set = (i == 0 || c_idx > 0) ? 0 : 2;
        if (first_sb_done && greater1_ctx == 0) ctx_set++;
        first_sb_done = 1;
        greater1_ctx = 1;
        int absv[16], g1[16], first_g1 = -1;
        for (int m = 0; m < n_sig; m++) {
            int px, py;
            scan_pos(scan, 2, sig[m], &px, &py);
            absv[m] = abs(lev[(sy * 4 + py) * n + sx * 4 + px]);
            g1[m] = 0;
        }
        const int n_g1 = n_sig < 8 ? n_sig : 8;
        for (int m = 0; m < n_g1; m++) {
            g1[m] = absv[m] > 1;
            enc_bin(c, C_GT1 + (c_idx ? 16 : 0) + ctx_set * 4 + greater1_ctx, g1[m]);
            if (g1[m]) { greater1_ctx = 0; if (first_g1 < 0) first_g1 = m; }
            else if (greater1_ctx > 0 && greater1_ctx < 3) greater1_ctx++;
        }
        int g2 = 0;
        if (first_g1 >= 0) { g2 = absv[first_g1] > 2; enc_bin(c, C_GT2 + (c_idx ? 4 : 0) + ctx_set, g2); }
        /* signs; with sign data hiding the last one coded (the sub-block's first position) is inferred */
        const int hidden = sdh && sig[0] - sig[n_sig - 1] >= 4;
        for (int m = 0; m < n_sig - hidden; m++) {
            int px, py;
            scan_pos(scan, 2, sig[m], &px, &py);
            enc_bypass(c, lev[(sy * 4 + py) * n + sx * 4 + px] < 0);
        }
        /* remaining levels */
        const int persist = w->p->persistent_rice != 0, sb_type = 2 * (c_idx == 0) + (tskip || w->cu_bypass);
        int rice = persist ? w->stat_coeff[sb_type] / 4 : 0, rice_init = 0;
        for (int m = 0; m < n_sig; m++) {
            const int base = m < 8 ? (m == first_g1 ? 3 : 2) : 1;
            const int have = m < 8 ? 1 + g1[m] + (m == first_g1 ? g2 : 0) : 1;
            if (have == base) {
                const int rem = absv[m] - base;
                code_remaining(c, rem, rice);
                if (absv[m] > 3 * (1 << rice)) rice = persist ? rice + 1 : rice < 4 ? rice + 1 : 4;
                if (persist && !rice_init) {               /* the sub-block's first escape value moves the statistic (9.3.3.11) */
                    const int r0 = w->stat_coeff[sb_type] / 4;
                    if (rem >= (3 << r0)) w->stat_coeff[sb_type]++;
                    else if (2 * rem < (1 << r0) && w->stat_coeff[sb_type] > 0) w->stat_coeff[sb_type]--;
                    rice_init = 1;
                }
            }
        }
    }
}

/* ---- transform tree ---- */
typedef struct Cu { int x, y, log2, intra, part, bypass, merge_2Nx2N; int ipm[4], ipm_c[4], cm_c[4]; } Cu;   /* ipm_c / cm_c: chroma mode and intra_chroma_pred_mode per partition (one unless 4:4:4 NxN) */

static int g_c444;                                    /* ChromaArrayType == 3 in the stream being written */
static int g_c422;                                    /* ChromaArrayType == 2: two square chroma blocks per transform unit, one above the other */
static int scan_of(int mode, int log2, int c_idx)
{
    if (!(log2 == 2 || (log2 == 3 && (c_idx == 0 || g_c444))))
        return 0;
    if (mode >= 6 && mode <= 14) return 2;
    if (mode >= 22 && mode <= 30) return 1;
    return 0;
}

static void code_tu(W *w, const Cu *cu, int x, int y, int log2, int depth, int blk, int cbf_cb, int cbf_cr, int cbf_luma)
{
    Cabac *c = &w->c;
    const int chroma_here = log2 > 2, chroma_parent = log2 == 2 && blk == 3;
    if (cbf_luma || cbf_cb || cbf_cr) {                    /* for 4x4 luma blocks the chroma flags are those of the 8x8 parent (7.3.8.10) */
        if (w->p->cu_qp_delta && w->qp_delta_pending) {
            /* mostly none or small, now and then anything the range allows (7.4.9.10: -(26 + QpBdOffset / 2) .. 25 + QpBdOffset / 2) */
            const int lim = 25 + 3 * (w->p->bit_depth - 8);
            const int d = rnd(&w->g, 3) ? 0 : rnd(&w->g, 6) ? rnd(&w->g, 7) - 3 : rnd(&w->g, 2 * lim + 1) - lim;
            const int a = abs(d);
            enc_bin(c, C_QP_DELTA, a > 0);
            for (int i = 1; i < 5 && i <= a; i++) enc_bin(c, C_QP_DELTA + 1, i < a);
            if (a >= 5) {                                  /* suffix: 0-th order Exp-Golomb of a - 5, bypass bins (9.3.3.10) */
                int v = a - 5, k = 0;
                while (v >= (1 << k)) { enc_bypass(c, 1); v -= 1 << k; k++; }
                enc_bypass(c, 0);
                enc_bypass_bits(c, (uint32_t)v, k);
            }
            tr(OH_SE_QP_DELTA_ABS, a);
            if (a) { enc_bypass(c, d < 0); tr(OH_SE_QP_DELTA_SIGN, d < 0); }
            w->qp_delta_pending = 0;
        }
    }
    const int pu = cu->part == PART_NxN && cu->intra ? w->tt_pu : 0;       /* the mode of the partition the block lies in (hevc.c:1461-1474: set at depth 1) */
    const int pc = g_c444 ? pu : 0;                        /* which chroma mode applies */
    if (cbf_luma)
        code_residual(w, log2, 0, cu->intra ? scan_of(cu->ipm[pu], log2, 0) : 0, cu->intra, cu->intra ? cu->ipm[pu] : -1);
    if (g_c444) {
        /* 4:4:4: chroma blocks have the luma block's size, 4x4 included; cross-component prediction (7.3.8.12) in front of each */
        const int cross = w->p->cross_component_pred && cbf_luma && (!cu->intra || cu->cm_c[pc] == 4);
        for (int ci = 1; ci <= 2; ci++) {
            int scale = 0;
            if (cross) {
                const int a = rnd(&w->g, 5), neg = rnd(&w->g, 2);        /* log2_res_scale_abs_plus1: truncated unary, 4 contexts per component */
                for (int i = 0; i < 4 && i <= a; i++) { if (i < a) enc_bin(c, C_RES_SCALE + 4 * (ci - 1) + i, 1); else enc_bin(c, C_RES_SCALE + 4 * (ci - 1) + i, 0); }
                tr(OH_SE_RES_SCALE_ABS, a);
                if (a) { enc_bin(c, C_RES_SIGN + ci - 1, neg); tr(OH_SE_RES_SCALE_SIGN, neg); scale = (1 << (a - 1)) * (1 - 2 * neg); }
            }
            w->cross_pf = cross; w->res_scale = scale;
            const int cbf = ci == 1 ? cbf_cb : cbf_cr;
            if (cbf) code_residual(w, log2, ci, cu->intra ? scan_of(cu->ipm_c[pc], log2, ci) : 0, cu->intra, cu->intra ? cu->ipm_c[pc] : -1);
            else if (cross) log_cross_only(w, log2, ci);
            w->cross_pf = 0; w->res_scale = 0;
        }
    } else if (chroma_here || chroma_parent) {
        /* 4:2:0 / 4:2:2: chroma blocks of half the luma width (of the 8x8 parent for 4x4 luma blocks); 4:2:2 has two per component,
         * the second below the first (7.3.8.10: all of Cb, then all of Cr; hevc.c:1302-1391) */
        const int lc2 = chroma_here ? log2 - 1 : 2;
        for (int ci = 1; ci <= 2; ci++)
            for (int i = 0; i < (g_c422 ? 2 : 1); i++)
                if (((ci == 1 ? cbf_cb : cbf_cr) >> i) & 1)
                    code_residual(w, lc2, ci, cu->intra ? scan_of(cu->ipm_c[0], lc2, ci) : 0, cu->intra, cu->intra ? cu->ipm_c[0] : -1);
    }
    (void)x; (void)y; (void)depth;
}

static void code_tt(W *w, const Cu *cu, int x, int y, int log2, int depth, int blk, int pcb, int pcr, int max_depth)
{
    Cabac *c = &w->c;
    const OhStreamParams *p = w->p;
    const int intra_split = cu->intra && cu->part == PART_NxN;
    if (depth <= 1) w->tt_pu = depth ? blk : 0;
    const int inter_split = p->max_th_depth_inter == 0 && !cu->intra && cu->part != PART_2Nx2N && depth == 0;
    int split;
    if (log2 <= p->log2_max_tb_size && log2 > p->log2_min_tb_size && depth < max_depth && !(intra_split && depth == 0)) {
        split = pct(&w->g, p->split_pct);
        enc_bin(c, C_SPLIT_TU + 5 - log2, split); tr(OH_SE_SPLIT_TU, split);
    } else {
        split = log2 > p->log2_max_tb_size || (intra_split && depth == 0) || inter_split;
    }
    int cb = 0, cr = 0;                                    /* bit 0: the (first) chroma block, bit 1: 4:2:2's second block below it */
    if (log2 > 2 || g_c444) {
        const int two = g_c422 && (!split || log2 == 3);   /* 7.3.8.8: the second flag where the chroma blocks are coded at this level */
        for (int comp = 0; comp < 2; comp++) {
            int *f = comp ? &cr : &cb;
            if (!(depth == 0 || ((comp ? pcr : pcb) & 1)))
                continue;
            for (int i = 0; i < (two ? 2 : 1); i++) {
                const int v = pct(&w->g, p->cbf_pct / 2 + 5);
                enc_bin(c, C_CBF_CHROMA + depth, v); tr(OH_SE_CBF_CHROMA, v);
                *f |= v << i;
            }
        }
    } else {
        cb = pcb; cr = pcr;                                /* 4x4 luma blocks: the chroma flags of the 8x8 parent apply */
    }
    if (split) {
        const int h = 1 << (log2 - 1);
        code_tt(w, cu, x, y, log2 - 1, depth + 1, 0, cb, cr, max_depth);
        code_tt(w, cu, x + h, y, log2 - 1, depth + 1, 1, cb, cr, max_depth);
        code_tt(w, cu, x, y + h, log2 - 1, depth + 1, 2, cb, cr, max_depth);
        code_tt(w, cu, x + h, y + h, log2 - 1, depth + 1, 3, cb, cr, max_depth);
    } else {
        int luma = 1;
        if (cu->intra || depth != 0 || cb || cr) { luma = pct(&w->g, p->cbf_pct); enc_bin(c, C_CBF_LUMA + (depth == 0 ? 1 : 0), luma); tr(OH_SE_CBF_LUMA, luma); }
        code_tu(w, cu, x, y, log2, depth, blk, cb, cr, luma);
    }
}

/* ---- prediction unit ---- */
static void code_mvd(W *w)
{
    Cabac *c = &w->c;
    int v[2], a[2];
    for (int k = 0; k < 2; k++) {
        int r = w->p->mvd_range;
        v[k] = rnd(&w->g, 2 * r + 1) - r;
        if (rnd(&w->g, 50) == 0) v[k] = rnd(&w->g, 65536) - 32768;
        if (rnd(&w->g, 4) == 0) v[k] = 0;
        if (w->el) v[k] = 0;                               /* inter-layer prediction: the motion vector into the up-sampled base-layer picture is zero (H.8.x; the
                                                              reference up-samples the CTBs under the prediction block, hevc.c:2077-2097) */
        a[k] = abs(v[k]);
    }
    enc_bin(c, C_MVD_GT0, a[0] > 0); enc_bin(c, C_MVD_GT0, a[1] > 0);
    if (a[0]) enc_bin(c, C_MVD_GT1, a[0] > 1);
    if (a[1]) enc_bin(c, C_MVD_GT1, a[1] > 1);
    for (int k = 0; k < 2; k++) {
        if (!a[k]) continue;
        if (a[k] > 1) {                                    /* abs_mvd_minus2: exp-Golomb of order 1 */
            int val = a[k] - 2, kk = 1;
            while (val >= (1 << kk)) { enc_bypass(c, 1); val -= 1 << kk; kk++; }
            enc_bypass(c, 0);
            enc_bypass_bits(c, (uint32_t)val, kk);
        }
        enc_bypass(c, v[k] < 0);
    }
    tr(OH_SE_MVD_X, v[0]); tr(OH_SE_MVD_Y, v[1]);
}

static void code_pu(W *w, const Cu *cu, int pw, int ph, int skip, int *merge_out)
{
    Cabac *c = &w->c;
    const Slice *sl = &w->sl;
    int merge = 1;
    if (!skip) { merge = pct(&w->g, w->p->merge_pct); enc_bin(c, C_MERGE_FLAG, merge); tr(OH_SE_MERGE_FLAG, merge); }
    *merge_out = merge;
    if (merge) {
        if (sl->max_merge > 1) {
            const int idx = rnd(&w->g, sl->max_merge);
            for (int i = 0; i < sl->max_merge - 1; i++) {
                const int b = i < idx;
                if (i == 0) enc_bin(c, C_MERGE_IDX, b); else enc_bypass(c, b);
                if (!b) break;
            }
            tr(OH_SE_MERGE_IDX, idx);
        }
        return;
    }
    int dir = 0;                                           /* 0 L0, 1 L1, 2 BI */
    if (sl->type == SLICE_B) {
        const int depth = *cell(w->pic.depth, w, cu->x, cu->y);
        dir = pw + ph != 12 && pct(&w->g, w->p->bi_pct) ? 2 : rnd(&w->g, 2);
        if (pw + ph != 12) enc_bin(c, C_INTER_DIR + depth, dir == 2);
        if (dir != 2) enc_bin(c, C_INTER_DIR + 4, dir);
        tr(OH_SE_INTER_DIR, dir);
    }
    for (int l = 0; l < 2; l++) {
        if ((l == 0 && dir == 1) || (l == 1 && dir == 0))
            continue;
        if (sl->n_ref[l] > 1) {
            const int idx = rnd(&w->g, sl->n_ref[l]), cmax = sl->n_ref[l] - 1;
            for (int i = 0; i < cmax; i++) {
                const int b = i < idx;
                if (i < 2) enc_bin(c, C_REF_IDX + i, b); else enc_bypass(c, b);
                if (!b) break;
            }
            tr(OH_SE_REF_IDX, idx);
        } else {
            tr(OH_SE_REF_IDX, 0);                          /* nothing is coded; the reference still passes through its ref_idx function */
        }
        if (!(l == 1 && sl->mvd_l1_zero && dir == 2))
            code_mvd(w);
        { const int mvp = rnd(&w->g, 2); enc_bin(c, C_MVP, mvp); tr(OH_SE_MVP, mvp); }
    }
}

/* ---- intra mode derivation (8.4.2): the writer needs IntraPredModeY to pick the coefficient scan ---- */
static int derive_ipm(W *w, int x, int y, int prev_flag, int mpm_idx, int rem)
{
    int cand[3], a = 1, b = 1;                             /* DC when the neighbour is missing or not intra */
    if (avail(w, x, y, x - 1, y) && *cell(w->pic.intra, w, x - 1, y) && !*cell(w->pic.pcm, w, x - 1, y)) a = *cell(w->pic.ipm, w, x - 1, y);
    if (avail(w, x, y, x, y - 1) && *cell(w->pic.intra, w, x, y - 1) && !*cell(w->pic.pcm, w, x, y - 1) && ((y - 1) >> w->lc) == (y >> w->lc))
        b = *cell(w->pic.ipm, w, x, y - 1);
    if (a == b) {
        if (a < 2) { cand[0] = 0; cand[1] = 1; cand[2] = 26; }
        else { cand[0] = a; cand[1] = 2 + ((a + 29) % 32); cand[2] = 2 + ((a - 2 + 1) % 32); }
    } else {
        cand[0] = a; cand[1] = b;
        cand[2] = (a != 0 && b != 0) ? 0 : ((a != 1 && b != 1) ? 1 : 26);
    }
    if (prev_flag)
        return cand[mpm_idx];
    if (cand[0] > cand[1]) { int t = cand[0]; cand[0] = cand[1]; cand[1] = t; }
    if (cand[0] > cand[2]) { int t = cand[0]; cand[0] = cand[2]; cand[2] = t; }
    if (cand[1] > cand[2]) { int t = cand[1]; cand[1] = cand[2]; cand[2] = t; }
    int m = rem;
    for (int i = 0; i < 3; i++) if (m >= cand[i]) m++;
    return m;
}

/* ---- coding unit ---- */
static void code_cu(W *w, int x, int y, int log2)
{
    Cabac *c = &w->c;
    const OhStreamParams *p = w->p;
    const Slice *sl = &w->sl;
    const int n = 1 << log2;
    Cu cu;
    memset(&cu, 0, sizeof(cu));
    cu.x = x; cu.y = y; cu.log2 = log2;
    if (p->transquant_bypass) { cu.bypass = pct(&w->g, p->bypass_pct); enc_bin(c, C_BYPASS_FLAG, cu.bypass); tr(OH_SE_BYPASS_FLAG, cu.bypass); }
    w->cu_bypass = cu.bypass;
    int skip = 0;
    if (sl->type != SLICE_I) {
        const int l = avail(w, x, y, x - 1, y) && *cell(w->pic.skip, w, x - 1, y), u = avail(w, x, y, x, y - 1) && *cell(w->pic.skip, w, x, y - 1);
        skip = pct(&w->g, p->skip_pct);
        enc_bin(c, C_SKIP + l + u, skip); tr(OH_SE_SKIP, skip);
    }
    fill(w->pic.skip, w, x, y, n, skip);
    fill(w->pic.intra, w, x, y, n, 0);
    fill(w->pic.pcm, w, x, y, n, 0);
    if (skip) {
        int m;
        code_pu(w, &cu, n, n, 1, &m);
        return;
    }
    cu.intra = 1;
    if (sl->type != SLICE_I) { cu.intra = pct(&w->g, p->intra_pct); enc_bin(c, C_PRED_MODE, cu.intra); tr(OH_SE_PRED_MODE, cu.intra); }
    cu.part = PART_2Nx2N;
    const int mcb = w->min_cb_log2;
    if (!cu.intra || log2 == mcb) {
        if (cu.intra) {
            cu.part = log2 > p->log2_min_tb_size && pct(&w->g, 40) ? PART_NxN : PART_2Nx2N;
            enc_bin(c, C_PART_MODE, cu.part == PART_2Nx2N); tr(OH_SE_PART_MODE, cu.part);
        } else {
            const int r = rnd(&w->g, 10);
            if (r < 4) cu.part = PART_2Nx2N;
            else if (log2 == 3) cu.part = r < 7 ? PART_2NxN : PART_Nx2N;                    /* an 8x8 coding block: no NxN, no AMP */
            else if (log2 == mcb) cu.part = r < 6 ? PART_2NxN : r < 8 ? PART_Nx2N : PART_NxN;  /* the smallest coding block, larger than 8x8: NxN instead of AMP */
            else if (!p->amp) cu.part = r < 7 ? PART_2NxN : PART_Nx2N;
            else cu.part = r == 4 ? PART_2NxN : r == 5 ? PART_Nx2N : r == 6 ? PART_2NxnU : r == 7 ? PART_2NxnD : r == 8 ? PART_nLx2N : PART_nRx2N;
            enc_bin(c, C_PART_MODE, cu.part == PART_2Nx2N);
            if (cu.part != PART_2Nx2N) {
                const int horiz = cu.part == PART_2NxN || cu.part == PART_2NxnU || cu.part == PART_2NxnD;
                enc_bin(c, C_PART_MODE + 1, horiz);
                if (log2 == mcb) {
                    if (log2 > 3 && !horiz) enc_bin(c, C_PART_MODE + 2, cu.part == PART_Nx2N);      /* ff_hevc_part_mode_decode, hevc_cabac.c: 001 Nx2N, 000 NxN */
                } else if (p->amp) {
                    const int sym = cu.part == PART_2NxN || cu.part == PART_Nx2N;
                    enc_bin(c, C_PART_MODE + 3, sym);
                    if (!sym) enc_bypass(c, cu.part == PART_2NxnD || cu.part == PART_nRx2N);
                }
            }
            tr(OH_SE_PART_MODE, cu.part);
        }
    }
    if (cu.intra) {
        fill(w->pic.intra, w, x, y, n, 1);
        int pcm = 0;
        if (cu.part == PART_2Nx2N && p->pcm && log2 >= mcb && log2 <= (p->log2_ctb_size < 5 ? p->log2_ctb_size : 5)) {
            pcm = pct(&w->g, p->pcm_pct);
            enc_terminate(c, pcm); tr(OH_SE_PCM_FLAG, pcm);
        }
        if (pcm) {
            fill(w->pic.pcm, w, x, y, n, 1);
            fill(w->pic.ipm, w, x, y, n, 1);
            byte_align_zero(c->out);                       /* pcm_alignment_zero_bit */
            for (int i = 0; i < n * n; i++) put_bits(c->out, (uint32_t)rnd(&w->g, 1 << p->bit_depth), p->bit_depth);
            const int n_chroma = 2 * (n >> (p->chroma_format_idc != 3)) * (n >> (p->chroma_format_idc == 1));     /* both chroma blocks (7.3.8.7) */
            for (int i = 0; i < n_chroma; i++) put_bits(c->out, (uint32_t)rnd(&w->g, 1 << p->bit_depth), p->bit_depth);
            cabac_start(c, c->out);                        /* 9.3.2.5: the arithmetic engine starts over, the contexts stay */
            return;
        }
        const int np = cu.part == PART_NxN ? 4 : 1, h = n >> 1;
        int prev[4], mpm[4], rem[4];
        for (int k = 0; k < np; k++) { prev[k] = rnd(&w->g, 2); enc_bin(c, C_PREV_INTRA, prev[k]); tr(OH_SE_PREV_INTRA, prev[k]); }
        for (int k = 0; k < np; k++) {
            const int px = x + (k & 1) * h * (np == 4), py = y + (k >> 1) * h * (np == 4);
            mpm[k] = rnd(&w->g, 3); rem[k] = rnd(&w->g, 32);
            if (prev[k]) { enc_bypass(c, mpm[k] > 0); if (mpm[k] > 0) enc_bypass(c, mpm[k] > 1); tr(OH_SE_MPM_IDX, mpm[k]); }
            else { enc_bypass_bits(c, (uint32_t)rem[k], 5); tr(OH_SE_REM_INTRA, rem[k]); }
            cu.ipm[k] = derive_ipm(w, px, py, prev[k], mpm[k], rem[k]);
            fill(w->pic.ipm, w, px, py, np == 4 ? h : n, cu.ipm[k]);
        }
        for (int k = 0; k < (g_c444 ? np : 1); k++) {      /* 4:4:4: one intra_chroma_pred_mode per partition (7.3.8.5) */
            const int cm = rnd(&w->g, 5);                  /* 4 = derived from luma */
            enc_bin(c, C_CHROMA_MODE, cm != 4);
            if (cm != 4) enc_bypass_bits(c, (uint32_t)cm, 2);
            tr(OH_SE_CHROMA_MODE, cm);
            static const uint8_t tab[4] = { 0, 26, 10, 1 };
            cu.cm_c[k] = cm;
            cu.ipm_c[k] = cm == 4 ? cu.ipm[k] : (tab[cm] == cu.ipm[k] ? 34 : tab[cm]);
            if (g_c422) {                                  /* the 4:2:2 mode mapping (table 8-3; hevc.c:2252-2254, 2299-2310) */
                static const uint8_t tab422[35] = { 0, 1, 2, 2, 2, 2, 3, 5, 7, 8, 10, 12, 13, 15, 17, 18, 19, 20,
                                                    21, 22, 23, 23, 24, 24, 25, 25, 26, 27, 27, 28, 28, 29, 29, 30, 31 };
                cu.ipm_c[k] = tab422[cu.ipm_c[k]];
            }
        }
    } else {
        fill(w->pic.ipm, w, x, y, n, 1);
        const int q = n >> 2, h = n >> 1;
        int m0 = 0, m;
        switch (cu.part) {
        case PART_2Nx2N: code_pu(w, &cu, n, n, 0, &m0); break;
        case PART_2NxN:  code_pu(w, &cu, n, h, 0, &m); code_pu(w, &cu, n, h, 0, &m); break;
        case PART_Nx2N:  code_pu(w, &cu, h, n, 0, &m); code_pu(w, &cu, h, n, 0, &m); break;
        case PART_NxN:   for (int k = 0; k < 4; k++) code_pu(w, &cu, h, h, 0, &m); break;
        case PART_2NxnU: code_pu(w, &cu, n, q, 0, &m); code_pu(w, &cu, n, n - q, 0, &m); break;
        case PART_2NxnD: code_pu(w, &cu, n, n - q, 0, &m); code_pu(w, &cu, n, q, 0, &m); break;
        case PART_nLx2N: code_pu(w, &cu, q, n, 0, &m); code_pu(w, &cu, n - q, n, 0, &m); break;
        default:         code_pu(w, &cu, n - q, n, 0, &m); code_pu(w, &cu, q, n, 0, &m); break;
        }
        cu.merge_2Nx2N = cu.part == PART_2Nx2N && m0;
    }
    int root = 1;
    if (!cu.intra && !cu.merge_2Nx2N) { root = pct(&w->g, p->cbf_pct + 20); enc_bin(c, C_ROOT_CBF, root); tr(OH_SE_ROOT_CBF, root); }
    if (root) {
        const int max_depth = cu.intra ? p->max_th_depth_intra + (cu.part == PART_NxN) : p->max_th_depth_inter;
        code_tt(w, &cu, x, y, log2, 0, 0, 0, 0, max_depth);
    }
}

/* ---- coding quadtree; returns 1 when the slice ended with this CTB's last CU ---- */
static void code_cqt(W *w, int x, int y, int log2, int depth)
{
    Cabac *c = &w->c;
    const OhStreamParams *p = w->p;
    const int n = 1 << log2;
    int split;
    if (x + n <= p->width && y + n <= p->height && log2 > w->min_cb_log2) {
        const int l = avail(w, x, y, x - 1, y) && *cell(w->pic.depth, w, x - 1, y) > depth;
        const int u = avail(w, x, y, x, y - 1) && *cell(w->pic.depth, w, x, y - 1) > depth;
        split = pct(&w->g, p->split_pct);
        enc_bin(c, C_SPLIT_CU + l + u, split); tr(OH_SE_SPLIT_CU, split);
    } else {
        split = log2 > w->min_cb_log2;
    }
    if (p->cu_qp_delta && log2 >= p->log2_ctb_size - (p->log2_ctb_size > w->min_cb_log2))
        w->qp_delta_pending = 1;                           /* a new quantisation group */
    if (split) {
        const int h = n >> 1;
        for (int k = 0; k < 4; k++) {
            const int xx = x + (k & 1) * h, yy = y + (k >> 1) * h;
            if (xx < p->width && yy < p->height)
                code_cqt(w, xx, yy, log2 - 1, depth + 1);
        }
    } else {
        fill(w->pic.depth, w, x, y, n, depth);
        code_cu(w, x, y, log2);
    }
}

/* ================================================================================================= pictures */
typedef struct Dpb { int poc[8]; int n; } Dpb;

static void write_slice_header(W *w, Bits *b, int nal_type, int first, int poc, const Dpb *dpb, int n_entry, const uint32_t *entry, int idr)
{
    const OhStreamParams *p = w->p;
    const Slice *sl = &w->sl;
    put_bit(b, first);
    if (nal_type >= 16 && nal_type <= 23) put_bit(b, 0);   /* no_output_of_prior_pics_flag */
    put_ue(b, (uint32_t)w->el);                            /* pps id */
    if (!first) {
        if (p->dependent_slices) put_bit(b, sl->dependent);
        int len = 0;
        while ((1 << len) < w->n_ctb) len++;
        put_bits(b, (uint32_t)sl->addr, len);
    }
    if (!sl->dependent) {
        put_ue(b, (uint32_t)sl->type);
        if (idr && w->el)
            put_bits(b, (uint32_t)poc & 255, 8);           /* a picture of layer 1 carries pic_order_cnt_lsb even when it is an IDR picture (hevc.c:728) */
        if (!idr) {
            put_bits(b, (uint32_t)poc & 255, 8);
            put_bit(b, 0);                                 /* short_term_ref_pic_set_sps_flag: the set follows */
            /* the pictures kept, all of them usable by this picture: earlier ones nearest first, then later ones nearest first (7.3.7) */
            int neg[8], pos[8], nn = 0, np_ = 0;
            for (int i = 0; i < dpb->n; i++) { if (dpb->poc[i] < poc) neg[nn++] = dpb->poc[i]; else pos[np_++] = dpb->poc[i]; }
            for (int i = 1; i < nn; i++) for (int j = i; j > 0 && neg[j] > neg[j - 1]; j--) { const int t = neg[j]; neg[j] = neg[j - 1]; neg[j - 1] = t; }
            for (int i = 1; i < np_; i++) for (int j = i; j > 0 && pos[j] < pos[j - 1]; j--) { const int t = pos[j]; pos[j] = pos[j - 1]; pos[j - 1] = t; }
            put_ue(b, (uint32_t)nn); put_ue(b, (uint32_t)np_);
            int prev = poc;
            for (int i = 0; i < nn; i++) { put_ue(b, (uint32_t)(prev - neg[i] - 1)); put_bit(b, 1); prev = neg[i]; }
            prev = poc;
            for (int i = 0; i < np_; i++) { put_ue(b, (uint32_t)(pos[i] - prev - 1)); put_bit(b, 1); prev = pos[i]; }
            if (p->tmvp) put_bit(b, sl->tmvp);
        }
        if (w->el) put_bit(b, 1);                          /* inter_layer_pred_enabled_flag: the one direct reference layer is active (hevc.c:806-828) */
        if (p->sao) { put_bit(b, sl->sao_luma); put_bit(b, sl->sao_chroma); }
        if (sl->type != SLICE_I) {
            put_bit(b, 1);                                 /* num_ref_idx_active_override_flag */
            put_ue(b, (uint32_t)sl->n_ref[0] - 1);
            if (sl->type == SLICE_B) put_ue(b, (uint32_t)sl->n_ref[1] - 1);
            if (sl->type == SLICE_B) put_bit(b, sl->mvd_l1_zero);
            if (p->cabac_init_present) put_bit(b, sl->cabac_init_flag);
            if (sl->tmvp) {
                if (sl->type == SLICE_B) put_bit(b, 1);    /* collocated_from_l0 */
                if (sl->n_ref[0] > 1) put_ue(b, 0);
            }
            if (p->weighted_pred) {
                /* pred_weight_table (7.3.6.3): random but legal weights */
                const int denom = rnd(&w->g, 8);
                put_ue(b, (uint32_t)denom);
                const int dc = rnd(&w->g, 3) - 1;
                const int cd = denom + dc < 0 ? 0 : (denom + dc > 7 ? 7 : denom + dc);
                put_se(b, cd - denom);
                for (int l = 0; l < (sl->type == SLICE_B ? 2 : 1); l++) {
                    int lf[16], cf[16];
                    for (int i = 0; i < sl->n_ref[l]; i++) { lf[i] = rnd(&w->g, 2); put_bit(b, lf[i]); }
                    for (int i = 0; i < sl->n_ref[l]; i++) { cf[i] = rnd(&w->g, 2); put_bit(b, cf[i]); }
                    for (int i = 0; i < sl->n_ref[l]; i++) {
                        if (lf[i]) { put_se(b, rnd(&w->g, 41) - 20); put_se(b, rnd(&w->g, 41) - 20); }
                        if (cf[i]) for (int j = 0; j < 2; j++) { put_se(b, rnd(&w->g, 41) - 20); put_se(b, rnd(&w->g, 201) - 100); }
                    }
                }
            }
            put_ue(b, (uint32_t)(5 - sl->max_merge));
        }
        put_se(b, sl->qp - 26);
        int deblock_off = 0;                               /* slice_deblocking_filter_disabled_flag: the PPS's (0) unless overridden */
        if (p->deblocking_override) {
            const int ov = rnd(&w->g, 2);
            put_bit(b, ov);
            if (ov) {
                deblock_off = sl->deblock_disabled;
                put_bit(b, deblock_off);
                if (!deblock_off) { put_se(b, rnd(&w->g, 7) - 3); put_se(b, rnd(&w->g, 7) - 3); }
            }
        }
        if (p->lf_across_slices && (sl->sao_luma || sl->sao_chroma || !deblock_off))
            put_bit(b, sl->lf_across);
    }
    if (w->tcols > 1 || w->trows > 1 || p->wpp) {
        put_ue(b, (uint32_t)n_entry);
        if (n_entry > 0) {
            uint32_t mx = 0;
            for (int i = 0; i < n_entry; i++) if (entry[i] - 1 > mx) mx = entry[i] - 1;
            int len = 1;
            while (len < 32 && (mx >> len)) len++;
            put_ue(b, (uint32_t)len - 1);
            for (int i = 0; i < n_entry; i++) put_bits(b, entry[i] - 1, len);
        }
    }
    rbsp_trailing(b);                                      /* byte_alignment(): a one, then zeros */
}

/* the CTBs of one slice segment in tile-scan order, CABAC-coded into `data`; entry points at tile / CTB-row starts */
static void write_slice_data(W *w, Bits *data, int ts_first, int ts_end, uint32_t *entry, int *n_entry)
{
    const OhStreamParams *p = w->p;
    Cabac *c = &w->c;
    const int init_type = w->sl.type == SLICE_I ? 0 : (w->sl.type == SLICE_P ? (w->sl.cabac_init_flag ? 2 : 1) : (w->sl.cabac_init_flag ? 1 : 2));
    uint8_t *wpp_ctx = w->wpp_ctx;
    size_t sub_start = 0;
    *n_entry = 0;
    cabac_start(c, data);
    {
        /* a dependent slice segment goes on with the states the segment before it ended with — unless it opens a tile (fresh
         * states) or, with wavefronts, a CTB row: there the reference reloads what it saved after the second CTB of the row above
         * (ff_hevc_cabac_init, hevc_cabac.c:606-628: load_states whatever the availability of that CTB) */
        const int rs0 = w->rs_of_ts[ts_first];
        const int opens_tile = ts_first > 0 && w->tile_of[rs0] != w->tile_of[w->rs_of_ts[ts_first - 1]];
        if (!w->sl.dependent || opens_tile) {
            cabac_init_contexts(c, init_type, w->sl.qp);
            memset(w->stat_coeff, 0, sizeof(w->stat_coeff));
        } else if (p->wpp && rs0 % w->ctbw == 0) {
            if (w->ctbw == 1) { cabac_init_contexts(c, init_type, w->sl.qp); memset(w->stat_coeff, 0, sizeof(w->stat_coeff)); }
            else if (w->have_wpp) memcpy(c->state, wpp_ctx, N_CTX);
        }
    }
    for (int ts = ts_first; ts < ts_end; ts++) {
        const int rs = w->rs_of_ts[ts], rx = rs % w->ctbw, ry = rs / w->ctbw;
        const int tile_start = ts > ts_first && w->tile_of[rs] != w->tile_of[w->rs_of_ts[ts - 1]];
        const int new_sub = ts > ts_first && (tile_start || (p->wpp && rx == 0));
        if (new_sub) {
            /* the previous substream was closed after its last CTB (below); this one starts byte aligned with a fresh engine */
            entry[(*n_entry)++] = (uint32_t)(data->n / 8 - sub_start);
            sub_start = data->n / 8;
            cabac_start(c, data);
            if (tile_start) { cabac_init_contexts(c, init_type, w->sl.qp); memset(w->stat_coeff, 0, sizeof(w->stat_coeff)); }
            else if (w->have_wpp && ry > 0 && rx + 1 < w->ctbw && avail(w, rx << w->lc, ry << w->lc, (rx + 1) << w->lc, (ry - 1) << w->lc))
                memcpy(c->state, wpp_ctx, N_CTX);          /* synchronisation: the states after the second CTB of the row above */
            else cabac_init_contexts(c, init_type, w->sl.qp);
        }
        if (p->cu_qp_delta) w->qp_delta_pending = 1;
        code_sao(w, rx, ry);
        code_cqt(w, rx << w->lc, ry << w->lc, w->lc, 0);
        if (p->wpp && (rx == 1 || (w->ctbw == 1 && rx == 0))) { memcpy(wpp_ctx, c->state, N_CTX); w->have_wpp = 1; }
        const int last = ts + 1 == ts_end;
        enc_terminate(c, last);                            /* end_of_slice_segment_flag */
        tr(OH_SE_END_OF_SLICE, last);
        if (!last) {
            const int next_rs = w->rs_of_ts[ts + 1];
            if (w->tile_of[next_rs] != w->tile_of[rs] || (p->wpp && next_rs % w->ctbw == 0)) {
                enc_terminate(c, 1);                       /* end_of_subset_one_bit */
                byte_align_zero(data);
            }
        }
    }
    byte_align_zero(data);                                 /* the flush wrote the stop bit */
}

static void write_picture(W *w, int idx, int poc, int type, Dpb *dpb, int idr, int nonref)
{
    const OhStreamParams *p = w->p;
    memset(w->pic.depth, 0xff, (size_t)w->pic.w4 * w->pic.h4);
    memset(w->pic.skip, 0, (size_t)w->pic.w4 * w->pic.h4);
    memset(w->pic.intra, 0, (size_t)w->pic.w4 * w->pic.h4);
    memset(w->pic.pcm, 0, (size_t)w->pic.w4 * w->pic.h4);
    memset(w->pic.ipm, 1, (size_t)w->pic.w4 * w->pic.h4);
    /* slice starts in tile scan: whole tiles when tiles are on, else random CTBs */
    int starts[64], ns = 1;
    starts[0] = 0;
    const int tiles = w->tcols > 1 || w->trows > 1;
    for (int k = 1; k < p->n_slices && ns < 63; k++) {
        int ts;
        if (tiles) {
            const int t = 1 + rnd(&w->g, w->tcols * w->trows - 1);
            ts = 0;
            while (ts < w->n_ctb && w->tile_of[w->rs_of_ts[ts]] != t) ts++;
        } else {
            ts = 1 + rnd(&w->g, w->n_ctb - 1);
            if (p->wpp) ts -= ts % w->ctbw;                /* with wavefronts, slice segments that start mid-row must end in their row: none does */
        }
        int dup = ts <= 0 || ts >= w->n_ctb;
        for (int i = 0; i < ns; i++) dup |= starts[i] == ts;
        if (!dup) starts[ns++] = ts;
    }
    for (int i = 1; i < ns; i++)
        for (int j = i; j > 0 && starts[j] < starts[j - 1]; j--) { int t = starts[j]; starts[j] = starts[j - 1]; starts[j - 1] = t; }
    starts[ns] = w->n_ctb;
    const int nal_type = idr ? 19 : nonref ? 0 : 1;        /* IDR_W_RADL / TRAIL_N / TRAIL_R */
    w->have_wpp = 0;
    int slice_addr = 0;                                    /* SliceAddrRs: the address of the slice's first (independent) segment */
    for (int s = 0; s < ns; s++) {
        Slice *sl = &w->sl;
        if (p->dependent_slices && s > 0 && rnd(&w->g, 2)) {
            /* a dependent slice segment: its header carries the address only, everything else is the slice's (7.3.6.1) */
            sl->dependent = 1;
            sl->addr = w->rs_of_ts[starts[s]];
            goto coded;
        }
        memset(sl, 0, sizeof(*sl));
        sl->type = type;
        sl->addr = slice_addr = w->rs_of_ts[starts[s]];
        sl->qp = p->qp + rnd(&w->g, 7) - 3;
        sl->n_ref[0] = type == SLICE_I ? 0 : 1 + rnd(&w->g, dpb->n);
        sl->n_ref[1] = type == SLICE_B ? 1 + rnd(&w->g, dpb->n) : 0;
        sl->max_merge = 1 + rnd(&w->g, 5);
        sl->cabac_init_flag = p->cabac_init_present && rnd(&w->g, 2);
        sl->tmvp = p->tmvp && !idr && type != SLICE_I && dpb->n > 0;
        sl->mvd_l1_zero = type == SLICE_B && rnd(&w->g, 4) == 0;
        sl->deblock_disabled = p->deblocking_override && rnd(&w->g, 3) == 0;
        sl->lf_across = rnd(&w->g, 2);
        sl->sao_luma = p->sao && rnd(&w->g, 4) != 0;
        sl->sao_chroma = p->sao && rnd(&w->g, 4) != 0;
    coded:
        for (int ts = starts[s]; ts < starts[s + 1]; ts++) w->slice_of[w->rs_of_ts[ts]] = slice_addr;
        Bits data = { 0 }, hdr = { 0 };
        uint32_t entry[4096];
        int n_entry = 0;
        write_slice_data(w, &data, starts[s], starts[s + 1], entry, &n_entry);
        if (n_entry) {
            /* entry_point_offset counts bytes of the NAL unit payload, emulation prevention bytes included (7.4.7.1): measure the
             * substreams as they will be escaped.  The header ends with its alignment byte, which is never zero. */
            const size_t nbytes = data.n / 8;
            size_t pos = 0, esc = 0, sub_start = 0, sub_end = entry[0];
            int k = 0, zeros = 0;
            for (; pos < nbytes; pos++) {
                if (k < n_entry && pos == sub_end) {
                    const uint32_t plain_next = k + 1 < n_entry ? entry[k + 1] : 0;
                    entry[k++] = (uint32_t)(esc - sub_start);
                    sub_start = esc; sub_end = pos + plain_next;
                }
                if (zeros >= 2 && data.buf[pos] <= 3) { esc++; zeros = 0; }
                esc++;
                zeros = data.buf[pos] == 0 ? zeros + 1 : 0;
            }
        }
        write_slice_header(w, &hdr, nal_type, s == 0, poc, dpb, n_entry, entry, idr);
        Bits nal = { 0 };
        for (size_t i = 0; i < hdr.n / 8; i++) put_bits(&nal, hdr.buf[i], 8);
        for (size_t i = 0; i < data.n / 8; i++) put_bits(&nal, data.buf[i], 8);
        emit_nal(&w->out, nal_type, nal.buf, nal.n / 8);
        free(nal.buf); free(data.buf); free(hdr.buf);
    }
    (void)idx;
}

/* ================================================================================================= entry points */
void oh_stream_defaults(OhStreamParams *p, int width, int height, uint64_t seed)
{
    memset(p, 0, sizeof(*p));
    p->seed = seed; p->width = width; p->height = height; p->bit_depth = 8; p->log2_ctb_size = 6;
    p->log2_min_tb_size = 2; p->log2_max_tb_size = 5; p->max_th_depth_intra = 2; p->max_th_depth_inter = 2;
    p->n_pictures = 4; p->gop = 2; p->n_refs = 2; p->idr_period = 0; p->qp = 30; p->chroma_format_idc = 1;
    p->amp = 1; p->sao = 1; p->strong_intra_smoothing = 1; p->tmvp = 0;
    p->n_slices = 1; p->tile_cols = 1; p->tile_rows = 1; p->lf_across_slices = 1; p->lf_across_tiles = 1;
    p->split_pct = 50; p->intra_pct = 20; p->skip_pct = 25; p->merge_pct = 30; p->bi_pct = 40; p->cbf_pct = 55;
    p->pcm_pct = 8; p->bypass_pct = 8; p->tskip_pct = 25; p->sao_pct = 50; p->mvd_range = 64; p->coeff_density = 100;
}

/* the writer's state for one layer (one W per layer of a two-layer stream) */
static int w_init(W *w, const OhStreamParams *p, int mcb_log2)
{
    memset(w, 0, sizeof(*w));
    w->p = p;
    g_trace_on = p->trace != 0; g_trace_n = 0;
    g_lev_on = p->levels != 0 && !p->cu_qp_delta; g_lev_n = 0;
    g_c444 = p->chroma_format_idc == 3;
    g_cb_off = p->chroma_qp_offsets ? p->cb_qp_offset : 1; g_cr_off = p->chroma_qp_offsets ? p->cr_qp_offset : -2;
    if (g_cb_off < -12 || g_cb_off > 12 || g_cr_off < -12 || g_cr_off > 12) return -1;
    g_c422 = p->chroma_format_idc == 2;
    w->g.s = p->seed * 0x2545F4914F6CDD1Dull + 77;
    w->lc = p->log2_ctb_size; w->ctb = 1 << w->lc; w->min_cb_log2 = mcb_log2;
    w->ctbw = (p->width + w->ctb - 1) >> w->lc; w->ctbh = (p->height + w->ctb - 1) >> w->lc; w->n_ctb = w->ctbw * w->ctbh;
    w->tcols = p->tile_cols > 1 ? (p->tile_cols < w->ctbw ? p->tile_cols : w->ctbw) : 1;
    w->trows = p->tile_rows > 1 ? (p->tile_rows < w->ctbh ? p->tile_rows : w->ctbh) : 1;
    if (p->wpp && (w->tcols > 1 || w->trows > 1))
        return -1;                                         /* both at once is legal in version 2 only; not produced */
    w->pic.w4 = (p->width + 3) >> 2; w->pic.h4 = (p->height + 3) >> 2;
    const size_t cells = (size_t)w->pic.w4 * w->pic.h4;
    w->pic.skip = (uint8_t *)malloc(cells); w->pic.depth = (uint8_t *)malloc(cells); w->pic.intra = (uint8_t *)malloc(cells);
    w->pic.ipm = (uint8_t *)malloc(cells); w->pic.pcm = (uint8_t *)malloc(cells);
    w->slice_of = (int *)calloc((size_t)w->n_ctb, sizeof(int)); w->tile_of = (int *)calloc((size_t)w->n_ctb, sizeof(int));
    w->rs_of_ts = (int *)calloc((size_t)w->n_ctb + 1, sizeof(int)); w->ts_of_rs = (int *)calloc((size_t)w->n_ctb + 1, sizeof(int));
    for (int i = 0; i <= w->tcols; i++) w->col_bd[i] = (i * w->ctbw) / w->tcols;
    for (int i = 0; i <= w->trows; i++) w->row_bd[i] = (i * w->ctbh) / w->trows;
    {
        int ts = 0;
        for (int tr = 0; tr < w->trows; tr++)
            for (int tc = 0; tc < w->tcols; tc++)
                for (int y = w->row_bd[tr]; y < w->row_bd[tr + 1]; y++)
                    for (int x = w->col_bd[tc]; x < w->col_bd[tc + 1]; x++) {
                        const int rs = y * w->ctbw + x;
                        w->tile_of[rs] = tr * w->tcols + tc; w->rs_of_ts[ts] = rs; w->ts_of_rs[rs] = ts; ts++;
                    }
    }
    return 0;
}
static void w_free(W *w)
{
    free(w->pic.skip); free(w->pic.depth); free(w->pic.intra); free(w->pic.ipm); free(w->pic.pcm);
    free(w->slice_of); free(w->tile_of); free(w->rs_of_ts); free(w->ts_of_rs);
}

int oh_stream_write(const OhStreamParams *p, OhStream *out)
{
    if (!p || !out || p->width < 8 || p->height < 8 || (p->width & 7) || (p->height & 7) || (p->bit_depth != 8 && p->bit_depth != 9 && p->bit_depth != 10 && p->bit_depth != 12) ||
        p->log2_ctb_size < 4 || p->log2_ctb_size > 6 || p->log2_min_tb_size < 2 || p->log2_min_tb_size > 4 || p->log2_min_tb_size > p->log2_max_tb_size || p->log2_max_tb_size > 5 || p->log2_max_tb_size > p->log2_ctb_size ||
        p->log2_max_tb_size < 3 || p->n_refs < 1 || p->n_refs > 4 || p->n_pictures < 1 || p->max_th_depth_intra < 0 || p->max_th_depth_intra > 3 ||
        p->max_th_depth_inter < 0 || p->max_th_depth_inter > 3)
        return -1;
    if (p->chroma_format_idc < 1 || p->chroma_format_idc > 3 || (p->cross_component_pred && p->chroma_format_idc != 3) ||
        (p->chroma_format_idc >= 2 && (p->conf_win_left || p->conf_win_right || p->conf_win_top || p->conf_win_bottom)))
        return -1;                                         /* 4:2:2 / 4:4:4: no window (the reference doubles the offsets) */
    const int mcb_log2 = p->log2_min_cb_size ? p->log2_min_cb_size : 3;
    if (mcb_log2 < 3 || mcb_log2 > 5 || mcb_log2 > p->log2_ctb_size || (p->width & ((1 << mcb_log2) - 1)) || (p->height & ((1 << mcb_log2) - 1)) ||
        (p->pcm && mcb_log2 > (p->log2_ctb_size < 5 ? p->log2_ctb_size : 5)) || p->log2_min_tb_size >= mcb_log2)
        return -1;                                         /* the picture is a whole number of smallest coding blocks; the smallest transform block is smaller than they are */
    if (p->conf_win_left < 0 || p->conf_win_right < 0 || p->conf_win_top < 0 || p->conf_win_bottom < 0 ||
        p->conf_win_left + p->conf_win_right >= p->width || p->conf_win_top + p->conf_win_bottom >= p->height)
        return -1;                                         /* the conformance window leaves a picture */
    if (p->max_th_depth_intra > p->log2_ctb_size - p->log2_min_tb_size || p->max_th_depth_inter > p->log2_ctb_size - p->log2_min_tb_size)
        return -1;                                         /* 7.4.3.2.1: the transform hierarchy cannot be deeper than CTB / smallest transform block */
    {
        const int smax = p->bit_depth > 10 ? p->bit_depth - 10 : 0;
        if (p->sao_offset_scale_luma < 0 || p->sao_offset_scale_luma > smax || p->sao_offset_scale_chroma < 0 || p->sao_offset_scale_chroma > smax)
            return -1;                                     /* 7.4.3.3.2 */
    }
    if (p->chroma_format_idc == 2 && p->log2_min_tb_size > 2)
        return -1;                                         /* 4:2:2's second chroma block of a transform unit sits HALF a min-TB down when that is 8x8 or more; the reference
                                                              derives intra availability in whole min-TBs (hevcpred_template.c:73-109), calls its up-right neighbour
                                                              available and predicts from samples that are not decoded yet: output that depends on stale memory */
    if ((p->log2_max_tskip_size && (p->log2_max_tskip_size < 2 || p->log2_max_tskip_size > 5)) || (p->persistent_rice && p->wpp))
        return -1;
    if (p->gop < 0 || p->gop > 3 || (p->gop == 3 && p->n_refs < 2))
        return -1;                                         /* the hierarchical GOP keeps three pictures beside the current one */
    W w;
    if (w_init(&w, p, mcb_log2) != 0)
        return -1;
    /* two layers (SHVC spatial scalability): an enhancement layer of shvc_el_width x shvc_el_height on top of this stream as its base
     * layer.  Every access unit carries a picture of each layer; the enhancement-layer pictures are P slices whose ONE reference is the
     * up-sampled base-layer picture of the same access unit (inter-layer prediction, zero motion vectors) plus intra blocks and residuals
     * — the path that reaches the reference's up-sampling slots (hevc.c:2077-2097 ff_upsample_block, hevcdsp_template.c:1834-2162). */
    const int two = p->shvc_el_width > 0 && p->shvc_el_height > 0;
    OhStreamParams pe = *p;
    W we;
    memset(&we, 0, sizeof(we));
    if (two) {
        /* refused beyond the obvious: (1) the range extensions — the reference parses the extension bits of a layer-1 PPS / SPS as something
         * else ("PPS extension flag is partially implemented"), the enhancement layer would be decoded out of step; (2) an enhancement layer
         * of a single CTB row or column — the reference's block up-sampler emulates ONE picture edge per block and direction
         * (videodsp_template.c:110-116, 141-151 return after the left / top edge), a block that touches both reads samples nobody wrote;
         * (3) ratios above 2 — the reference's CTB path and its whole-picture slot then produce different chroma rows at the picture's
         * bottom (the engine follows the whole-picture slot: tests/test_upsample_vs_ref.py), so there is no single reference output;
         * (4) x1.5 beyond 2048 enhancement-layer columns or rows — the x1.5 block slots position by exact thirds ((x << 1) / 3, x % 3:
         * hevcdsp_template.c:2073-2077), the whole-picture slot by the 16.16 fixed-point scale 43691, whose rounding reaches a sixteenth
         * of a sample at x = 2048: the two paths pick different filter phases from there on */
        const int ctb = 1 << (p->log2_ctb_size ? p->log2_ctb_size : 6);
        const int x1_5 = 2 * p->shvc_el_width == 3 * p->width || 2 * p->shvc_el_height == 3 * p->height;
        if (rext_profile(p) || (x1_5 && (p->shvc_el_width > 2048 || p->shvc_el_height > 2048)) || p->shvc_el_width < ctb + 16 || p->shvc_el_height < ctb + 16 || p->shvc_el_width > 2 * p->width || p->shvc_el_height > 2 * p->height ||
            p->bit_depth != 8 || p->chroma_format_idc != 1 || p->gop == 3 || p->shvc_el_width < p->width || p->shvc_el_height < p->height ||
            (p->shvc_el_width & ((1 << mcb_log2) - 1)) || (p->shvc_el_height & ((1 << mcb_log2) - 1)) || p->trace || p->levels ||
            p->conf_win_left || p->conf_win_right || p->conf_win_top || p->conf_win_bottom) {
            w_free(&w);
            return -1;                                     /* the reference's up-sampler is written for 8-bit 4:2:0 (hevc_filter.c:34); the writer keeps the window empty */
        }
        pe.width = p->shvc_el_width; pe.height = p->shvc_el_height;
        pe.gop = 1; pe.n_refs = 1; pe.tmvp = 0; pe.pcm = 0; pe.scaling_list = 0; pe.weighted_pred = 0; pe.n_slices = 1; pe.tile_cols = pe.tile_rows = 1;
        pe.dependent_slices = 0; pe.idr_period = p->idr_period; pe.shvc_el_width = pe.shvc_el_height = 0;
        pe.seed = p->seed ^ 0x5348564300000001ull;
        if (w_init(&we, &pe, mcb_log2) != 0) { w_free(&w); return -1; }
        we.el = 1;
        g_trace_on = 0; g_lev_on = 0;
    }
    out->au_offset = (size_t *)calloc((size_t)p->n_pictures + 1, sizeof(size_t));
    Dpb dpb = { { 0 }, 0 };
    int poc = 0, idr_at = 0;
    for (int i = 0; i < p->n_pictures; i++) {
        byte_align_zero(&w.out);
        out->au_offset[i] = w.out.n / 8;
        const int idr = i == 0 || (p->idr_period > 0 && i % p->idr_period == 0);
        if (idr && two) {
            write_vps_two_layers(&w, pe.width, pe.height); write_sps(&w); write_pps(&w);
            g_layer = 1; write_sps(&we); write_pps(&we); g_layer = 0;
            bits_drain(&w.out, &we.out);
            dpb.n = 0; poc = 0;
        } else
        if (idr) { write_vps(&w); write_sps(&w); write_pps(&w); dpb.n = 0; poc = 0; }
        if (idr) idr_at = i;
        const int type = idr || p->gop == 0 ? SLICE_I : (p->gop == 1 ? SLICE_P : SLICE_B);
        if (p->gop == 3) {
            /* hierarchical B, mini-GOPs of four behind the IDR picture, decode order anchor +4, +2, +1, +3 (output order differs from
             * decode order: two pictures of reordering).  Reference picture sets: the anchor keeps the anchor before it, +2 both
             * anchors, +1 and +3 (sub-layer non-reference pictures) those and +2 */
            int nonref = 0;
            Dpb rps = { { 0 }, 0 };
            if (!idr) {
                static const int ofs[4] = { 4, 2, 1, 3 };
                const int j = i - idr_at - 1, base = 4 * (j / 4), r = j % 4;
                poc = base + ofs[r];
                nonref = r >= 2;
                rps.poc[rps.n++] = base;
                if (r >= 1) rps.poc[rps.n++] = base + 4;
                if (r >= 2) rps.poc[rps.n++] = base + 2;
            }
            write_picture(&w, i, poc, type, &rps, idr, nonref);
            dpb = rps;
            if (!nonref) dpb.poc[dpb.n++] = poc;
            continue;
        }
        write_picture(&w, i, poc, type, &dpb, idr, 0);
        if (two) {                                         /* the enhancement layer's picture of the same access unit */
            Dpb none = { { 0 }, 0 };
            g_layer = 1;
            write_picture(&we, i, poc, SLICE_P, &none, idr, 0);
            g_layer = 0;
            byte_align_zero(&we.out);
            bits_drain(&w.out, &we.out);
        }
        /* every picture is a reference; the newest n_refs are kept */
        for (int k = dpb.n < p->n_refs ? dpb.n : p->n_refs - 1; k > 0; k--) dpb.poc[k] = dpb.poc[k - 1];
        dpb.poc[0] = poc;
        if (dpb.n < p->n_refs) dpb.n++;
        poc++;
    }
    byte_align_zero(&w.out);
    out->au_offset[p->n_pictures] = w.out.n / 8;
    out->data = w.out.buf; out->size = w.out.n / 8; out->n_pictures = p->n_pictures;
    w_free(&w);
    if (two) { w_free(&we); free(we.out.buf); }
    return 0;
}

int oh_stream_add_md5(const OhStream *in, const uint8_t *md5, OhStream *out)
{
    if (!in || !md5 || !out)
        return -1;
    Bits b = { 0 };
    out->au_offset = (size_t *)calloc((size_t)in->n_pictures + 1, sizeof(size_t));
    for (int i = 0; i < in->n_pictures; i++) {
        out->au_offset[i] = b.n / 8;
        for (size_t k = in->au_offset[i]; k < in->au_offset[i + 1]; k++) put_bits(&b, in->data[k], 8);
        uint8_t sei[2 + 1 + 48];
        sei[0] = 132; sei[1] = 49; sei[2] = 0;             /* payload type: decoded picture hash; size; hash_type MD5 */
        memcpy(sei + 3, md5 + (size_t)i * 48, 48);
        Bits r = { 0 };
        for (size_t k = 0; k < sizeof(sei); k++) put_bits(&r, sei[k], 8);
        rbsp_trailing(&r);
        emit_nal(&b, 40, r.buf, r.n / 8);                  /* SUFFIX_SEI_NUT */
        free(r.buf);
    }
    out->au_offset[in->n_pictures] = b.n / 8;
    out->data = b.buf; out->size = b.n / 8; out->n_pictures = in->n_pictures;
    return 0;
}

void oh_stream_free(OhStream *s)
{
    if (!s) return;
    free(s->data); free(s->au_offset);
    s->data = NULL; s->au_offset = NULL; s->size = 0;
}
