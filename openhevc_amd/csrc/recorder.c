/*
 * recorder.c — host-side work-item recorder and the intra CTU-wavefront scheduler.
 * Plain C, no GPU dependency (part of libohevc_host.so and libohevc_hip.so).
 *
 * The scheduler is the GPU counterpart of the reference's WPP progress counters
 * (pthread_slice.c:238-263, hevc.c:2782/2808: "row r may decode CTU k once row r-1 finished
 * k+2").  Here the dependency is taken from what the intra blocks actually read:
 *   - every intra block gets a SUB-LEVEL inside its CTU: 1 + max sub-level of the same-CTU
 *     intra blocks whose samples it gathers (hevcpred_template.c:164-183);
 *   - every CTU with intra blocks gets a LEVEL: 1 + max level of the neighbouring CTUs (left,
 *     up-left, up, up-right) whose intra samples its blocks gather.
 * Inter-predicted samples never create a dependency: they are complete after passes 1-2.
 * Pass 3 launches one kernel per CTU level, one workgroup per CTU; sub-levels are separated by
 * workgroup barriers inside the CU instead of kernel launches.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/ohevc_recorder.h"

/* The appending entry points may be called from the reference's slice / wavefront threads (pthread_slice.c: one CTU row or slice
 * each), sixteen of them at once.  Every recording thread appends to lists of its OWN (a shard, picked on the thread's first call
 * for the picture): no lock on the hot path — with one shared list and a mutex the front end ran SLOWER on 16 threads than on one
 * (96 against 42 ms per 4K picture on the GPU box's EPYC, profiles/r03_front_end.txt).  The indices the entry points hand back
 * (oh_rec_tu*, oh_rec_intra_idx, oh_rec_n_intra) count inside the calling thread's shard; the blocks of a coding unit all come
 * from the thread that decodes its CTU row, so the links between them (intra block -> its residual, chroma block -> the luma
 * residual it is predicted from, PU -> its weights) never cross shards.  oh_rec_finish() concatenates the shards and rebases the
 * links; a picture recorded by one thread is handed on as it lies. */
#define OH_MAX_SHARDS 64
typedef struct Shard {
    OhPu      *pu;      uint32_t n_pu, cap_pu;
    OhWeights *wp;      uint32_t n_wp, cap_wp;
    OhTu      *tu;      uint32_t n_tu, cap_tu;
    int16_t   *coeffs;  uint64_t n_coeff, cap_coeff;
    OhIntra   *intra;   uint32_t n_intra, cap_intra;     /* in recording order */
    uint32_t  *it_ctu;                          /* CTU raster index of intra[i] */
    uint16_t  *it_sub;                          /* sub-level (1-based) of intra[i] */
    uint32_t  *sparse, *tu_sparse, *tu_cross; uint64_t cap_sparse, cap_tu_sparse, cap_tu_cross; uint32_t n_sparse;
    int any_sparse, any_matrix, any_cross, oom;
} Shard;

struct OhRecorder {
    pthread_mutex_t mu;                         /* hands out shards; nothing else */
    OhFrame  f;
    Shard     sh[OH_MAX_SHARDS]; int n_sh;      /* shards in use for the picture in flight */
    unsigned  epoch;                            /* changes with every oh_rec_begin(): a thread's cached shard is for one picture */
    /* the concatenated lists (pictures recorded by several threads) */
    OhPu      *pu;      uint32_t cap_pu;
    OhWeights *wp;      uint32_t cap_wp;
    OhTu      *tu;      uint32_t cap_tu;
    int16_t   *coeffs;  uint64_t cap_coeff;
    OhIntra   *sorted;  uint32_t cap_sorted;
    /* schedule */
    int        n_ctb, ctbw, ctbh;
    uint8_t   *ctu_dep;                         /* per CTU: bit0 left, bit1 up-left, bit2 up, bit3 up-right */
    uint16_t  *ctu_nsub;                        /* per CTU: highest sub-level recorded */
    uint16_t  *ctu_level;                       /* per CTU: wavefront level (1-based, 0 = no intra blocks) */
    uint32_t  *ctu_entry;                       /* per CTU: index into ictu[] */
    OhIntraCtu *ictu;
    uint32_t  *sub_start; uint32_t cap_sub;
    uint32_t  *level_start;
    /* side arrays */
    uint8_t *vbs, *hbs, *is_pcm, *is_intra;
    uint32_t *sparse, *tu_sparse, *tu_cross; uint64_t cap_sparse, cap_tu_sparse, cap_tu_cross;
    const OhBsInputs *bs_in;                                  /* caller-owned maps for the GPU boundary-strength pass, or NULL */
    OhBsInputs own_bs; OhMvField *bs_mvf; uint8_t *bs_cbf, *bs_call, *bs_flags;     /* recorder-owned maps (oh_rec_bs_maps) */
    OhScalingList scaling;
    OhCtbMaps ctb_maps; int ctb_maps_on;                       /* slices / tiles (oh_rec_ctb_maps); off = one slice, one tile */
    uint8_t *sao_pending;                                     /* OhFrame.sao_pending of tiled pictures with 16x16 CTBs (oh_sao_pending_driver) */
    int8_t  *qp;
    OhDeblockCtb *deblock;
    OhSaoCtb *sao;
    /* per-plane sub-level map at 4x4-sample granularity, and the luma "decoded" map (4x4 luma) */
    uint16_t *lvl[3]; int lw[3], lh[3];
    uint8_t  *decoded; int dw, dh;
    int oom;                                     /* an allocation failed while recording this picture: items were dropped, oh_rec_finish() returns NULL */
    int clean;                                   /* the per-picture maps are already zero (oh_rec_recycle) */
};

/* the table slots that feed the recorder cannot fail (void returns), so an allocation failure is latched in r->oom: the block
 * is kept as it was (no write through NULL, the old block is not lost), the item is dropped, and oh_rec_finish() reports it */
static void *grow(int *oom, void *p, size_t elem, uint64_t *cap, uint64_t need)
{
    if (need <= *cap)
        return p;
    uint64_t n = *cap ? *cap : 1024;
    while (n < need)
        n *= 2;
    void *q = realloc(p, (size_t)(n * elem));
    if (!q) {
        *oom = 1;
        return p;
    }
    *cap = n;
    return q;
}
#define GROW32(own, ptr, cap, need) do { uint64_t c_ = (cap); (ptr) = grow(&(own)->oom, (ptr), sizeof(*(ptr)), &c_, (need)); (cap) = (uint32_t)c_; } while (0)

static unsigned g_epoch;                          /* picture epochs are unique across recorders (a recorder may be re-created at the same address) */
static __thread struct { const OhRecorder *r; unsigned epoch; Shard *s; } tls_shard;
/* the calling thread's lists for the picture in flight */
static Shard *my_shard(OhRecorder *r)
{
    if (tls_shard.r == r && tls_shard.epoch == r->epoch)
        return tls_shard.s;
    pthread_mutex_lock(&r->mu);
    Shard *s;
    if (r->n_sh < OH_MAX_SHARDS) {
        s = &r->sh[r->n_sh++];
    } else {                                              /* more recording threads than shards: the picture is refused (oh_rec_finish returns NULL) */
        s = &r->sh[OH_MAX_SHARDS - 1];
        r->oom = 1;
    }
    pthread_mutex_unlock(&r->mu);
    tls_shard.r = r; tls_shard.epoch = r->epoch; tls_shard.s = s;
    return s;
}

void oh_rec_destroy(OhRecorder *r);

OhRecorder *oh_rec_create(const OhPicParams *p)
{
    /* the intra schedule counts CTBs in 16 bits (OhIntraCtu.ctu): 8K with 16x16 CTBs (129 600) does not fit; the engine refuses such
     * pictures with OH_E_UNSUPPORTED, the recorder does not start on them */
    if (!p || (long long)oh_ctb_width(p) * oh_ctb_height(p) > 65535) {
        if (p) fprintf(stderr, "ohevc recorder: %dx%d with %dx%d CTBs has more than 65535 CTBs (unsupported: use larger CTBs)\n",
                       p->width, p->height, 1 << p->log2_ctb_size, 1 << p->log2_ctb_size);
        return NULL;
    }
    OhRecorder *r = (OhRecorder *)calloc(1, sizeof(*r));
    if (r) pthread_mutex_init(&r->mu, NULL);
    if (!r)
        return NULL;
    r->f.p = *p;
    int nplanes = p->chroma_format_idc ? 3 : 1;
    for (int c = 0; c < nplanes; c++) {
        int pw = p->width >> oh_hshift(p, c), ph = p->height >> oh_vshift(p, c);
        r->lw[c] = (pw + 3) >> 2;
        r->lh[c] = (ph + 3) >> 2;
        r->lvl[c] = (uint16_t *)calloc((size_t)r->lw[c] * r->lh[c], sizeof(uint16_t));
    }
    r->dw = (p->width + 3) >> 2;
    r->dh = (p->height + 3) >> 2;
    r->decoded = (uint8_t *)calloc((size_t)r->dw * r->dh, 1);
    r->ctbw = oh_ctb_width(p); r->ctbh = oh_ctb_height(p); r->n_ctb = r->ctbw * r->ctbh;
    r->ctu_dep = (uint8_t *)calloc((size_t)r->n_ctb, 1);
    r->ctu_nsub = (uint16_t *)calloc((size_t)r->n_ctb, sizeof(uint16_t));
    r->ctu_level = (uint16_t *)calloc((size_t)r->n_ctb, sizeof(uint16_t));
    r->ctu_entry = (uint32_t *)calloc((size_t)r->n_ctb, sizeof(uint32_t));
    r->ictu = (OhIntraCtu *)calloc((size_t)r->n_ctb, sizeof(OhIntraCtu));
    r->level_start = (uint32_t *)calloc((size_t)r->n_ctb + 2, sizeof(uint32_t));
    r->f.bs_size = oh_bs_size(p);
    r->vbs = (uint8_t *)calloc(r->f.bs_size, 1);
    r->hbs = (uint8_t *)calloc(r->f.bs_size, 1);
    r->qp = (int8_t *)calloc(oh_qp_tab_size(p), 1);
    r->is_pcm = (uint8_t *)calloc((size_t)oh_min_pu_width(p) * oh_min_pu_height(p) + 1, 1);
    r->is_intra = (uint8_t *)calloc((size_t)oh_min_pu_width(p) * oh_min_pu_height(p) + 1, 1);
    r->deblock = (OhDeblockCtb *)calloc((size_t)r->n_ctb, sizeof(OhDeblockCtb));
    r->sao = (OhSaoCtb *)calloc((size_t)r->n_ctb, sizeof(OhSaoCtb));
    int ok = r->decoded && r->ctu_dep && r->ctu_nsub && r->ctu_level && r->ctu_entry && r->ictu && r->level_start && r->vbs && r->hbs &&
             r->qp && r->is_pcm && r->is_intra && r->deblock && r->sao;
    for (int c = 0; c < nplanes; c++) ok = ok && r->lvl[c];
    if (!ok) {
        oh_rec_destroy(r);
        return NULL;
    }
    return r;
}

const OhPicParams *oh_rec_params(const OhRecorder *r) { return &r->f.p; }

void oh_rec_destroy(OhRecorder *r)
{
    if (r) pthread_mutex_destroy(&r->mu);
    if (!r)
        return;
    for (int k = 0; k < OH_MAX_SHARDS; k++) {
        Shard *sh = &r->sh[k];
        free(sh->pu); free(sh->wp); free(sh->tu); free(sh->coeffs); free(sh->intra); free(sh->it_ctu); free(sh->it_sub);
        free(sh->sparse); free(sh->tu_sparse); free(sh->tu_cross);
    }
    free(r->pu); free(r->wp); free(r->tu); free(r->coeffs);
    free(r->sorted); free(r->ctu_dep); free(r->ctu_nsub); free(r->ctu_level); free(r->ctu_entry); free(r->ictu);
    free(r->sub_start); free(r->level_start);
    free(r->bs_mvf); free(r->bs_cbf); free(r->bs_call); free(r->bs_flags);
    free(r->ctb_maps.slice_addr); free(r->ctb_maps.filter_slice_edges); free(r->ctb_maps.deblock_disabled); free(r->ctb_maps.tile_id);
    free(r->sao_pending);
    free(r->vbs); free(r->hbs); free(r->is_pcm); free(r->is_intra); free(r->sparse); free(r->tu_sparse); free(r->tu_cross); free(r->qp); free(r->deblock); free(r->sao);
    for (int c = 0; c < 3; c++)
        free(r->lvl[c]);
    free(r->decoded);
    free(r);
}

static void oh_rec_clear_maps(OhRecorder *r);
void oh_rec_begin(OhRecorder *r, int cur_pic, const int32_t *ref_pics, int n_ref_pics)
{
    r->f.cur_pic = cur_pic;
    for (int i = 0; i < OH_MAX_REFS; i++)
        r->f.ref_pics[i] = i < n_ref_pics ? ref_pics[i] : -1;
    r->f.n_pu = r->f.n_wp = r->f.n_tu = r->f.n_intra = r->f.n_levels = r->f.n_ictu = r->f.n_sub = 0;
    r->f.n_sparse = 0; r->bs_in = NULL;
    r->f.n_coeff = 0;
    r->oom = 0;
    for (int k = 0; k < r->n_sh; k++) {
        Shard *sh = &r->sh[k];
        sh->n_pu = sh->n_wp = sh->n_tu = sh->n_intra = sh->n_sparse = 0; sh->n_coeff = 0;
        sh->any_sparse = sh->any_matrix = sh->any_cross = sh->oom = 0;
    }
    r->n_sh = 0;
    r->epoch = __sync_add_and_fetch(&g_epoch, 1);
    if (r->clean) {                                          /* oh_rec_recycle() already cleared the maps behind the previous picture */
        r->clean = 0;
        return;
    }
    oh_rec_clear_maps(r);
}

/* the per-picture maps back to zero: ~4.6 MB for a 4K picture.  oh_rec_begin() does it unless oh_rec_recycle() has: a decoder with
 * frame threads calls that one behind the hand-over of the previous picture, on the worker's own time — oh_rec_begin() sits between
 * frame start and ff_thread_finish_setup, the one stretch that is serial across the workers */
static void oh_rec_clear_maps(OhRecorder *r)
{
    const OhPicParams *p = &r->f.p;
    r->ctb_maps_on = 0;
    memset(r->vbs, 0, r->f.bs_size);
    memset(r->hbs, 0, r->f.bs_size);
    memset(r->is_pcm, 0, (size_t)oh_min_pu_width(p) * oh_min_pu_height(p));
    memset(r->is_intra, 0, (size_t)oh_min_pu_width(p) * oh_min_pu_height(p));
    memset(r->sao, 0, (size_t)r->n_ctb * sizeof(OhSaoCtb));
    memset(r->deblock, 0, (size_t)r->n_ctb * sizeof(OhDeblockCtb));
    memset(r->ctu_dep, 0, (size_t)r->n_ctb);
    memset(r->ctu_nsub, 0, (size_t)r->n_ctb * sizeof(uint16_t));
    for (int c = 0; c < 3; c++)
        if (r->lvl[c])
            memset(r->lvl[c], 0, (size_t)r->lw[c] * r->lh[c] * sizeof(uint16_t));
    memset(r->decoded, 0, (size_t)r->dw * r->dh);
}
void oh_rec_recycle(OhRecorder *r)
{
    if (!r || r->clean)
        return;
    oh_rec_clear_maps(r);
    r->clean = 1;
}

static int oh_rec_pu_u(OhRecorder *r, Shard *s, int x, int y, int w, int h, int ref0, int mv0x, int mv0y,
              int ref1, int mv1x, int mv1y, const OhWeights *wp)
{
    const OhPicParams *p = &r->f.p;
    if (x < 0 || y < 0 || w < 4 || h < 4 || w > 64 || h > 64 || x + w > p->width || y + h > p->height)
        return -1;
    if (ref0 < 0 && ref1 < 0)
        return -1;
    GROW32(s, s->pu, s->cap_pu, (uint64_t)s->n_pu + 1);
    if (wp) GROW32(s, s->wp, s->cap_wp, (uint64_t)s->n_wp + 1);
    if (s->oom)
        return -1;
    OhPu *it = &s->pu[s->n_pu++];
    memset(it, 0, sizeof(*it));
    it->x = (uint16_t)x; it->y = (uint16_t)y; it->w = (uint8_t)w; it->h = (uint8_t)h;
    it->ref[0] = ref0 < 0 ? OH_NO_REF : (uint8_t)ref0;
    it->ref[1] = ref1 < 0 ? OH_NO_REF : (uint8_t)ref1;
    it->mv[0][0] = (int16_t)mv0x; it->mv[0][1] = (int16_t)mv0y;
    it->mv[1][0] = (int16_t)mv1x; it->mv[1][1] = (int16_t)mv1y;
    it->wp = OH_NO_WP;
    if (wp) {
        s->wp[s->n_wp] = *wp;
        it->wp = (uint16_t)s->n_wp++;
    }
    return 0;
}

static uint32_t oh_rec_tu_u(Shard *s, int c_idx, int x, int y, int log2_size, int kind, int flags,
                   const int16_t *coeffs)
{
    uint32_t n2 = 1u << (2 * log2_size);
    GROW32(s, s->tu, s->cap_tu, (uint64_t)s->n_tu + 1);
    s->coeffs = (int16_t *)grow(&s->oom, s->coeffs, sizeof(int16_t), &s->cap_coeff, s->n_coeff + n2);
    s->tu_sparse = (uint32_t *)grow(&s->oom, s->tu_sparse, sizeof(uint32_t), &s->cap_tu_sparse, (uint64_t)s->n_tu + 1);
    if (s->any_cross) s->tu_cross = (uint32_t *)grow(&s->oom, s->tu_cross, sizeof(uint32_t), &s->cap_tu_cross, (uint64_t)s->n_tu + 1);
    if (s->oom)
        return OH_NO_COEFF;
    OhTu *it = &s->tu[s->n_tu];
    it->x = (uint16_t)x; it->y = (uint16_t)y; it->c_idx = (uint8_t)c_idx; it->log2_size = (uint8_t)log2_size;
    it->kind = (uint8_t)kind; it->flags = (uint8_t)flags;
    it->coeff_off = (uint32_t)s->n_coeff;
    memcpy(s->coeffs + s->n_coeff, coeffs, n2 * sizeof(int16_t));
    s->n_coeff += n2;
    s->tu_sparse[s->n_tu] = OH_NO_COEFF;
    if (s->any_cross) s->tu_cross[s->n_tu] = OH_NO_COEFF;
    return s->n_tu++;
}

static uint32_t oh_rec_tu_sparse_u(Shard *s, int c_idx, int x, int y, int log2_size, int kind, int flags,
                          int qp, int matrix_id, int n, const uint32_t *pairs)
{
    uint32_t n2 = 1u << (2 * log2_size);
    GROW32(s, s->tu, s->cap_tu, (uint64_t)s->n_tu + 1);
    s->coeffs = (int16_t *)grow(&s->oom, s->coeffs, sizeof(int16_t), &s->cap_coeff, s->n_coeff + n2);
    s->tu_sparse = (uint32_t *)grow(&s->oom, s->tu_sparse, sizeof(uint32_t), &s->cap_tu_sparse, (uint64_t)s->n_tu + 1);
    s->sparse = (uint32_t *)grow(&s->oom, s->sparse, sizeof(uint32_t), &s->cap_sparse, (uint64_t)s->n_sparse + 1 + (uint64_t)n);
    if (s->any_cross) s->tu_cross = (uint32_t *)grow(&s->oom, s->tu_cross, sizeof(uint32_t), &s->cap_tu_cross, (uint64_t)s->n_tu + 1);
    if (s->oom)
        return OH_NO_COEFF;
    OhTu *it = &s->tu[s->n_tu];
    it->x = (uint16_t)x; it->y = (uint16_t)y; it->c_idx = (uint8_t)c_idx; it->log2_size = (uint8_t)log2_size;
    it->kind = (uint8_t)kind; it->flags = (uint8_t)(flags | OH_TUF_SPARSE);
    it->coeff_off = (uint32_t)s->n_coeff;
    memset(s->coeffs + s->n_coeff, 0, n2 * sizeof(int16_t));
    s->n_coeff += n2;
    s->tu_sparse[s->n_tu] = s->n_sparse;
    s->sparse[s->n_sparse] = (uint32_t)n | ((uint32_t)(qp & 0xff) << 16) | ((uint32_t)(matrix_id & 0xff) << 24);
    memcpy(s->sparse + s->n_sparse + 1, pairs, sizeof(uint32_t) * (size_t)n);
    s->n_sparse += 1 + (uint32_t)n;
    s->any_sparse = 1;
    if ((matrix_id & 0xff) != OH_FLAT_MATRIX) s->any_matrix = 1;
    if (s->any_cross) s->tu_cross[s->n_tu] = OH_NO_COEFF;
    return s->n_tu++;
}

OhScalingList *oh_rec_scaling_list(OhRecorder *r) { return &r->scaling; }

void oh_rec_bs_inputs(OhRecorder *r, const OhBsInputs *in)
{
    r->bs_in = in;
}

OhBsInputs *oh_rec_bs_maps(OhRecorder *r)
{
    const OhPicParams *p = &r->f.p;
    const size_t n_pu = (size_t)oh_min_pu_width(p) * oh_min_pu_height(p);
    const size_t n_tb = (size_t)(p->width >> p->log2_min_tb_size) * (p->height >> p->log2_min_tb_size);
    if (!r->bs_mvf) {
        r->bs_mvf = (OhMvField *)malloc(n_pu * sizeof(OhMvField));
        r->bs_cbf = (uint8_t *)malloc(n_tb); r->bs_call = (uint8_t *)malloc(n_tb); r->bs_flags = (uint8_t *)malloc((size_t)r->n_ctb);
        if (!r->bs_mvf || !r->bs_cbf || !r->bs_call || !r->bs_flags)
            return NULL;
    }
    if (r->bs_in != &r->own_bs) {                             /* first request for this picture: cleared like the grids they replace */
        memset(r->bs_mvf, 0, n_pu * sizeof(OhMvField));
        memset(r->bs_cbf, 0, n_tb); memset(r->bs_call, 0, n_tb); memset(r->bs_flags, 0, (size_t)r->n_ctb);
        r->own_bs.mvf = r->bs_mvf; r->own_bs.cbf_luma = r->bs_cbf; r->own_bs.call_log2 = r->bs_call; r->own_bs.ctb_flags = r->bs_flags;
        r->own_bs.loop_filter_across_tiles = 1;
        r->bs_in = &r->own_bs;
    }
    return &r->own_bs;
}

/* see ohevc_recorder.h.  ff_hevc_hls_filters (hevc_filter.c:1055-1064) after decoding CTB (X, Y) calls ff_hevc_hls_filter for
 * (X-1, Y-1), for (X, Y-1) at the end of a CTB row and for (X-1, Y) in the last CTB row; the bottom-right CTB gets its call when the
 * slice data ends (hevc.c:2693-2695).  ff_hevc_hls_filter(x, y) (:1027-1052) = deblocking_filter_CTB(x, y), then sao_filter_CTB of
 * (x-1, y-1), of (x-1, y) in the last row, of (x, y-1) in the last column and of (x, y) in the corner. */
void oh_sao_pending_driver(const int32_t *tile_id, int W, int H, uint8_t *out)
{
    const int n = W * H;
    uint32_t *t_db = (uint32_t *)calloc((size_t)n * 2, sizeof(uint32_t)), *t_sao = t_db ? t_db + n : NULL;
    int *order = (int *)malloc((size_t)n * sizeof(int));
    memset(out, 0, (size_t)n);
    if (!t_db || !order) { free(t_db); free(order); return; }
    /* decoding order: tile after tile (ids grow in tile scan), raster inside a tile — a stable counting sort by id */
    int max_id = 0;
    for (int i = 0; i < n; i++) if (tile_id[i] > max_id) max_id = tile_id[i];
    int *start = (int *)calloc((size_t)max_id + 2, sizeof(int));
    if (!start) { free(t_db); free(order); return; }
    for (int i = 0; i < n; i++) start[(tile_id[i] < 0 ? 0 : tile_id[i]) + 1]++;
    for (int k = 0; k <= max_id; k++) start[k + 1] += start[k];
    for (int i = 0; i < n; i++) order[start[tile_id[i] < 0 ? 0 : tile_id[i]]++] = i;
    free(start);
    uint32_t t = 0;
#define OH_FILTER_(x, y)                                                                   \
    do {                                                                                   \
        const int fx = (x), fy = (y), xe = fx == W - 1, ye = fy == H - 1;                  \
        t_db[fy * W + fx] = ++t;                                                           \
        if (fy && fx) t_sao[(fy - 1) * W + fx - 1] = ++t;                                  \
        if (fx && ye) t_sao[fy * W + fx - 1] = ++t;                                        \
        if (fy && xe) t_sao[(fy - 1) * W + fx] = ++t;                                      \
        if (xe && ye) t_sao[fy * W + fx] = ++t;                                            \
    } while (0)
    for (int k = 0; k < n; k++) {
        const int X = order[k] % W, Y = order[k] / W, xe = X == W - 1, ye = Y == H - 1;
        if (Y && X) OH_FILTER_(X - 1, Y - 1);
        if (Y && xe) OH_FILTER_(X, Y - 1);
        if (X && ye) OH_FILTER_(X - 1, Y);
    }
    OH_FILTER_(W - 1, H - 1);
#undef OH_FILTER_
    for (int cy = 0; cy < H; cy++)
        for (int cx = 0; cx + 2 < W; cx++) {
            const uint32_t me = t_sao[cy * W + cx];
            int bits = 0;
            if (!(t_db[cy * W + cx + 2] < me)) bits |= 1;
            if (cy + 1 < H && !(t_db[(cy + 1) * W + cx + 2] < me)) bits |= 2;
            out[cy * W + cx] = (uint8_t)bits;
        }
    free(t_db); free(order);
}

OhCtbMaps *oh_rec_ctb_maps(OhRecorder *r)
{
    OhCtbMaps *m = &r->ctb_maps;
    const size_t n = (size_t)r->n_ctb;
    if (!m->slice_addr) {
        m->slice_addr = (int32_t *)malloc(n * sizeof(int32_t)); m->tile_id = (int32_t *)malloc(n * sizeof(int32_t));
        m->filter_slice_edges = (uint8_t *)malloc(n); m->deblock_disabled = (uint8_t *)malloc(n);
        if (!m->slice_addr || !m->tile_id || !m->filter_slice_edges || !m->deblock_disabled) {
            free(m->slice_addr); free(m->tile_id); free(m->filter_slice_edges); free(m->deblock_disabled);
            memset(m, 0, sizeof(*m));
            r->oom = 1;
            return NULL;
        }
    }
    if (!r->ctb_maps_on) {
        memset(m->slice_addr, 0, n * sizeof(int32_t)); memset(m->tile_id, 0, n * sizeof(int32_t));
        memset(m->filter_slice_edges, 1, n); memset(m->deblock_disabled, 0, n);
        m->tiles_enabled = 0; m->loop_filter_across_tiles = 1;
        r->ctb_maps_on = 1;
    }
    return m;
}

const OhCtbMaps *oh_rec_ctb_maps_in_use(const OhRecorder *r) { return r->ctb_maps_on ? &r->ctb_maps : NULL; }

static int oh_rec_tu_cross_u(Shard *s, uint32_t tu_c, uint32_t tu_y, int res_scale_val)
{
    if (tu_c >= s->n_tu || tu_y >= s->n_tu || tu_y >= (1u << 24) || s->tu[tu_y].c_idx != 0 || s->tu[tu_c].c_idx == 0 ||
        s->tu[tu_y].log2_size != s->tu[tu_c].log2_size)
        return -1;
    if (!s->any_cross) {                                  /* first one of the picture (in this shard): start from "none" */
        s->tu_cross = (uint32_t *)grow(&s->oom, s->tu_cross, sizeof(uint32_t), &s->cap_tu_cross, (uint64_t)s->n_tu + 1);
        if (s->oom)
            return -1;
        for (uint32_t i = 0; i < s->n_tu; i++) s->tu_cross[i] = OH_NO_COEFF;
        s->any_cross = 1;
    }
    s->tu_cross[tu_c] = tu_y | ((uint32_t)(res_scale_val & 0xff) << 24);
    s->tu[tu_c].flags |= OH_TUF_CROSS;
    s->tu[tu_y].flags |= OH_TUF_KEEP_RES;
    return 0;
}

/* one neighbour cell (plane c, sample position x,y) read by a block of CTU (cx,cy) */
static inline void visit(OhRecorder *r, int c, int x, int y, int cx, int cy, unsigned *sub, unsigned *dep)
{
    const OhPicParams *p = &r->f.p;
    unsigned v = r->lvl[c][(y >> 2) * r->lw[c] + (x >> 2)];
    if (!v)
        return;                                           /* inter / PCM-free sample: no dependency */
    int nx = (x << oh_hshift(p, c)) >> p->log2_ctb_size, ny = (y << oh_vshift(p, c)) >> p->log2_ctb_size;
    if (nx == cx && ny == cy) {
        if (v > *sub) *sub = v;
    } else if (ny == cy) {
        *dep |= 1;                                        /* left CTU (also its rows below: bottom-left samples) */
    } else {
        *dep |= nx < cx ? 2 : (nx == cx ? 4 : 8);         /* up-left, up, up-right */
    }
}

static int oh_rec_intra_u(OhRecorder *r, Shard *s, int c_idx, int x, int y, int log2_size, int mode, int avail, uint32_t tu)
{
    const OhPicParams *p = &r->f.p;
    int n = 1 << log2_size, c = c_idx;
    int pw = p->width >> oh_hshift(p, c), ph = p->height >> oh_vshift(p, c);
    if (x < 0 || y < 0 || x + n > pw || y + n > ph)
        return -1;
    int cx = (x << oh_hshift(p, c)) >> p->log2_ctb_size, cy = (y << oh_vshift(p, c)) >> p->log2_ctb_size;
    unsigned sub = 0, dep = 0;
    /* the samples intra_pred() gathers (hevcpred_template.c:164-183): one column to the left over
     * 2N rows, one row above over 2N columns, and the corner — each only where available */
    if (avail & OH_AV_UP_LEFT) visit(r, c, x - 1, y - 1, cx, cy, &sub, &dep);
    if (avail & OH_AV_UP)
        for (int i = 0; i < n; i += 4) visit(r, c, x + i, y - 1, cx, cy, &sub, &dep);
    if (avail & OH_AV_UP_RIGHT)
        for (int i = n; i < 2 * n && x + i < pw; i += 4) visit(r, c, x + i, y - 1, cx, cy, &sub, &dep);
    if (avail & OH_AV_LEFT)
        for (int i = 0; i < n; i += 4) visit(r, c, x - 1, y + i, cx, cy, &sub, &dep);
    if (avail & OH_AV_BOTTOM_LEFT)
        for (int i = n; i < 2 * n && y + i < ph; i += 4) visit(r, c, x - 1, y + i, cx, cy, &sub, &dep);
    sub += 1;
    for (int yy = y; yy < y + n; yy += 4)
        for (int xx = x; xx < x + n; xx += 4)
            r->lvl[c][(yy >> 2) * r->lw[c] + (xx >> 2)] = (uint16_t)sub;
    int ctu = cy * r->ctbw + cx;
    r->ctu_dep[ctu] |= (uint8_t)dep;
    if (sub > r->ctu_nsub[ctu])
        r->ctu_nsub[ctu] = (uint16_t)sub;

    uint32_t cap = s->cap_intra;
    GROW32(s, s->intra, s->cap_intra, (uint64_t)s->n_intra + 1);
    if (s->cap_intra != cap) {
        uint32_t *nc = (uint32_t *)realloc(s->it_ctu, sizeof(uint32_t) * s->cap_intra);
        if (nc) s->it_ctu = nc;
        uint16_t *nsb = (uint16_t *)realloc(s->it_sub, sizeof(uint16_t) * s->cap_intra);
        if (nsb) s->it_sub = nsb;
        if (!nc || !nsb) { s->oom = 1; s->cap_intra = cap; }       /* the item list keeps its old capacity: nothing indexes past the side lists */
    }
    if (s->oom)
        return -1;
    OhIntra *it = &s->intra[s->n_intra];
    it->x = (uint16_t)x; it->y = (uint16_t)y; it->c_idx = (uint8_t)c; it->log2_size = (uint8_t)log2_size;
    it->mode = (uint8_t)mode; it->avail = (uint8_t)avail; it->tu = tu;
    s->it_ctu[s->n_intra] = (uint32_t)ctu;
    s->it_sub[s->n_intra] = (uint16_t)sub;
    s->n_intra++;
    return 0;
}

uint32_t oh_rec_n_intra(const OhRecorder *r) { return my_shard((OhRecorder *)r)->n_intra; }

static int oh_rec_intra_attach_tu_u(Shard *s, uint32_t intra_index, uint32_t tu)
{
    if (intra_index >= s->n_intra || tu >= s->n_tu)
        return -1;
    s->intra[intra_index].tu = tu;
    s->tu[tu].flags &= (uint8_t)~OH_TUF_ADD_NOW;
    return 0;
}

uint8_t      *oh_rec_vertical_bs(OhRecorder *r)   { return r->vbs; }
uint8_t      *oh_rec_horizontal_bs(OhRecorder *r) { return r->hbs; }
int8_t       *oh_rec_qp_y_tab(OhRecorder *r)      { return r->qp; }
uint8_t      *oh_rec_is_pcm(OhRecorder *r)        { return r->is_pcm; }
uint8_t      *oh_rec_is_intra(OhRecorder *r)      { return r->is_intra; }
OhDeblockCtb *oh_rec_deblock(OhRecorder *r)       { return r->deblock; }
OhSaoCtb     *oh_rec_sao(OhRecorder *r)           { return r->sao; }

const OhFrame *oh_rec_finish(OhRecorder *r)
{
    OhFrame *f = &r->f;
    const int W = r->ctbw;
    uint32_t max_level = 0, n_ictu = 0, n_sub = 0;

    /* 0. the threads' shards -> the picture's lists.  One shard (the common case: one decoding thread, the synthetic generator):
     * its lists ARE the picture's.  Several: concatenated in shard order, with the links rebased (see struct Shard). */
    int any_sparse = 0, any_matrix = 0, any_cross = 0;
    uint32_t base_tu[OH_MAX_SHARDS + 1], base_wp[OH_MAX_SHARDS + 1], base_pu[OH_MAX_SHARDS + 1], base_sparse[OH_MAX_SHARDS + 1];
    uint64_t base_coeff[OH_MAX_SHARDS + 1];
    base_tu[0] = base_wp[0] = base_pu[0] = base_sparse[0] = 0; base_coeff[0] = 0;
    f->n_intra = 0;
    for (int k = 0; k < r->n_sh; k++) {
        const Shard *sh = &r->sh[k];
        if (sh->oom) r->oom = 1;
        any_sparse |= sh->any_sparse; any_matrix |= sh->any_matrix; any_cross |= sh->any_cross;
        base_pu[k + 1] = base_pu[k] + sh->n_pu; base_wp[k + 1] = base_wp[k] + sh->n_wp; base_tu[k + 1] = base_tu[k] + sh->n_tu;
        base_sparse[k + 1] = base_sparse[k] + sh->n_sparse; base_coeff[k + 1] = base_coeff[k] + sh->n_coeff;
        f->n_intra += sh->n_intra;
    }
    if (r->oom || base_coeff[r->n_sh] >= 0xffffffffull)    /* OhTu.coeff_off is 32 bits */
        return NULL;
    f->n_pu = base_pu[r->n_sh]; f->n_wp = base_wp[r->n_sh]; f->n_tu = base_tu[r->n_sh]; f->n_sparse = base_sparse[r->n_sh];
    f->n_coeff = base_coeff[r->n_sh];
    const OhPu *l_pu = NULL; const OhWeights *l_wp = NULL; const OhTu *l_tu = NULL; const int16_t *l_coeffs = NULL;
    const uint32_t *l_sparse = NULL, *l_tu_sparse = NULL, *l_tu_cross = NULL;
    if (r->n_sh == 1) {
        const Shard *sh = &r->sh[0];
        l_pu = sh->pu; l_wp = sh->wp; l_tu = sh->tu; l_coeffs = sh->coeffs; l_sparse = sh->sparse; l_tu_sparse = sh->tu_sparse; l_tu_cross = sh->tu_cross;
    } else if (r->n_sh > 1) {
        GROW32(r, r->pu, r->cap_pu, (uint64_t)f->n_pu + 1);
        GROW32(r, r->wp, r->cap_wp, (uint64_t)f->n_wp + 1);
        GROW32(r, r->tu, r->cap_tu, (uint64_t)f->n_tu + 1);
        r->coeffs = (int16_t *)grow(&r->oom, r->coeffs, sizeof(int16_t), &r->cap_coeff, f->n_coeff + 1);
        r->tu_sparse = (uint32_t *)grow(&r->oom, r->tu_sparse, sizeof(uint32_t), &r->cap_tu_sparse, (uint64_t)f->n_tu + 1);
        if (any_sparse) r->sparse = (uint32_t *)grow(&r->oom, r->sparse, sizeof(uint32_t), &r->cap_sparse, (uint64_t)f->n_sparse + 1);
        if (any_cross) r->tu_cross = (uint32_t *)grow(&r->oom, r->tu_cross, sizeof(uint32_t), &r->cap_tu_cross, (uint64_t)f->n_tu + 1);
        if (r->oom)
            return NULL;
        for (int k = 0; k < r->n_sh; k++) {
            const Shard *sh = &r->sh[k];
            for (uint32_t i = 0; i < sh->n_pu; i++) {
                OhPu v = sh->pu[i];
                if (v.wp != OH_NO_WP) v.wp = (uint16_t)(v.wp + base_wp[k]);
                r->pu[base_pu[k] + i] = v;
            }
            if (sh->n_wp) memcpy(r->wp + base_wp[k], sh->wp, sizeof(OhWeights) * sh->n_wp);
            for (uint32_t i = 0; i < sh->n_tu; i++) {
                OhTu v = sh->tu[i];
                v.coeff_off = (uint32_t)(v.coeff_off + base_coeff[k]);
                r->tu[base_tu[k] + i] = v;
                const uint32_t sp = sh->tu_sparse[i];
                r->tu_sparse[base_tu[k] + i] = sp == OH_NO_COEFF ? OH_NO_COEFF : sp + base_sparse[k];
                if (any_cross) {
                    const uint32_t cr = sh->any_cross ? sh->tu_cross[i] : OH_NO_COEFF;
                    r->tu_cross[base_tu[k] + i] = cr == OH_NO_COEFF ? OH_NO_COEFF : (((cr & 0xffffffu) + base_tu[k]) | (cr & 0xff000000u));
                }
            }
            if (sh->n_coeff) memcpy(r->coeffs + base_coeff[k], sh->coeffs, sizeof(int16_t) * sh->n_coeff);
            if (sh->n_sparse) memcpy(r->sparse + base_sparse[k], sh->sparse, sizeof(uint32_t) * sh->n_sparse);
        }
        if (f->n_wp > 0xfffe || (any_cross && f->n_tu >= (1u << 24)))     /* the 16-bit weight index / the 24-bit luma link of a cross-component block */
            return NULL;
        l_pu = r->pu; l_wp = r->wp; l_tu = r->tu; l_coeffs = r->coeffs; l_sparse = r->sparse; l_tu_sparse = r->tu_sparse; l_tu_cross = r->tu_cross;
    }
    /* 1. CTU levels, in raster order (every dependency points to an earlier CTU) */
    for (int i = 0; i < r->n_ctb; i++) {
        unsigned lv = 0;
        if (r->ctu_nsub[i]) {
            int x = i % W;
            unsigned d = r->ctu_dep[i], t;
            lv = 1;
            if ((d & 1) && x > 0                 && (t = r->ctu_level[i - 1]))     { if (t + 1 > lv) lv = t + 1; }
            if ((d & 2) && x > 0 && i >= W       && (t = r->ctu_level[i - W - 1])) { if (t + 1 > lv) lv = t + 1; }
            if ((d & 4) && i >= W                && (t = r->ctu_level[i - W]))     { if (t + 1 > lv) lv = t + 1; }
            if ((d & 8) && x + 1 < W && i >= W   && (t = r->ctu_level[i - W + 1])) { if (t + 1 > lv) lv = t + 1; }
            if (lv > max_level) max_level = lv;
            n_ictu++;
        }
        r->ctu_level[i] = (uint16_t)lv;
    }
    /* 2. order the CTUs by level (counting sort, raster order inside a level) */
    memset(r->level_start, 0, sizeof(uint32_t) * (max_level + 2));
    for (int i = 0; i < r->n_ctb; i++)
        if (r->ctu_level[i])
            r->level_start[r->ctu_level[i]]++;
    uint32_t acc = 0;
    for (uint32_t l = 1; l <= max_level; l++) {
        uint32_t cnt = r->level_start[l];
        r->level_start[l - 1] = acc;
        acc += cnt;
    }
    r->level_start[max_level] = acc;
    uint32_t *cursor = (uint32_t *)malloc(sizeof(uint32_t) * (max_level + 1));
    if (!cursor)
        return NULL;
    memcpy(cursor, r->level_start, sizeof(uint32_t) * (max_level + 1));
    for (int i = 0; i < r->n_ctb; i++)
        if (r->ctu_level[i])
            r->ctu_entry[i] = cursor[r->ctu_level[i] - 1]++;
    free(cursor);
    /* 3. sub-level ranges: CTU after CTU in schedule order */
    for (int i = 0; i < r->n_ctb; i++)
        if (r->ctu_level[i]) {
            OhIntraCtu *e = &r->ictu[r->ctu_entry[i]];
            e->ctu = (uint16_t)i;
            e->n_sub = r->ctu_nsub[i];
        }
    for (uint32_t k = 0; k < n_ictu; k++) {
        r->ictu[k].sub_first = n_sub;
        n_sub += r->ictu[k].n_sub;
    }
    GROW32(r, r->sub_start, r->cap_sub, (uint64_t)n_sub + 2);
    GROW32(r, r->sorted, r->cap_sorted, (uint64_t)f->n_intra + 1);
    uint32_t *pos = (uint32_t *)malloc(sizeof(uint32_t) * (n_sub + 1));
    if (r->oom || !pos) {
        free(pos);
        return NULL;
    }
    memset(r->sub_start, 0, sizeof(uint32_t) * (n_sub + 2));
    for (int k = 0; k < r->n_sh; k++) {                    /* histogram at slot+1 */
        const Shard *sh = &r->sh[k];
        for (uint32_t i = 0; i < sh->n_intra; i++)
            r->sub_start[r->ictu[r->ctu_entry[sh->it_ctu[i]]].sub_first + sh->it_sub[i] - 1 + 1]++;
    }
    for (uint32_t s = 0; s < n_sub; s++)
        r->sub_start[s + 1] += r->sub_start[s];
    /* 4. place the blocks (stable: recording order inside one (CTU, sub-level)) */
    memcpy(pos, r->sub_start, sizeof(uint32_t) * (n_sub + 1));
    for (int k = 0; k < r->n_sh; k++) {
        const Shard *sh = &r->sh[k];
        for (uint32_t i = 0; i < sh->n_intra; i++) {
            OhIntra v = sh->intra[i];
            if (v.tu != OH_NO_COEFF) v.tu += base_tu[k];
            r->sorted[pos[r->ictu[r->ctu_entry[sh->it_ctu[i]]].sub_first + sh->it_sub[i] - 1]++] = v;
        }
    }
    free(pos);

    f->sao_pending = NULL;
    if (r->ctb_maps_on) {                                  /* slices / tiles: what the SAO and BS passes need of them */
        for (int i = 0; i < r->n_ctb; i++)
            r->sao[i].edge_flags = (uint8_t)oh_ctb_sao_edge_flags(&r->ctb_maps, r->ctbw, r->ctbh, i);
        /* tiles change the order of the reference's filter calls; it shows in one configuration (ohevc_frame.h: sao_pending) */
        if (r->ctb_maps.tiles_enabled && f->p.log2_ctb_size == 4 && oh_hshift(&f->p, 1)) {
            if (!r->sao_pending)
                r->sao_pending = (uint8_t *)malloc((size_t)r->n_ctb);
            if (r->sao_pending) {
                oh_sao_pending_driver(r->ctb_maps.tile_id, r->ctbw, r->ctbh, r->sao_pending);
                f->sao_pending = r->sao_pending;
            }
        }
        if (r->bs_in == &r->own_bs) {
            for (int i = 0; i < r->n_ctb; i++)
                r->bs_flags[i] = (uint8_t)oh_ctb_bs_flags(&r->ctb_maps, r->ctbw, i);
            r->own_bs.loop_filter_across_tiles = r->ctb_maps.loop_filter_across_tiles;
        }
    }
    f->pu = l_pu; f->wp = l_wp; f->tu = l_tu; f->coeffs = l_coeffs;
    f->intra = r->sorted;
    f->n_ictu = n_ictu; f->ictu = r->ictu;
    f->n_sub = n_sub; f->sub_start = r->sub_start;
    f->n_levels = max_level; f->level_start = r->level_start;
    f->vertical_bs = r->vbs; f->horizontal_bs = r->hbs; f->qp_y_tab = r->qp;
    f->is_pcm = (f->p.pcm_loop_filter_disable || f->p.transquant_bypass_enable) ? r->is_pcm : NULL;
    f->is_intra = f->p.constrained_intra_pred ? r->is_intra : NULL;
    f->sparse = any_sparse ? l_sparse : NULL;
    f->tu_sparse = any_sparse ? l_tu_sparse : NULL;
    f->scaling = any_matrix ? &r->scaling : NULL;
    f->tu_cross = any_cross ? l_tu_cross : NULL;
    f->bs_in = r->bs_in;
    f->deblock = r->deblock;
    f->sao = f->p.sao_enabled ? r->sao : NULL;
    return f;
}

/* ---- availability from the recorder's own decoded map ---- */
static int dec_at(const OhRecorder *r, int x, int y)
{
    if (x < 0 || y < 0 || x >= r->f.p.width || y >= r->f.p.height)
        return 0;
    return r->decoded[(y >> 2) * r->dw + (x >> 2)];
}

/* decoded earlier AND in the same slice and tile as (x0, y0) (6.4.1; the reference's ctb_*_flag, hevc.c:2638-2641) */
static int avail_at(const OhRecorder *r, int x0, int y0, int x, int y)
{
    if (!dec_at(r, x, y))
        return 0;
    if (r->ctb_maps_on) {
        const int lc = r->f.p.log2_ctb_size, a = (y0 >> lc) * r->ctbw + (x0 >> lc), b = (y >> lc) * r->ctbw + (x >> lc);
        if (r->ctb_maps.slice_addr[a] != r->ctb_maps.slice_addr[b] || r->ctb_maps.tile_id[a] != r->ctb_maps.tile_id[b])
            return 0;
    }
    return 1;
}

int oh_rec_avail(const OhRecorder *r, int x, int y, int w, int h)
{
    int a = 0;
    if (avail_at(r, x, y, x - 1, y + h)) a |= OH_AV_BOTTOM_LEFT;
    if (avail_at(r, x, y, x - 1, y))     a |= OH_AV_LEFT;
    if (avail_at(r, x, y, x - 1, y - 1)) a |= OH_AV_UP_LEFT;
    if (avail_at(r, x, y, x, y - 1))     a |= OH_AV_UP;
    if (avail_at(r, x, y, x + w, y - 1)) a |= OH_AV_UP_RIGHT;
    return a;
}

static void oh_rec_mark_decoded_u(OhRecorder *r, int x, int y, int w, int h)
{
    for (int yy = y; yy < y + h && yy < r->f.p.height; yy += 4)
        for (int xx = x; xx < x + w && xx < r->f.p.width; xx += 4)
            r->decoded[(yy >> 2) * r->dw + (xx >> 2)] = 1;
}

/* ---- the public appending entry points: the bodies above on the calling thread's own lists (no lock) ---- */
int oh_rec_pu(OhRecorder *r, int x, int y, int w, int h, int ref0, int mv0x, int mv0y,
              int ref1, int mv1x, int mv1y, const OhWeights *wp)
{
    return oh_rec_pu_u(r, my_shard(r), x, y, w, h, ref0, mv0x, mv0y, ref1, mv1x, mv1y, wp);
}
uint32_t oh_rec_tu(OhRecorder *r, int c_idx, int x, int y, int log2_size, int kind, int flags,
                   const int16_t *coeffs)
{
    return oh_rec_tu_u(my_shard(r), c_idx, x, y, log2_size, kind, flags, coeffs);
}
uint32_t oh_rec_tu_sparse(OhRecorder *r, int c_idx, int x, int y, int log2_size, int kind, int flags,
                          int qp, int matrix_id, int n, const uint32_t *pairs)
{
    return oh_rec_tu_sparse_u(my_shard(r), c_idx, x, y, log2_size, kind, flags, qp, matrix_id, n, pairs);
}
int oh_rec_tu_cross(OhRecorder *r, uint32_t tu_c, uint32_t tu_y, int res_scale_val)
{
    return oh_rec_tu_cross_u(my_shard(r), tu_c, tu_y, res_scale_val);
}
/* The sub-level map (lvl[]) and the per-CTU dependency words are shared and written without a lock: a block only reads cells of
 * blocks decoded BEFORE it in the same slice and tile (its available neighbours), and the reference's wavefront threads are ordered
 * exactly there (row r starts CTU k when row r-1 has finished k+2: ff_thread_await_progress2, a mutex); a CTU's own words are
 * written by the one thread that decodes it. */
int oh_rec_intra(OhRecorder *r, int c_idx, int x, int y, int log2_size, int mode, int avail, uint32_t tu)
{
    return oh_rec_intra_u(r, my_shard(r), c_idx, x, y, log2_size, mode, avail, tu);
}
int oh_rec_intra_attach_tu(OhRecorder *r, uint32_t intra_index, uint32_t tu)
{
    return oh_rec_intra_attach_tu_u(my_shard(r), intra_index, tu);
}
void oh_rec_mark_decoded(OhRecorder *r, int x, int y, int w, int h)
{
    oh_rec_mark_decoded_u(r, x, y, w, h);
}
uint32_t oh_rec_intra_idx(OhRecorder *r, int c_idx, int x, int y, int log2_size, int mode, int avail, uint32_t tu)
{
    Shard *s = my_shard(r);
    const uint32_t idx = s->n_intra;
    const int rc = oh_rec_intra_u(r, s, c_idx, x, y, log2_size, mode, avail, tu);
    return rc ? OH_NO_COEFF : idx;
}
