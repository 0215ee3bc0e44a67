/*
 * ref_harness.c — thin C entry points around the REAL reference kernels.
 * TEST INFRASTRUCTURE ONLY.  Compiled together with hevcdsp.c / hevcpred.c / videodsp.c /
 * hevc_filter.c taken directly from /root/reference (see oracle/Makefile); the result,
 * oracle/_ref/libohevc_ref.so, exists only in this container (it is git-ignored and is never
 * needed on the GPU box: fixtures generated from it are committed under tests/golden/).
 *
 * Nothing here re-implements reference arithmetic: the functions fill the reference's own
 * structs (HEVCContext, HEVCSPS, HEVCPPS, SAOParams ...) from plain arguments / an OhFrame and
 * call the reference's tables (ff_hevc_dsp_init, ff_hevc_pred_init, ff_videodsp_init) and
 * drivers (ff_hevc_hls_filters / ff_hevc_hls_filter).
 */
#include <stdlib.h>
#include <string.h>

#include "libavcodec/hevc.h"
#include "libavcodec/hevcdsp.h"
#include "libavcodec/hevcpred.h"
#include "libavcodec/videodsp.h"

#include "../include/ohevc_frame.h"
#include "../include/ohevc_recorder.h"          /* OhCtbMaps: the per-CTB slice / tile maps of a picture */

#define API __attribute__((visibility("default")))

/* slices / tiles of the pictures the next calls work on (NULL = one slice, one tile); the maps hold what the reference's CTU
 * loop leaves in s->tab_slice_address, s->filter_slice_edges and pps->tile_id (hevc.c:2600, 2679; hevc_ps.c) */
static const OhCtbMaps *g_maps;
API void ref_set_ctb_maps(const OhCtbMaps *m) { g_maps = m; }

static HEVCDSPContext  g_dsp[15];
static HEVCPredContext g_pred[15];
static VideoDSPContext g_vdsp[15];
static int g_ready[15];

static void tables(int bd)
{
    if (g_ready[bd])
        return;
    ff_hevc_dsp_init(&g_dsp[bd], bd);          /* hevcdsp.c:1071 */
    ff_hevc_pred_init(&g_pred[bd], bd);        /* hevcpred.c:47  */
    ff_videodsp_init(&g_vdsp[bd], bd);         /* videodsp.c:33  */
    g_ready[bd] = 1;
}

#include "replay_slots.inc"                    /* pel_idx(), replay_pu(), replay_inverse(), replay_pcm(): our CTU-loop stand-in */

/* ---------------- residual slots ---------------- */
API void ref_transform_add(int bd, int log2, uint8_t *dst, int16_t *coeffs, ptrdiff_t stride)
{ tables(bd); g_dsp[bd].transform_add[log2 - 2](dst, coeffs, stride); }
API void ref_transform_skip(int bd, int16_t *c, int log2)
{ tables(bd); g_dsp[bd].transform_skip(c, (int16_t)log2); }
API void ref_transform_rdpcm(int bd, int16_t *c, int log2, int mode)
{ tables(bd); g_dsp[bd].transform_rdpcm(c, (int16_t)log2, mode); }
API void ref_idct_4x4_luma(int bd, int16_t *c)
{ tables(bd); g_dsp[bd].idct_4x4_luma(c); }
API void ref_idct(int bd, int log2, int16_t *c, int col_limit)
{ tables(bd); g_dsp[bd].idct[log2 - 2](c, col_limit); }
API void ref_idct_dc(int bd, int log2, int16_t *c)
{ tables(bd); g_dsp[bd].idct_dc[log2 - 2](c); }

/* ---------------- interpolation slots ----------------
 * variant: 0 put (dst is int16_t*, dststride in elements), 1 uni, 2 bi, 3 uni_w, 4 bi_w */
API void ref_mc(int bd, int epel, int variant, uint8_t *dst, ptrdiff_t dststride,
                uint8_t *src, ptrdiff_t srcstride, int16_t *src2, ptrdiff_t src2stride,
                int h, int denom, int wx0, int wx1, int ox0, int ox1, int mx, int my, int w)
{
    HEVCDSPContext *d;
    int i = pel_idx(w), a = !!my, b = !!mx;
    tables(bd);
    d = &g_dsp[bd];
    switch (variant) {
    case 0:
        (epel ? d->put_hevc_epel : d->put_hevc_qpel)[i][a][b]((int16_t *)dst, dststride, src, srcstride, h, mx, my, w);
        break;
    case 1:
        (epel ? d->put_hevc_epel_uni : d->put_hevc_qpel_uni)[i][a][b](dst, dststride, src, srcstride, h, mx, my, w);
        break;
    case 2:
        (epel ? d->put_hevc_epel_bi : d->put_hevc_qpel_bi)[i][a][b](dst, dststride, src, srcstride, src2, src2stride, h, mx, my, w);
        break;
    case 3:
        (epel ? d->put_hevc_epel_uni_w : d->put_hevc_qpel_uni_w)[i][a][b](dst, dststride, src, srcstride, h, denom, wx0, ox0, mx, my, w);
        break;
    default:
        /* argument POSITIONS as at the call sites hevc.c:1767-1773 / 1940-1948 */
        (epel ? d->put_hevc_epel_bi_w : d->put_hevc_qpel_bi_w)[i][a][b](dst, dststride, src, srcstride, src2, src2stride,
                                                                      h, denom, wx0, wx1, ox0, ox1, mx, my, w);
        break;
    }
}

API void ref_emulated_edge_mc(int bd, uint8_t *buf, const uint8_t *src, ptrdiff_t buf_linesize,
                              ptrdiff_t src_linesize, int block_w, int block_h, int src_x, int src_y, int w, int h)
{ tables(bd); g_vdsp[bd].emulated_edge_mc(buf, src, buf_linesize, src_linesize, block_w, block_h, src_x, src_y, w, h); }

/* ---------------- intra slots ---------------- */
API void ref_pred_planar(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride)
{ tables(bd); g_pred[bd].pred_planar[log2 - 2](src, top, left, stride); }
API void ref_pred_dc(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left, ptrdiff_t stride, int c_idx)
{ tables(bd); g_pred[bd].pred_dc(src, top, left, stride, log2, c_idx); }
API void ref_pred_angular(int bd, int log2, uint8_t *src, const uint8_t *top, const uint8_t *left,
                          ptrdiff_t stride, int c_idx, int mode)
{ tables(bd); g_pred[bd].pred_angular[log2 - 2](src, top, left, stride, c_idx, mode); }

/* ---------------- deblock slots; which: 0 h_luma 1 v_luma 2 h_chroma 3 v_chroma, +4 for _c ---------------- */
API void ref_loop_filter(int bd, int which, uint8_t *pix, ptrdiff_t stride, int beta, int *tc,
                         uint8_t *no_p, uint8_t *no_q)
{
    HEVCDSPContext *d;
    tables(bd);
    d = &g_dsp[bd];
    switch (which) {
    case 0: d->hevc_h_loop_filter_luma(pix, stride, beta, tc, no_p, no_q); break;
    case 1: d->hevc_v_loop_filter_luma(pix, stride, beta, tc, no_p, no_q); break;
    case 2: d->hevc_h_loop_filter_chroma(pix, stride, tc, no_p, no_q); break;
    case 3: d->hevc_v_loop_filter_chroma(pix, stride, tc, no_p, no_q); break;
    case 4: d->hevc_h_loop_filter_luma_c(pix, stride, beta, tc, no_p, no_q); break;
    case 5: d->hevc_v_loop_filter_luma_c(pix, stride, beta, tc, no_p, no_q); break;
    case 6: d->hevc_h_loop_filter_chroma_c(pix, stride, tc, no_p, no_q); break;
    default: d->hevc_v_loop_filter_chroma_c(pix, stride, tc, no_p, no_q); break;
    }
}

/* ---------------- SAO slots ---------------- */
static void fill_sao(SAOParams *s, int c_idx, const int16_t *offset_val, int band_position, int eo_class)
{
    memset(s, 0, sizeof(*s));
    for (int k = 0; k < 5; k++)
        s->offset_val[c_idx][k] = offset_val[k];
    s->band_position[c_idx] = (uint8_t)band_position;
    s->eo_class[c_idx] = (uint8_t)eo_class;
}
API void ref_sao_band(int bd, uint8_t *dst, uint8_t *src, ptrdiff_t sd, ptrdiff_t ss, const int16_t *offset_val,
                      int band_position, int *borders, int w, int h, int c_idx)
{
    SAOParams s;
    tables(bd);
    fill_sao(&s, c_idx, offset_val, band_position, 0);
    g_dsp[bd].sao_band_filter(dst, src, sd, ss, &s, borders, w, h, c_idx);
}
API void ref_sao_edge(int bd, int variant, uint8_t *dst, uint8_t *src, ptrdiff_t sd, ptrdiff_t ss,
                      const int16_t *offset_val, int eo_class, int *borders, int w, int h, int c_idx,
                      uint8_t *vert_edge, uint8_t *horiz_edge, uint8_t *diag_edge)
{
    SAOParams s;
    tables(bd);
    fill_sao(&s, c_idx, offset_val, 0, eo_class);
    g_dsp[bd].sao_edge_filter[variant](dst, src, sd, ss, &s, borders, w, h, c_idx, vert_edge, horiz_edge, diag_edge);
}

/* =========================================================================================
 * picture-level: a synthetic HEVCContext around caller-owned planes
 * ======================================================================================= */
typedef struct RefCtx {
    HEVCContext       s;
    HEVCLocalContext  lc;
    HEVCSPS           sps;
    HEVCPPS           pps;
    HEVCFrame         ref;
    AVFrame           frame, sao_frame;
    int              *zs_tab, *rs_to_ts, *ts_to_rs, *tile_id;
} RefCtx;

static void ctx_free(RefCtx *r)
{
    free(r->zs_tab); free(r->rs_to_ts); free(r->ts_to_rs); free(r->tile_id);
    free(r->s.filter_slice_edges); free(r->s.tab_slice_address);
    free(r->s.sao); free(r->s.deblock);
    free(r);
}

/* fills sps/pps exactly as hevc_ps.c derives them from the syntax (hevc_ps.c:2001-2011, 2545-2569),
 * single slice, single tile */
static RefCtx *ctx_new(const OhPicParams *p, uint8_t *const data[3], const ptrdiff_t stride[3])
{
    RefCtx *r = calloc(1, sizeof(*r));
    HEVCSPS *sps = &r->sps;
    HEVCPPS *pps = &r->pps;
    int ctbs, n, d;

    sps->width = p->width; sps->height = p->height;
    sps->bit_depth = p->bit_depth; sps->pixel_shift = p->bit_depth > 8;
    sps->chroma_array_type = p->chroma_format_idc; sps->chroma_format_idc = p->chroma_format_idc;
    sps->hshift[0] = sps->vshift[0] = 0;
    sps->hshift[1] = sps->hshift[2] = oh_hshift(p, 1);
    sps->vshift[1] = sps->vshift[2] = oh_vshift(p, 1);
    sps->log2_ctb_size = p->log2_ctb_size; sps->log2_min_cb_size = p->log2_min_cb_size;
    sps->log2_min_tb_size = p->log2_min_tb_size; sps->log2_min_pu_size = p->log2_min_pu_size;
    sps->ctb_width = oh_ctb_width(p); sps->ctb_height = oh_ctb_height(p);
    sps->ctb_size = sps->ctb_width * sps->ctb_height;
    sps->min_cb_width = p->width >> p->log2_min_cb_size; sps->min_cb_height = p->height >> p->log2_min_cb_size;
    sps->min_tb_width = p->width >> p->log2_min_tb_size; sps->min_tb_height = p->height >> p->log2_min_tb_size;
    sps->min_pu_width = p->width >> p->log2_min_pu_size; sps->min_pu_height = p->height >> p->log2_min_pu_size;
    sps->tb_mask = (1 << (p->log2_ctb_size - p->log2_min_tb_size)) - 1;
    sps->qp_bd_offset = 6 * (p->bit_depth - 8);
    sps->sao_enabled = (uint8_t)p->sao_enabled;
    sps->pcm_enabled_flag = p->pcm_loop_filter_disable;
    sps->pcm.loop_filter_disable_flag = (uint8_t)p->pcm_loop_filter_disable;
    sps->sps_strong_intra_smoothing_enable_flag = (uint8_t)p->strong_intra_smoothing;
    sps->spsRext.intra_smoothing_disabled_flag = (uint8_t)p->intra_smoothing_disabled;

    pps->cb_qp_offset = p->cb_qp_offset; pps->cr_qp_offset = p->cr_qp_offset;
    pps->transquant_bypass_enable_flag = (uint8_t)p->transquant_bypass_enable;
    pps->constrained_intra_pred_flag = (uint8_t)p->constrained_intra_pred;
    pps->loop_filter_across_tiles_enabled_flag = 1;
    ctbs = sps->ctb_size;
    r->rs_to_ts = malloc(sizeof(int) * (size_t)(ctbs + 1));
    r->ts_to_rs = malloc(sizeof(int) * (size_t)(ctbs + 1));
    r->tile_id  = calloc((size_t)(ctbs + 1), sizeof(int));
    for (int i = 0; i <= ctbs; i++) r->rs_to_ts[i] = r->ts_to_rs[i] = i;
    if (g_maps && g_maps->tiles_enabled) {
        /* tile scan (6.5.1): tiles in raster order of the tile grid, CTBs in raster order inside a tile; pps->tile_id is
         * indexed by the tile-scan address (hevc_ps.c:2151) */
        int max_id = 0, ts = 0;
        for (int i = 0; i < ctbs; i++) if (g_maps->tile_id[i] > max_id) max_id = g_maps->tile_id[i];
        for (int t = 0; t <= max_id; t++)
            for (int i = 0; i < ctbs; i++)
                if (g_maps->tile_id[i] == t) { r->rs_to_ts[i] = ts; r->ts_to_rs[ts] = i; r->tile_id[ts] = t; ts++; }
        pps->tiles_enabled_flag = 1;
        pps->loop_filter_across_tiles_enabled_flag = (uint8_t)g_maps->loop_filter_across_tiles;
    }
    pps->ctb_addr_rs_to_ts = r->rs_to_ts; pps->ctb_addr_ts_to_rs = r->ts_to_rs; pps->tile_id = r->tile_id;
    n = sps->tb_mask + 2; d = p->log2_ctb_size - p->log2_min_tb_size;
    r->zs_tab = malloc(sizeof(int) * (size_t)(n * n));
    pps->min_tb_addr_zs_tab = r->zs_tab;
    pps->min_tb_addr_zs = &r->zs_tab[n + 1];
    for (int y = 0; y < n; y++) { r->zs_tab[y * n] = -1; r->zs_tab[y] = -1; }
    for (int y = 0; y < sps->tb_mask + 1; y++)
        for (int x = 0; x < sps->tb_mask + 1; x++) {
            int tb_x = x >> d, tb_y = y >> d;
            int val = pps->ctb_addr_rs_to_ts[sps->ctb_width * tb_y + tb_x] << (d * 2);
            for (int i = 0; i < d; i++) {
                int m = 1 << i;
                val += (m & x ? m * m : 0) + (m & y ? 2 * m * m : 0);
            }
            pps->min_tb_addr_zs[y * n + x] = val;
        }

    for (int c = 0; c < 3; c++) {
        r->frame.data[c] = data[c];
        r->frame.linesize[c] = (int)stride[c];
    }
    r->s.sps = sps; r->s.pps = pps;
    r->s.HEVClc = &r->lc; r->s.HEVClcList[0] = &r->lc;
    r->s.frame = &r->frame; r->s.sao_frame = &r->sao_frame;
    r->ref.frame = &r->frame; r->s.ref = &r->ref;
    r->s.threads_type = 0;
    ff_hevc_dsp_init(&r->s.hevcdsp, p->bit_depth);
    ff_hevc_pred_init(&r->s.hpc, p->bit_depth);
    ff_videodsp_init(&r->s.vdsp, p->bit_depth);
    return r;
}

/* Runs the intra items of an OhFrame through the reference's s->hpc.intra_pred[] slots in list
 * order, adding the (already inverse-transformed) residual after each block like
 * hls_transform_unit does (hevc.c:1215-1417).  The candidate flags are derived HERE the way the
 * reference derives lc->na for a single-slice, single-tile picture (hevc.c:2592-2642 +
 * hevc_mvs.c:41-58) — they are NOT taken from OhIntra.avail, so the test also checks the
 * availability computation of whoever built the work list. */
API int ref_intra_picture(const OhFrame *f, uint8_t *const data[3], const ptrdiff_t stride[3],
                          const int16_t *residuals)
{
    const OhPicParams *p = &f->p;
    RefCtx *r = ctx_new(p, data, stride);
    HEVCContext *s = &r->s;
    HEVCLocalContext *lc = &r->lc;
    int ctb = 1 << p->log2_ctb_size;
    MvField *mvf = NULL;

    if (p->constrained_intra_pred && f->is_intra) {            /* IS_INTRA() reads s->ref->tab_mvf[].pred_flag (hevcpred_template.c:34-41) */
        size_t n = (size_t)r->sps.min_pu_width * r->sps.min_pu_height;
        mvf = calloc(n, sizeof(*mvf));
        for (size_t k = 0; k < n; k++) mvf[k].pred_flag = f->is_intra[k] ? PF_INTRA : PF_L0;
        r->ref.tab_mvf = mvf;
    }
    for (uint32_t i = 0; i < f->n_intra; i++) {
        const OhIntra *it = &f->intra[i];
        int c = it->c_idx, hs = oh_hshift(p, c), vs = oh_vshift(p, c);
        int x0 = it->x << hs, y0 = it->y << vs;                /* luma units, as hevc.c passes them */
        int n_h = (1 << it->log2_size) << hs, n_v = (1 << it->log2_size) << vs;
        int x_ctb = x0 & ~(ctb - 1), y_ctb = y0 & ~(ctb - 1);
        int x0b = x0 & (ctb - 1), y0b = y0 & (ctb - 1);
        int ctb_addr = (y_ctb >> p->log2_ctb_size) * r->sps.ctb_width + (x_ctb >> p->log2_ctb_size);
        /* hls_decode_neighbour (hevc.c:2592-2642) for the CTB's slice address and tile */
        {
            const int W = r->sps.ctb_width, tiles = g_maps && g_maps->tiles_enabled;
            const int in_slice = ctb_addr - (g_maps ? g_maps->slice_addr[ctb_addr] : 0);
#define TID(rs) (g_maps ? g_maps->tile_id[rs] : 0)
            const int tile_left = tiles && x_ctb > 0 && TID(ctb_addr) != TID(ctb_addr - 1);
            const int tile_up = tiles && y_ctb > 0 && TID(ctb_addr) != TID(ctb_addr - W);
            lc->ctb_left_flag     = x_ctb > 0 && in_slice > 0 && !tile_left;
            lc->ctb_up_flag       = y_ctb > 0 && in_slice >= W && !tile_up;
            lc->ctb_up_right_flag = y_ctb > 0 && in_slice + 1 >= W && TID(ctb_addr) == TID(ctb_addr + 1 - W);
            lc->ctb_up_left_flag  = x_ctb > 0 && y_ctb > 0 && in_slice - 1 >= W && TID(ctb_addr) == TID(ctb_addr - 1 - W);
            lc->end_of_tiles_x = p->width;
            if (tiles) {                                       /* right edge of the CTB's tile (column_width[], hevc.c:2609-2613) */
                int cx = x_ctb >> p->log2_ctb_size;
                while (cx + 1 < W && TID(ctb_addr - (x_ctb >> p->log2_ctb_size) + cx + 1) == TID(ctb_addr)) cx++;
                lc->end_of_tiles_x = (cx + 1) << p->log2_ctb_size < p->width ? (cx + 1) << p->log2_ctb_size : p->width;
            }
#undef TID
        }
        lc->end_of_tiles_y = y_ctb + ctb < p->height ? y_ctb + ctb : p->height;
        /* ff_hevc_set_neighbour_available(s, x0, y0, n_h, n_v) */
        lc->na.cand_up       = lc->ctb_up_flag || y0b;
        lc->na.cand_left     = lc->ctb_left_flag || x0b;
        lc->na.cand_up_left  = (!x0b && !y0b) ? lc->ctb_up_left_flag : lc->na.cand_left && lc->na.cand_up;
        lc->na.cand_up_right_sap = (x0b + n_h == ctb) ? lc->ctb_up_right_flag && !y0b : lc->na.cand_up;
        lc->na.cand_up_right = lc->na.cand_up_right_sap && (x0 + n_h) < lc->end_of_tiles_x;
        lc->na.cand_bottom_left = ((y0 + n_v) >= lc->end_of_tiles_y) ? 0 : lc->na.cand_left;
        lc->tu.intra_pred_mode = lc->tu.intra_pred_mode_c = it->mode;

        s->hpc.intra_pred[it->log2_size - 2](s, x0, y0, c);

        if (it->tu != OH_NO_COEFF) {
            const OhTu *tu = &f->tu[it->tu];
            uint8_t *dst = data[c] + (ptrdiff_t)tu->y * stride[c] + ((ptrdiff_t)tu->x << r->sps.pixel_shift);
            s->hevcdsp.transform_add[tu->log2_size - 2](dst, (int16_t *)(residuals + tu->coeff_off), stride[c]);
        }
    }
    free(mvf);
    ctx_free(r);
    return 0;
}

/* Runs the reference's in-loop filters over a whole picture in the reference's own order: the
 * CTU loop of hls_decode_entry (hevc.c:2666-2695) calls ff_hevc_hls_filters after every CTB and
 * ff_hevc_hls_filter for the last one.  `data` is filtered in place; sao planes are scratch with
 * the same geometry (the reference's s->sao_frame, hevc.c:369-385). */
API int ref_filter_picture(const OhFrame *f, uint8_t *const data[3], const ptrdiff_t stride[3],
                           uint8_t *const sao_data[3])
{
    const OhPicParams *p = &f->p;
    RefCtx *r = ctx_new(p, data, stride);
    HEVCContext *s = &r->s;
    int ctb = 1 << p->log2_ctb_size;
    int ctbs = r->sps.ctb_size;
    int pic_size_in_ctb = (r->sps.min_cb_width + 1) * (r->sps.min_cb_height + 1);

    for (int c = 0; c < 3; c++) {
        r->sao_frame.data[c] = sao_data[c];
        r->sao_frame.linesize[c] = (int)stride[c];
    }
    s->bs_width = p->width >> 2; s->bs_height = p->height >> 2;
    s->vertical_bs = (uint8_t *)f->vertical_bs; s->horizontal_bs = (uint8_t *)f->horizontal_bs;
    s->qp_y_tab = (int8_t *)f->qp_y_tab;
    s->is_pcm = (uint8_t *)f->is_pcm;
    s->deblock = calloc((size_t)ctbs, sizeof(*s->deblock));
    s->sao = calloc((size_t)ctbs, sizeof(*s->sao));
    s->filter_slice_edges = malloc((size_t)ctbs);
    s->tab_slice_address = calloc((size_t)pic_size_in_ctb, sizeof(*s->tab_slice_address));
    for (int i = 0; i < ctbs; i++) {
        s->deblock[i].beta_offset = f->deblock[i].beta_offset;
        s->deblock[i].tc_offset = f->deblock[i].tc_offset;
        s->filter_slice_edges[i] = g_maps ? g_maps->filter_slice_edges[i] : 1;
        s->tab_slice_address[i] = g_maps ? g_maps->slice_addr[i] : 0;          /* read per CTB through CTB(), hevc_filter.c:195 */
        if (f->sao)
            for (int c = 0; c < 3; c++) {
                for (int k = 0; k < 5; k++)
                    s->sao[i].offset_val[c][k] = f->sao[i].offset_val[c][k];
                s->sao[i].band_position[c] = f->sao[i].band_position[c];
                s->sao[i].eo_class[c] = f->sao[i].eo_class[c];
                s->sao[i].type_idx[c] = f->sao[i].type_idx[c];
            }
    }
    if (!p->deblock_enabled) {            /* the reference has no such switch: all-zero BS is the same */
        ctx_free(r);
        return -1;
    }
    /* in DECODING order (tile scan, hevc.c:2666-2669: ctb_addr_rs = ctb_addr_rs_to_ts^-1[ctb_addr_ts]), as the single-threaded CTU
     * loop does: the order of the calls shows in the output for 16x16 CTBs with subsampled chroma (OhFrame.sao_pending) */
    const int ctbw = (p->width + ctb - 1) / ctb, n_ctbs = ctbw * ((p->height + ctb - 1) / ctb);
    for (int ts = 0; ts < n_ctbs; ts++) {
        const int rs = r->ts_to_rs[ts], x_ctb = (rs % ctbw) * ctb, y_ctb = (rs / ctbw) * ctb;
        ff_hevc_hls_filters(s, x_ctb, y_ctb, ctb);                           /* hevc.c:2690 */
        if (x_ctb + ctb >= p->width && y_ctb + ctb >= p->height)
            ff_hevc_hls_filter(s, x_ctb, y_ctb, ctb);                        /* hevc.c:2693-2695 */
    }
    ctx_free(r);
    return 0;
}

/* ---------------- SHVC up-sampling slots (hevcdsp.h:98-123) ---------------- */
static void up_fill(const OhUpsample *u, struct HEVCWindow *w, struct UpsamplInf *inf)
{
    w->left_offset = u->win_left; w->right_offset = u->win_right; w->top_offset = u->win_top; w->bottom_offset = u->win_bottom;
    inf->addXLum = u->add_x_lum; inf->addYLum = u->add_y_lum; inf->scaleXLum = u->scale_x_lum; inf->scaleYLum = u->scale_y_lum;
    inf->addXCr = u->add_x_cr; inf->addYCr = u->add_y_cr; inf->scaleXCr = u->scale_x_cr; inf->scaleYCr = u->scale_y_cr;
    inf->idx = u->idx;
}
/* which: 0 luma_h, 1 cr_h (dst int16) ; 2 luma_v, 3 cr_v (src int16) */
API void ref_up_block_h(int bd, int cr, int variant, int16_t *dst, ptrdiff_t dststride, uint8_t *src, ptrdiff_t srcstride,
                        int x_el, int x_bl, int block_w, int block_h, int width_el, const OhUpsample *u)
{
    struct HEVCWindow w; struct UpsamplInf inf;
    tables(bd); up_fill(u, &w, &inf);
    (cr ? g_dsp[bd].upsample_filter_block_cr_h : g_dsp[bd].upsample_filter_block_luma_h)[variant](dst, dststride, src, srcstride, x_el, x_bl,
                                                                                            block_w, block_h, width_el, &w, &inf);
}
API void ref_up_block_v(int bd, int cr, int variant, uint8_t *dst, ptrdiff_t dststride, int16_t *src, ptrdiff_t srcstride,
                        int y_bl, int x_el, int y_el, int block_w, int block_h, int width_el, int height_el, const OhUpsample *u)
{
    struct HEVCWindow w; struct UpsamplInf inf;
    tables(bd); up_fill(u, &w, &inf);
    (cr ? g_dsp[bd].upsample_filter_block_cr_v : g_dsp[bd].upsample_filter_block_luma_v)[variant](dst, dststride, src, srcstride, y_bl, x_el, y_el,
                                                                                            block_w, block_h, width_el, height_el, &w, &inf);
}
/* the whole-picture slot (hevc.c:3241); planes/strides in bytes, 8-bit */
API void ref_up_frame(uint8_t *const el[3], const int el_stride[3], int el_w, int el_h,
                      uint8_t *const bl[3], const int bl_stride[3], int bl_w, int bl_h, const OhUpsample *u)
{
    struct HEVCWindow w; struct UpsamplInf inf;
    AVFrame fe, fb;
    short *buf[3];
    size_t n = (size_t)el_w * (size_t)(el_h > bl_h ? el_h : bl_h);
    tables(8); up_fill(u, &w, &inf);
    memset(&fe, 0, sizeof(fe)); memset(&fb, 0, sizeof(fb));
    for (int c = 0; c < 3; c++) {
        fe.data[c] = el[c]; fe.linesize[c] = el_stride[c];
        fb.data[c] = bl[c]; fb.linesize[c] = bl_stride[c];
        buf[c] = (short *)malloc(n * sizeof(short));
    }
    fe.coded_width = el_w; fe.coded_height = el_h; fb.coded_width = bl_w; fb.coded_height = bl_h;
    g_dsp[8].upsample_base_layer_frame(&fe, &fb, buf, &w, &inf, 1);
    for (int c = 0; c < 3; c++) free(buf[c]);
}

/* ---------------- whole picture through the reference's kernels ----------------
 * Passes 1-5 of one work list with the reference's own slots and drivers: PUs through put_hevc_{q,e}pel* with
 * emulated_edge_mc (replay_slots.inc issues the calls the way hevc.c:1641-1949 does), inverse transforms through
 * idct* / transform_skip / rdpcm, inter and PCM blocks added at once, intra blocks through hpc.intra_pred[] with the
 * residual added after each block (ref_intra_picture, blocks re-sorted into decode order here), then ff_hevc_hls_filters.
 * Used to pin the oracle at picture level incl. the MC driver logic, and as bench.py's CPU baseline ("reference"). */
/* decode (z-scan) order of the intra blocks: CTB raster order, then Morton order of the block's luma position inside the
 * CTB, then plane; the 4:2:0 chroma blocks of four 4x4 luma blocks are coded after the fourth one (hevc.c:1395) */
static const OhPicParams *g_sort_p;
static uint64_t decode_key(const OhIntra *it)
{
    const OhPicParams *p = g_sort_p;
    int hs = oh_hshift(p, it->c_idx), vs = oh_vshift(p, it->c_idx), ctb = 1 << p->log2_ctb_size;
    int x = it->x << hs, y = it->y << vs, xi = x & (ctb - 1), yi = y & (ctb - 1);
    uint64_t m = 0;
    if (it->c_idx && p->chroma_format_idc == 1 && it->log2_size == 2) { xi += 4; yi += 4; }
    for (int b = 0; b < 7; b++)
        m |= (uint64_t)((xi >> b) & 1) << (2 * b) | (uint64_t)((yi >> b) & 1) << (2 * b + 1);
    return ((uint64_t)(y >> p->log2_ctb_size) << 44) | ((uint64_t)(x >> p->log2_ctb_size) << 28) | (m << 4) | it->c_idx;
}
static int decode_cmp(const void *a, const void *b)
{
    uint64_t ka = decode_key((const OhIntra *)a), kb = decode_key((const OhIntra *)b);
    return ka < kb ? -1 : ka > kb;
}

API int ref_bs_derive(const OhPicParams *p, const OhBsInputs *in, uint8_t *vbs, uint8_t *hbs);

API int ref_frame(const OhFrame *f_in, uint8_t *const cur[3], const ptrdiff_t cur_stride[3],
                  uint8_t *const refs[][3], int n_refs, const ptrdiff_t ref_stride[3], uint8_t *const sao_scratch[3])
{
    OhFrame fcopy = *f_in;
    const OhFrame *f = &fcopy;
    const OhPicParams *p = &f->p;
    ReplayCtx *k = calloc(1, sizeof(*k));
    int16_t *pool = malloc(sizeof(int16_t) * (size_t)(f->n_coeff ? f->n_coeff : 1));
    OhIntra *sorted = malloc(sizeof(OhIntra) * (size_t)(f->n_intra ? f->n_intra : 1));
    int rc = 0;
    memcpy(sorted, f_in->intra, sizeof(OhIntra) * (size_t)f->n_intra);      /* the work list is in wavefront order */
    g_sort_p = p;
    qsort(sorted, f->n_intra, sizeof(OhIntra), decode_cmp);
    fcopy.intra = sorted;
    tables(p->bit_depth);
    k->f = f; k->bpp = p->bit_depth > 8 ? 2 : 1;
    k->d = g_dsp[p->bit_depth]; k->hp = g_pred[p->bit_depth]; k->v = g_vdsp[p->bit_depth];
    for (int c = 0; c < 3; c++) { k->cur[c] = cur[c]; k->cur_ls[c] = (int)cur_stride[c]; k->ref_ls[c] = (int)ref_stride[c]; }
    for (int s = 0; s < n_refs && s < OH_MAX_REFS; s++)
        for (int c = 0; c < 3; c++) k->ref[s][c] = refs[s][c];
    for (uint32_t i = 0; i < f->n_pu; i++)
        replay_pu(k, &f->pu[i]);
    memcpy(pool, f->coeffs, sizeof(int16_t) * (size_t)f->n_coeff);
    for (uint32_t i = 0; i < f->n_tu; i++)
        if (f->tu[i].kind != OH_TU_PCM)
            replay_inverse(k, &f->tu[i], pool + f->tu[i].coeff_off);
    for (uint32_t i = 0; i < f->n_tu; i++) {
        const OhTu *tu = &f->tu[i];
        if (tu->kind == OH_TU_PCM)
            replay_pcm(k, tu);
        else if (tu->flags & OH_TUF_ADD_NOW)
            k->d.transform_add[tu->log2_size - 2](cur[tu->c_idx] + (ptrdiff_t)tu->y * cur_stride[tu->c_idx] + (ptrdiff_t)tu->x * k->bpp,
                                                  pool + tu->coeff_off, cur_stride[tu->c_idx]);
    }
    if (f->n_intra)
        rc = ref_intra_picture(f, cur, cur_stride, pool);
    uint8_t *dv = NULL, *dh = NULL;
    if (!rc && p->deblock_enabled && f->bs_in) {           /* the strengths come from the reference's own derivation (ref_bs_derive below) */
        dv = malloc(oh_bs_size(p)); dh = malloc(oh_bs_size(p));
        rc = ref_bs_derive(p, f->bs_in, dv, dh);
        fcopy.vertical_bs = dv; fcopy.horizontal_bs = dh; fcopy.bs_size = oh_bs_size(p);
    }
    if (!rc && (p->deblock_enabled || p->sao_enabled))
        rc = ref_filter_picture(f, cur, cur_stride, sao_scratch);
    free(dv); free(dh);
    free(pool);
    free(sorted);
    free(k);
    return rc;
}

/* ---------------- SHVC: the PU-driven block path (ACTIVE_PU_UPSAMPLING, hevc.h:117) ----------------
 * ff_upsample_block (hevc_filter.c:1370-1426) for every CTB of the enhancement-layer picture: upsample_block_luma / _mc
 * with emulated_edge_up_h/v and the block slots.  The base-layer planes must carry an edge border like the reference's
 * frames (the emulation WRITES into it).  Motion fields are left empty (pred_flag 0: ff_upscale_mv_block only clears). */
API int ref_up_blocks(uint8_t *const el[3], const int el_stride[3], int el_w, int el_h,
                      uint8_t *const bl[3], const int bl_stride[3], int bl_w, int bl_h, const OhUpsample *u, int log2_ctb)
{
    HEVCContext *s = calloc(1, sizeof(*s));
    HEVCSPS *sps = calloc(1, sizeof(*sps));
    HEVCVPS *vps = calloc(1, sizeof(*vps));
    HEVCLocalContext *lc = calloc(1, sizeof(*lc));
    HEVCFrame *blf = calloc(1, sizeof(*blf)), *ilr = calloc(1, sizeof(*ilr)), *ref0 = calloc(1, sizeof(*ref0));
    AVFrame *fb = calloc(1, sizeof(*fb)), *fe = calloc(1, sizeof(*fe));
    struct HEVCWindow w;
    int ctb = 1 << log2_ctb;
    if (!s || !sps || !vps || !lc || !blf || !ilr || !ref0 || !fb || !fe)
        return -1;
    up_fill(u, &w, &s->up_filter_inf);
    sps->log2_ctb_size = log2_ctb; sps->width = el_w; sps->height = el_h;
    sps->ctb_width = (el_w + ctb - 1) >> log2_ctb; sps->ctb_height = (el_h + ctb - 1) >> log2_ctb;
    sps->log2_min_pu_size = 2;
    sps->scaled_ref_layer_window[0] = w;
    s->sps = sps; s->vps = vps; s->nuh_layer_id = 1;            /* m_refLayerId[1][0] == 0 */
    s->HEVClc = lc; s->HEVClcList[0] = lc;
    s->sh.slice_type = P_SLICE;
    for (int c = 0; c < 3; c++) {
        fb->data[c] = bl[c]; fb->linesize[c] = bl_stride[c];
        fe->data[c] = el[c]; fe->linesize[c] = el_stride[c];
    }
    fb->coded_width = bl_w; fb->coded_height = bl_h; fe->coded_width = el_w; fe->coded_height = el_h;
    blf->frame = fb; ref0->frame = fe;
    blf->tab_mvf = calloc((size_t)(bl_w >> 2) * (bl_h >> 2) + 64, sizeof(MvField));
    ilr->tab_mvf = calloc((size_t)(el_w >> 2) * (el_h >> 2) + 64, sizeof(MvField));
    s->BL_frame = blf; s->inter_layer_ref = ilr;
    s->is_upsampled = calloc((size_t)sps->ctb_width * sps->ctb_height, 1);
    ff_hevc_dsp_init(&s->hevcdsp, 8);
    ff_videodsp_init(&s->vdsp, 8);
    for (int y0 = 0; y0 < el_h; y0 += ctb)
        for (int x0 = 0; x0 < el_w; x0 += ctb)
            ff_upsample_block(s, ref0, x0, y0, ctb, ctb);
    int missing = 0;
    for (int i = 0; i < sps->ctb_width * sps->ctb_height; i++) missing += !s->is_upsampled[i];
    free(s->is_upsampled); free(blf->tab_mvf); free(ilr->tab_mvf);
    free(fb); free(fe); free(blf); free(ilr); free(ref0); free(lc); free(vps); free(sps); free(s);
    return missing;
}

/* =========================================================================================
 * boundary strengths: the reference's own ff_hevc_deblocking_boundary_strengths (hevc_filter.c:805-941), called once per
 * block in raster order of the block origins with the per-CTB state hls_decode_neighbour leaves in lc / s->sh
 * (hevc.c:2636-2637).  MvField must have the layout OhMvField mirrors (TEST_MV_POC build).
 * ======================================================================================= */
typedef char oh_mvfield_layout_check[(sizeof(MvField) == sizeof(OhMvField)) ? 1 : -1];

API int ref_bs_derive(const OhPicParams *p, const OhBsInputs *in, uint8_t *vbs, uint8_t *hbs)
{
    uint8_t *none[3] = { NULL, NULL, NULL };
    ptrdiff_t nostride[3] = { 0, 0, 0 };
    RefCtx *r = ctx_new(p, none, nostride);
    HEVCContext *s = &r->s;
    HEVCLocalContext *lc = &r->lc;
    const int ltu = p->log2_min_tb_size, lcs = p->log2_ctb_size, mtw = p->width >> ltu, mth = p->height >> ltu;

    memset(vbs, 0, oh_bs_size(p));
    memset(hbs, 0, oh_bs_size(p));
    s->bs_width = p->width >> 2; s->bs_height = p->height >> 2;
    s->vertical_bs = vbs; s->horizontal_bs = hbs;
    s->cbf_luma = (uint8_t *)in->cbf_luma;
    r->ref.tab_mvf = (MvField *)in->mvf;
    r->pps.loop_filter_across_tiles_enabled_flag = (uint8_t)in->loop_filter_across_tiles;
    for (int ty = 0; ty < mth; ty++)
        for (int tx = 0; tx < mtw; tx++) {
            int log2 = in->call_log2[ty * mtw + tx], x0 = tx << ltu, y0 = ty << ltu;
            if (!log2 || (x0 & ((1 << log2) - 1)) || (y0 & ((1 << log2) - 1)))
                continue;
            int flags = in->ctb_flags[(y0 >> lcs) * r->sps.ctb_width + (x0 >> lcs)];
            lc->slice_or_tiles_up_boundary = flags & 3;
            lc->slice_or_tiles_left_boundary = (flags >> 2) & 3;
            s->sh.slice_loop_filter_across_slices_enabled_flag = (flags & OH_BSF_ACROSS_SLICES) != 0;
            ff_hevc_deblocking_boundary_strengths(s, x0, y0, log2);
        }
    ctx_free(r);
    return 0;
}
