/*
 * replay_driver.c — TEST INFRASTRUCTURE.  Plays the role of the reference's CTU loop towards the
 * function-pointer tables: for a given work list it issues the slot calls that hevc.c would issue
 * (hls_prediction_unit -> luma_mc_* / chroma_mc_*, hevc.c:1641-1949 and :2103-2153;
 * hls_transform_unit -> intra_pred[] + residual, hevc.c:1215-1417 and hevc_cabac.c:1868-1949),
 * with the reference's argument conventions (raw plane pointers, byte strides, edge emulation
 * through vdsp.emulated_edge_mc into an 80-sample-stride buffer).  The tables filled by
 * ff_hevcdsp_init_hip() record those calls into a second recorder; the test then checks that
 * the re-recorded work list reconstructs the same picture.
 */
#include <stdlib.h>
#include <string.h>
#include "../include/ohevc_tables.h"

static const OhIntra *g_item;
static void accessor(struct HEVCContext *s, int x0, int y0, int c_idx, int log2, int *mode, int *avail)
{
    (void)s; (void)x0; (void)y0; (void)c_idx; (void)log2;
    *mode = g_item->mode; *avail = g_item->avail;
}

#include "../oracle/replay_slots.inc"

/* returns the number of slot calls the tables could not translate, or -1 */
int replay_through_tables(const OhFrame *f, OhRecorder *out, uint8_t *const cur[3], const int cur_ls[3],
                          uint8_t *const refs[][3], int n_refs, const int ref_ls[3])
{
    ReplayCtx *k = calloc(1, sizeof(*k));
    const OhPicParams *p = &f->p;
    k->f = f; k->bpp = p->bit_depth > 8 ? 2 : 1;
    ff_hevcdsp_init_hip(&k->d, p->bit_depth);
    ff_hevcpred_init_hip(&k->hp, p->bit_depth);
    ff_videodsp_init_hip(&k->v, p->bit_depth);
    for (int c = 0; c < 3; c++) { k->cur[c] = cur[c]; k->cur_ls[c] = cur_ls[c]; k->ref_ls[c] = ref_ls[c]; }
    oh_rec_begin(out, f->cur_pic, f->ref_pics, OH_MAX_REFS);
    oh_tables_bind(out, cur, cur_ls);
    oh_tables_set_intra_accessor(accessor);
    for (int s = 0; s < n_refs; s++) {
        for (int c = 0; c < 3; c++) k->ref[s][c] = refs[s][c];
        oh_tables_bind_ref(s, refs[s], ref_ls);
    }
    /* side arrays are plain copies in the integration too */
    memcpy(oh_rec_vertical_bs(out), f->vertical_bs, f->bs_size);
    memcpy(oh_rec_horizontal_bs(out), f->horizontal_bs, f->bs_size);
    memcpy(oh_rec_qp_y_tab(out), f->qp_y_tab, oh_qp_tab_size(p));
    if (f->is_pcm) memcpy(oh_rec_is_pcm(out), f->is_pcm, (size_t)oh_min_pu_width(p) * oh_min_pu_height(p));
    memcpy(oh_rec_deblock(out), f->deblock, sizeof(OhDeblockCtb) * (size_t)oh_ctb_width(p) * oh_ctb_height(p));
    if (f->sao) memcpy(oh_rec_sao(out), f->sao, sizeof(OhSaoCtb) * (size_t)oh_ctb_width(p) * oh_ctb_height(p));

    for (uint32_t i = 0; i < f->n_pu; i++)
        replay_pu(k, &f->pu[i]);
    uint8_t *tu_done = calloc(f->n_tu + 1, 1);
    for (uint32_t i = 0; i < f->n_intra; i++) {           /* schedule order is a valid decode order */
        const OhIntra *it = &f->intra[i];
        g_item = it;
        k->hp.intra_pred[it->log2_size - 2](NULL, it->x << oh_hshift(p, it->c_idx), it->y << oh_vshift(p, it->c_idx), it->c_idx);
        if (it->tu != OH_NO_COEFF) { replay_residual(k, &f->tu[it->tu]); tu_done[it->tu] = 1; }
    }
    for (uint32_t i = 0; i < f->n_tu; i++) {
        const OhTu *tu = &f->tu[i];
        if (tu_done[i])
            continue;
        if (tu->kind == OH_TU_PCM)                        /* hls_pcm_sample (hevc.c:1587-1640): put_pcm reads the samples from the bitstream */
            replay_pcm(k, tu);
        else
            replay_residual(k, tu);
    }
    int bad = oh_tables_finish();
    free(tu_done);
    free(k);
    return bad;
}
