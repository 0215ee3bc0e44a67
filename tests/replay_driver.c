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

#define EDGE_STRIDE 80            /* EDGE_EMU_BUFFER_STRIDE, hevc.h:120 */

static const OhIntra *g_item;
static void accessor(struct HEVCContext *s, int x0, int y0, int c_idx, int log2, int *mode, int *avail)
{
    (void)s; (void)x0; (void)y0; (void)c_idx; (void)log2;
    *mode = g_item->mode; *avail = g_item->avail;
}

static int pel_idx(int w)
{
    switch (w) { case 2: return 0; case 4: return 1; case 6: return 2; case 8: return 3; case 12: return 4;
                 case 16: return 5; case 24: return 6; case 32: return 7; case 48: return 8; default: return 9; }
}

typedef struct Ctx {
    const OhFrame *f; HEVCDSPContext d; HEVCPredContext hp; VideoDSPContext v;
    uint8_t *cur[3]; int cur_ls[3]; uint8_t *ref[OH_MAX_REFS][3]; int ref_ls[3];
    int bpp; uint8_t emu[2][(64 + 7) * EDGE_STRIDE * 2];
} Ctx;

/* source pointer of one list for one plane, with the reference's edge test (hevc.c:1660-1675, 1816-1832) */
static uint8_t *mc_src(Ctx *k, int slot, int c, int x_off, int y_off, int bw, int bh, int which_emu, ptrdiff_t *stride)
{
    const OhPicParams *p = &k->f->p;
    int pw = p->width >> oh_hshift(p, c), ph = p->height >> oh_vshift(p, c);
    int before = c ? 1 : 3, after = c ? 2 : 4, extra = before + after;
    ptrdiff_t ls = k->ref_ls[c];
    uint8_t *src = k->ref[slot][c] + (ptrdiff_t)y_off * ls + (ptrdiff_t)x_off * k->bpp;
    *stride = ls;
    if (x_off < before || y_off < after || x_off >= pw - bw - after || y_off >= ph - bh - after) {
        ptrdiff_t es = EDGE_STRIDE * k->bpp;
        k->v.emulated_edge_mc(k->emu[which_emu], src - before * ls - before * k->bpp, es, ls, bw + extra, bh + extra,
                              x_off - before, y_off - before, pw, ph);
        src = k->emu[which_emu] + before * es + before * k->bpp;
        *stride = es;
    }
    return src;
}

static void replay_pu(Ctx *k, const OhPu *pu)
{
    const OhFrame *f = k->f;
    const OhPicParams *p = &f->p;
    const OhWeights *wp = pu->wp == OH_NO_WP ? NULL : &f->wp[pu->wp];
    int16_t tmp[64 * 64];
    int bi = pu->ref[0] != OH_NO_REF && pu->ref[1] != OH_NO_REF;
    for (int c = 0; c < (p->chroma_format_idc ? 3 : 1); c++) {
        int hs = oh_hshift(p, c), vs = oh_vshift(p, c);
        int bx = pu->x >> hs, by = pu->y >> vs, bw = pu->w >> hs, bh = pu->h >> vs, idx = pel_idx(bw);
        uint8_t *dst = k->cur[c] + (ptrdiff_t)by * k->cur_ls[c] + (ptrdiff_t)bx * k->bpp;
        int denom = wp ? wp->log2_denom[c ? 1 : 0] : 0;
        int mx[2], my[2], xo[2], yo[2];
        for (int l = 0; l < 2; l++) {
            int mvx = pu->mv[l][0], mvy = pu->mv[l][1];
            if (c == 0) { mx[l] = mvx & 3; my[l] = mvy & 3; xo[l] = bx + (mvx >> 2); yo[l] = by + (mvy >> 2); }
            else {
                mx[l] = mvx & ((1 << (2 + hs)) - 1); my[l] = mvy & ((1 << (2 + vs)) - 1);
                xo[l] = bx + (mvx >> (2 + hs)); yo[l] = by + (mvy >> (2 + vs));
            }
        }
        ptrdiff_t s0, s1;
        if (bi) {
            uint8_t *src0 = mc_src(k, pu->ref[0], c, xo[0], yo[0], bw, bh, 0, &s0);
            uint8_t *src1 = mc_src(k, pu->ref[1], c, xo[1], yo[1], bw, bh, 1, &s1);
            intptr_t fx0 = c ? mx[0] << (1 - hs) : mx[0], fy0 = c ? my[0] << (1 - vs) : my[0];
            intptr_t fx1 = c ? mx[1] << (1 - hs) : mx[1], fy1 = c ? my[1] << (1 - vs) : my[1];
            if (c == 0) {
                k->d.put_hevc_qpel[idx][!!my[0]][!!mx[0]](tmp, 64, src0, s0, bh, fx0, fy0, bw);
                if (!wp) k->d.put_hevc_qpel_bi[idx][!!my[1]][!!mx[1]](dst, k->cur_ls[c], src1, s1, tmp, 64, bh, fx1, fy1, bw);
                else k->d.put_hevc_qpel_bi_w[idx][!!my[1]][!!mx[1]](dst, k->cur_ls[c], src1, s1, tmp, 64, bh, denom, wp->w[0][c], wp->w[1][c],
                                                                    wp->o[0][c], wp->o[1][c], fx1, fy1, bw);
            } else {
                k->d.put_hevc_epel[idx][!!my[0]][!!mx[0]](tmp, 64, src0, s0, bh, fx0, fy0, bw);
                if (!wp) k->d.put_hevc_epel_bi[idx][!!my[1]][!!mx[1]](dst, k->cur_ls[c], src1, s1, tmp, 64, bh, fx1, fy1, bw);
                else k->d.put_hevc_epel_bi_w[idx][!!my[1]][!!mx[1]](dst, k->cur_ls[c], src1, s1, tmp, 64, bh, denom, wp->w[0][c], wp->w[1][c],
                                                                    wp->o[0][c], wp->o[1][c], fx1, fy1, bw);
            }
        } else {
            int l = pu->ref[0] != OH_NO_REF ? 0 : 1;
            uint8_t *src = mc_src(k, pu->ref[l], c, xo[l], yo[l], bw, bh, 0, &s0);
            intptr_t fx = c ? mx[l] << (1 - hs) : mx[l], fy = c ? my[l] << (1 - vs) : my[l];
            if (c == 0) {
                if (!wp) k->d.put_hevc_qpel_uni[idx][!!my[l]][!!mx[l]](dst, k->cur_ls[c], src, s0, bh, fx, fy, bw);
                else k->d.put_hevc_qpel_uni_w[idx][!!my[l]][!!mx[l]](dst, k->cur_ls[c], src, s0, bh, denom, wp->w[l][c], wp->o[l][c], fx, fy, bw);
            } else {
                if (!wp) k->d.put_hevc_epel_uni[idx][!!my[l]][!!mx[l]](dst, k->cur_ls[c], src, s0, bh, fx, fy, bw);
                else k->d.put_hevc_epel_uni_w[idx][!!my[l]][!!mx[l]](dst, k->cur_ls[c], src, s0, bh, denom, wp->w[l][c], wp->o[l][c], fx, fy, bw);
            }
        }
    }
}

/* residual of one TU the way ff_hevc_hls_residual_coding ends (hevc_cabac.c:1868-1949) */
static void replay_residual(Ctx *k, const OhTu *tu)
{
    int16_t buf[32 * 32];
    int n = 1 << tu->log2_size;
    memcpy(buf, k->f->coeffs + tu->coeff_off, sizeof(int16_t) * (size_t)(n * n));
    switch (tu->kind) {
    case OH_TU_BYPASS:
        if (tu->flags & OH_TUF_RDPCM) k->d.transform_rdpcm(buf, tu->log2_size, !!(tu->flags & OH_TUF_RDPCM_VER));
        break;
    case OH_TU_SKIP:
        if (tu->flags & OH_TUF_ROTATE)                 /* done by the caller in the reference, :1879-1882 */
            for (int i = 0; i < 8; i++) { int16_t t = buf[i]; buf[i] = buf[15 - i]; buf[15 - i] = t; }
        k->d.transform_skip(buf, tu->log2_size);
        if (tu->flags & OH_TUF_RDPCM) k->d.transform_rdpcm(buf, tu->log2_size, !!(tu->flags & OH_TUF_RDPCM_VER));
        break;
    case OH_TU_DST4: k->d.idct_4x4_luma(buf); break;
    default: {
        int only_dc = 1;
        for (int i = 1; i < n * n; i++) if (buf[i]) { only_dc = 0; break; }
        if (only_dc) k->d.idct_dc[tu->log2_size - 2](buf); else k->d.idct[tu->log2_size - 2](buf, n);
        break;
    }
    }
    uint8_t *dst = k->cur[tu->c_idx] + (ptrdiff_t)tu->y * k->cur_ls[tu->c_idx] + (ptrdiff_t)tu->x * k->bpp;
    k->d.transform_add[tu->log2_size - 2](dst, buf, k->cur_ls[tu->c_idx]);
}

/* returns the number of slot calls the tables could not translate, or -1 */
int replay_through_tables(const OhFrame *f, OhRecorder *out, uint8_t *const cur[3], const int cur_ls[3],
                          uint8_t *const refs[][3], int n_refs, const int ref_ls[3])
{
    Ctx *k = calloc(1, sizeof(*k));
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
        if (tu->kind == OH_TU_PCM) {                      /* hls_pcm_sample (hevc.c:1587-1640): put_pcm reads the samples from the bitstream */
            int n = 1 << tu->log2_size, bd = p->bit_depth, bit = 0;
            uint8_t bits[32 * 32 * 2 + 8];
            struct GetBitContext gb;
            memset(bits, 0, sizeof(bits));
            for (int i = 0; i < n * n; i++)
                for (int b = bd - 1; b >= 0; b--, bit++)
                    bits[bit >> 3] |= (uint8_t)(((f->coeffs[tu->coeff_off + i] >> b) & 1) << (7 - (bit & 7)));
            gb.buffer = bits; gb.buffer_end = bits + sizeof(bits); gb.index = 0; gb.size_in_bits = bit; gb.size_in_bits_plus8 = bit + 8;
            k->d.put_pcm(k->cur[tu->c_idx] + (ptrdiff_t)tu->y * k->cur_ls[tu->c_idx] + (ptrdiff_t)tu->x * k->bpp, k->cur_ls[tu->c_idx],
                         n, n, &gb, bd);
        } else
            replay_residual(k, tu);
    }
    int bad = oh_tables_finish();
    free(tu_done);
    free(k);
    return bad;
}
