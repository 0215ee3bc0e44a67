/*
 * ref_hooked_unit.c — TEST INFRASTRUCTURE ONLY: the first real run of INTEGRATION.md.
 *
 * The reference's hevc.c compiled as it lies in /root/reference (the #include at the bottom; nothing copied) with the three edits
 * INTEGRATION.md asks a maintainer for, made with the preprocessor instead of an editor:
 *   1. the arch-hook calls: ff_hevc_dsp_init / ff_hevc_pred_init / ff_videodsp_init are followed by this repository's
 *      ff_hevcdsp_init_hip / ff_hevcpred_init_hip / ff_videodsp_init_hip (openhevc_amd/csrc/tables.c) — the RECORDING slots;
 *   2. per picture: after ff_hevc_frame_rps() the recorder is bound to the picture and to its reference pictures (picture id =
 *      index of the HEVCFrame in s->DPB);
 *   3. the reference's in-loop filter driver calls (ff_hevc_hls_filters / ff_hevc_hls_filter) are dropped: passes 4-5 run from
 *      the arrays the CTU loop leaves behind, which ref_hooked_finish() copies into the recorder at the end of the access unit.
 * Everything else — NAL parsing, parameter sets, CABAC, MV derivation, de-quantisation, the CTU loop with its per-block slot
 * calls — is the reference's own code running as it always does.  The library (oracle/_ref/libopenhevc_hooked.so) decodes a
 * stream into WORK LISTS; tests/test_streams.py runs them through the checker (and -m gpu tests through the engine) and compares
 * the pictures with what the unmodified reference decoder (libopenhevc_ref.so) outputs for the same stream.
 *
 * Compiled a second time with -DOH_WITH_ENGINE and linked against openhevc_amd/libohevc_hip.so this unit is the decoder half of
 * the DROP-IN LIBRARY (oracle/_ref/libopenhevc_hip.so, the wrapper half is ref_wrapper_hip_unit.c): the AVCodec's decode callback
 * is followed by finish + oh_frame_submit (every picture's work list goes to the MI355X engine as soon as its access unit is
 * parsed), the decoder's own SEI picture-hash check takes its digests from the GPU (oh_pics_md5), and the wrapper's
 * libOpenHevcGetOutput / GetOutputCpy get the picture's samples out of HBM first — the 18 libOpenHevc* entry points with the
 * engine inside, which a caller of openHevcWrapper.h uses exactly like the reference's library.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "libavcodec/hevc.h"
#include "../include/ohevc_recorder.h"

/* the hooks and the binding API of openhevc_amd/csrc/tables.c (include/ohevc_tables.h re-declares the table types, which would
 * clash with the reference's own headers here: the three hooks are declared with untyped table pointers) */
void ff_hevcdsp_init_hip(void *c, const int bit_depth);
void ff_hevcpred_init_hip(void *c, const int bit_depth);
void ff_videodsp_init_hip(void *c, int bit_depth);
typedef void (*oh_intra_accessor)(struct HEVCContext *s, int x0, int y0, int c_idx, int log2_size, int *mode, int *avail);
void oh_tables_bind(OhRecorder *rec, uint8_t *const cur_data[3], const int cur_linesize[3]);
void oh_tables_bind_ref(int slot, uint8_t *const data[3], const int linesize[3]);
void oh_tables_set_intra_accessor(oh_intra_accessor fn);
int  oh_tables_finish(void);
void oh_tables_untranslated_by_family(int out[8]);

/* per decoding THREAD: the main thread (no threads, or slice / wavefront threads: it starts and ends every picture), or each of the
 * reference's frame threads (pthread_frame.c: a worker keeps its own HEVCContext and decodes whole pictures) */
/* one state per LAYER decoder: the wrapper runs the base-layer and the enhancement-layer decoder of an SHVC stream one after the other
 * on the same thread (openHevcWrapper.c:112-133), each with its own picture geometry and recorder */
static __thread struct HookState {
    HEVCContext *s;                   /* the context whose picture is being recorded on this thread */
    OhRecorder *rec;
    OhPicParams p;
    int open;                         /* a picture was started and not finished yet */
    int cur_id, n_refs, ref_ids[OH_MAX_REFS];
    int untranslated;
    int scaling_on;
    OhScalingList scaling;
    int last_engine_pic;              /* engine build: the engine picture the last submitted work list reconstructs */
    unsigned long long seq;           /* engine build: position of the picture in decode order (hand-over order) */
    /* engine build: the host buffers of the picture and of its references as they were when the picture started (an inter-layer
     * reference is released as soon as the last CTB is parsed, hevc.c:3471-3474 — before the hand-over) */
    const uint8_t *cur_base, *ref_base[OH_MAX_REFS];
    int cur_ls, ref_ls[OH_MAX_REFS];
    int ilr_slot;                     /* position of the inter-layer reference picture in the reference list, -1: none */
    const uint8_t *bl_base;           /* the base layer's picture it is the up-sampled version of (host buffer; DPB slot of the base layer's decoder) */
    int bl_id;
    OhUpsample up;
} HH[2];
static __thread int h_idx;            /* which of the two is current: set where a picture starts (and by the engine build's decode callback) */
#define H HH[h_idx]

static int g_bs_from_motion;          /* ref_hooked_bs_from_motion(): the work lists carry the boundary-strength INPUTS */
static __thread uint8_t *g_bs_call;   /* per min-TB cell: log2 size ff_hevc_deblocking_boundary_strengths was called with there, 0 = never */
static __thread size_t g_bs_call_n;
#ifdef OH_WITH_ENGINE
static unsigned long long next_picture_seq(void);
#endif

/* INTEGRATION.md §3 */
static void intra_from_hevc(struct HEVCContext *s, int x0, int y0, int c_idx, int log2_size, int *mode, int *avail)
{
    HEVCLocalContext *lc = s->HEVClc;
    int hshift = s->sps->hshift[c_idx], vshift = s->sps->vshift[c_idx];
    int size_in_tbs_h = ((1 << log2_size) << hshift) >> s->sps->log2_min_tb_size;
    int size_in_tbs_v = ((1 << log2_size) << vshift) >> s->sps->log2_min_tb_size;
    int x_tb = (x0 >> s->sps->log2_min_tb_size) & s->sps->tb_mask, y_tb = (y0 >> s->sps->log2_min_tb_size) & s->sps->tb_mask;
#define ZS(x, y) s->pps->min_tb_addr_zs[(y) * (s->sps->tb_mask + 2) + (x)]
    int cur = ZS(x_tb, y_tb);
    *mode  = c_idx ? lc->tu.intra_pred_mode_c : lc->tu.intra_pred_mode;
    *avail = (lc->na.cand_bottom_left && cur > ZS(x_tb - 1, (y_tb + size_in_tbs_v) & s->sps->tb_mask) ? OH_AV_BOTTOM_LEFT : 0)
           | (lc->na.cand_left ? OH_AV_LEFT : 0) | (lc->na.cand_up_left ? OH_AV_UP_LEFT : 0) | (lc->na.cand_up ? OH_AV_UP : 0)
           | (lc->na.cand_up_right && cur > ZS((x_tb + size_in_tbs_h) & s->sps->tb_mask, y_tb - 1) ? OH_AV_UP_RIGHT : 0);
#undef ZS
    /* picture borders truncate the candidates (hevcpred_template.c:111-114 works with sizes; a candidate wholly outside is none) */
    {
        const int n = 1 << log2_size, x = x0 >> hshift, y = y0 >> vshift, pw = s->sps->width >> hshift, ph = s->sps->height >> vshift;
        if (x + n >= pw) *avail &= ~OH_AV_UP_RIGHT;
        if (y + n >= ph) *avail &= ~OH_AV_BOTTOM_LEFT;
    }
}

#ifdef OH_WITH_ENGINE
double hook_now(void);
static __thread double tls_t_entry;
static struct { double to_rps, bind, to_newref, newref; } HT2;
static int timed_set_new_ref(HEVCContext *s, AVFrame **frame, int poc)
{
    const double t0 = hook_now();
    HT2.to_newref += t0 - tls_t_entry;
    const int r = ff_hevc_set_new_ref(s, frame, poc);
    HT2.newref += hook_now() - t0;
    return r;
}
#endif
static int frame_rps_and_bind(HEVCContext *s)
{
#ifdef OH_WITH_ENGINE
    const double t_b0 = hook_now();
    HT2.to_rps += t_b0 - tls_t_entry;
#endif
    int ret = ff_hevc_frame_rps(s);
    if (ret < 0)
        return ret;
    if (s->decoder_id > 1) {
        av_log(s->avctx, AV_LOG_ERROR, "recording hooks: one enhancement layer (the wrapper allocates two decoders, openHevcWrapper.c:28)\n");
        return AVERROR_PATCHWELCOME;
    }
    h_idx = s->decoder_id;
    const HEVCSPS *sps = s->sps;
    OhPicParams p;
    memset(&p, 0, sizeof(p));
    p.width = sps->width; p.height = sps->height; p.bit_depth = sps->bit_depth; p.chroma_format_idc = sps->chroma_format_idc;
    p.log2_ctb_size = sps->log2_ctb_size; p.log2_min_cb_size = sps->log2_min_cb_size; p.log2_min_tb_size = sps->log2_min_tb_size;
    p.log2_min_pu_size = sps->log2_min_pu_size;
    p.pcm_loop_filter_disable = sps->pcm_enabled_flag && sps->pcm.loop_filter_disable_flag;
    p.transquant_bypass_enable = s->pps->transquant_bypass_enable_flag;
    p.strong_intra_smoothing = sps->sps_strong_intra_smoothing_enable_flag;
    p.intra_smoothing_disabled = sps->spsRext.intra_smoothing_disabled_flag;
    p.cb_qp_offset = s->pps->cb_qp_offset; p.cr_qp_offset = s->pps->cr_qp_offset;
    p.sao_enabled = sps->sao_enabled; p.deblock_enabled = 1;
    p.constrained_intra_pred = s->pps->constrained_intra_pred_flag;
    if (!H.rec || memcmp(&p, &H.p, sizeof(p))) {
        if (H.rec) oh_rec_destroy(H.rec);
        H.rec = oh_rec_create(&p);
        H.p = p;
    }
    H.s = s;
    H.cur_id = (int)(s->ref - s->DPB);
    /* the pictures this one may reference: its reference picture set (hevc_refs.c:391-470), slot = position in this list */
    H.n_refs = 0;
    /* ... plus, in an enhancement layer, the inter-layer reference pictures (hevc_refs.c:738-756): the up-sampled base-layer picture */
    static const int lists[5] = { ST_CURR_BEF, ST_CURR_AFT, LT_CURR, IL_REF0, IL_REF1 };
    H.ilr_slot = -1;
    for (int l = 0; l < 5; l++)
        for (int i = 0; i < s->rps[lists[l]].nb_refs && H.n_refs < OH_MAX_REFS; i++) {
            if (l >= 3 && s->rps[lists[l]].ref[i] == s->inter_layer_ref) H.ilr_slot = H.n_refs;
            H.ref_ids[H.n_refs++] = (int)(s->rps[lists[l]].ref[i] - s->DPB);
        }
    int32_t ids[OH_MAX_REFS];
    for (int i = 0; i < H.n_refs; i++) ids[i] = H.ref_ids[i];
    oh_rec_begin(H.rec, H.cur_id, ids, H.n_refs);
    H.bl_base = NULL;
    if (H.ilr_slot >= 0 && s->BL_frame && s->BL_frame->frame) {
        /* what hevc.c:3241 hands to the up-sampling slot: the filter set-up of the SPS (hevc.c:446-501) and the scaled reference layer window */
        const HEVCWindow *win = &s->sps->scaled_ref_layer_window[s->vps->m_refLayerId[s->nuh_layer_id][0]];
        H.bl_base = s->BL_frame->frame->data[0];
        H.bl_id = (int)(s->BL_frame - ((HEVCContext *)((AVCodecContext *)s->avctx->BL_avcontext)->priv_data)->DPB);
        H.up.add_x_lum = s->up_filter_inf.addXLum; H.up.add_y_lum = s->up_filter_inf.addYLum;
        H.up.scale_x_lum = s->up_filter_inf.scaleXLum; H.up.scale_y_lum = s->up_filter_inf.scaleYLum;
        H.up.add_x_cr = s->up_filter_inf.addXCr; H.up.add_y_cr = s->up_filter_inf.addYCr;
        H.up.scale_x_cr = s->up_filter_inf.scaleXCr; H.up.scale_y_cr = s->up_filter_inf.scaleYCr;
        H.up.idx = s->up_filter_inf.idx;
        H.up.win_left = win->left_offset; H.up.win_right = win->right_offset; H.up.win_top = win->top_offset; H.up.win_bottom = win->bottom_offset;
    }
    if (g_bs_from_motion) {
        const size_t n_tb = (size_t)sps->min_tb_width * sps->min_tb_height;
        if (g_bs_call_n != n_tb) { free(g_bs_call); g_bs_call = (uint8_t *)malloc(n_tb); g_bs_call_n = g_bs_call ? n_tb : 0; }
        if (g_bs_call) memset(g_bs_call, 0, n_tb);
    }
#ifdef OH_WITH_ENGINE
    H.seq = next_picture_seq();       /* frame starts are serialised by the frame-thread protocol (ff_thread_finish_setup): decode order */
    H.cur_base = s->frame->data[0]; H.cur_ls = s->frame->linesize[0];
    for (int i = 0; i < H.n_refs; i++) {
        H.ref_base[i] = s->DPB[H.ref_ids[i]].frame->data[0]; H.ref_ls[i] = s->DPB[H.ref_ids[i]].frame->linesize[0];
    }
#endif
    oh_tables_set_intra_accessor(intra_from_hevc);
    oh_tables_bind(H.rec, s->frame->data, s->frame->linesize);
    for (int i = 0; i < H.n_refs; i++)
        oh_tables_bind_ref(i, s->DPB[H.ref_ids[i]].frame->data, s->DPB[H.ref_ids[i]].frame->linesize);
    H.open = 1;
#ifdef OH_WITH_ENGINE
    HT2.bind += hook_now() - t_b0;
#endif
    return ret;
}

/* end of the access unit: what passes 4-5 read goes from the context into the recorder; returns the work
 * list of the picture, or NULL when the access unit held no picture.  *untranslated: slot calls the hooks could not translate. */
__attribute__((visibility("default"))) const OhFrame *ref_hooked_finish(int *cur_id, int *poc, int *untranslated)
{
    if (!H.open)
        return NULL;
    HEVCContext *s = H.s;
    const HEVCSPS *sps = s->sps;
    H.open = 0;
    *untranslated = oh_tables_finish();
    if (*untranslated && getenv("OHEVC_HOOK_DEBUG")) {
        int why[8];
        oh_tables_untranslated_by_family(why);
        fprintf(stderr, "untranslated: transform_add %d, put_pcm %d, intra(no accessor) %d, intra(recorder) %d, emu %d, luma MC %d, l0 half %d, PU %d\n",
                why[0], why[1], why[2], why[3], why[4], why[5], why[6], why[7]);
    }
    *cur_id = H.cur_id; *poc = s->poc;
    const int n_ctb = sps->ctb_width * sps->ctb_height, n_pu = sps->min_pu_width * sps->min_pu_height;
    if (g_bs_from_motion && g_bs_call) {
        /* the grids stay zero: nothing can fall back on them */
        _Static_assert(sizeof(MvField) == sizeof(OhMvField), "OhMvField mirrors MvField as compiled");
        OhBsInputs *bi = oh_rec_bs_maps(H.rec);
        memcpy((void *)bi->mvf, s->ref->tab_mvf, (size_t)sps->min_pu_width * sps->min_pu_height * sizeof(MvField));
        memcpy((void *)bi->cbf_luma, s->cbf_luma, (size_t)sps->min_tb_width * sps->min_tb_height);
        memcpy((void *)bi->call_log2, g_bs_call, g_bs_call_n);
    } else {
        memcpy(oh_rec_vertical_bs(H.rec), s->vertical_bs, (size_t)s->bs_width * s->bs_height);
        memcpy(oh_rec_horizontal_bs(H.rec), s->horizontal_bs, (size_t)s->bs_width * s->bs_height);
    }
    memcpy(oh_rec_qp_y_tab(H.rec), s->qp_y_tab, (size_t)(sps->min_cb_width * sps->min_cb_height));
    memcpy(oh_rec_is_pcm(H.rec), s->is_pcm, (size_t)n_pu);
    if (s->pps->constrained_intra_pred_flag)
        for (int i = 0; i < n_pu; i++) oh_rec_is_intra(H.rec)[i] = s->ref->tab_mvf[i].pred_flag == PF_INTRA;
    OhCtbMaps *m = oh_rec_ctb_maps(H.rec);
    m->tiles_enabled = s->pps->tiles_enabled_flag;
    m->loop_filter_across_tiles = s->pps->loop_filter_across_tiles_enabled_flag;
    for (int ctb = 0; ctb < n_ctb; ctb++) {
        OhDeblockCtb *d = &oh_rec_deblock(H.rec)[ctb];
        d->beta_offset = (int8_t)s->deblock[ctb].beta_offset; d->tc_offset = (int8_t)s->deblock[ctb].tc_offset;
        OhSaoCtb *o = &oh_rec_sao(H.rec)[ctb];
        const SAOParams *a = &s->sao[ctb];
        memset(o, 0, sizeof(*o));
        for (int c = 0; c < 3; c++) {
            for (int k = 0; k < 5; k++) o->offset_val[c][k] = a->offset_val[c][k];
            o->band_position[c] = a->band_position[c]; o->eo_class[c] = (uint8_t)a->eo_class[c]; o->type_idx[c] = a->type_idx[c];
        }
        m->slice_addr[ctb] = s->tab_slice_address[ctb];
        m->filter_slice_edges[ctb] = s->filter_slice_edges[ctb];
        m->tile_id[ctb] = s->pps->tile_id[s->pps->ctb_addr_rs_to_ts[ctb]];
        m->deblock_disabled[ctb] = 0;                      /* such slices simply derive no boundary strengths (hevc.c:1577) */
    }
    /* the scaling lists in force (hevc_cabac.c:1480-1483 picks the PPS's when it carries its own): kept for ref_hooked_scaling_list */
    H.scaling_on = sps->scaling_list_enable_flag;
    if (H.scaling_on) {
        const ScalingList *sl = s->pps->scaling_list_data_present_flag ? &s->pps->scaling_list : &sps->scaling_list;
        memcpy(H.scaling.sl, sl->sl, sizeof(H.scaling.sl));
        memcpy(H.scaling.sl_dc, sl->sl_dc, sizeof(H.scaling.sl_dc));
    }
    return oh_rec_finish(H.rec);
}

/* two-layer streams: which layer's picture the next ref_hooked_finish / ref_hooked_inter_layer speak about (0 base, 1 enhancement) */
__attribute__((visibility("default"))) void ref_hooked_select_layer(int layer) { h_idx = layer == 1; }
/* the inter-layer reference of the enhancement-layer picture that was just recorded: its position in the work list's reference list
 * (-1: the picture has none), the DPB slot of the base layer's picture it is resampled from, and the resampling set-up */
__attribute__((visibility("default"))) int ref_hooked_inter_layer(int *bl_id, OhUpsample *up)
{
    if (H.ilr_slot >= 0 && H.bl_base) { *bl_id = H.bl_id; *up = H.up; }
    return H.bl_base ? H.ilr_slot : -1;
}

/* which picture an OUTPUT frame is: `luma` = plane 0 of the frame libOpenHevcGetOutput hands out, i.e. a DPB frame's data[0] moved
 * to the conformance window's origin (hevc_refs.c:248-254).  Returns the DPB slot (the picture id of the work lists) and the origin
 * in luma samples, -1 if the pointer lies in no picture of the DPB. */
__attribute__((visibility("default"))) int ref_hooked_locate(const void *luma, int *x, int *y)
{
    HEVCContext *s = H.s;
    if (!s || !s->sps || !luma)
        return -1;
    const uint8_t *p = (const uint8_t *)luma;
    for (int i = 0; i < (int)FF_ARRAY_ELEMS(s->DPB); i++) {
        const AVFrame *fr = s->DPB[i].frame;
        if (!fr || !fr->data[0] || fr->linesize[0] <= 0)
            continue;
        const ptrdiff_t off = p - fr->data[0];
        if (off < 0 || off >= (ptrdiff_t)fr->linesize[0] * s->sps->height)
            continue;
        *y = (int)(off / fr->linesize[0]);
        *x = (int)(off % fr->linesize[0]) >> s->sps->pixel_shift;
        return i;
    }
    return -1;
}

/* the scaling lists of the picture ref_hooked_finish just returned (as the decoder holds them after hevc_ps.c parsed or defaulted
 * them); returns 0 when the SPS has them off (flat 16).  tests/test_sparse_pin.py hands them over with the sparse levels. */
__attribute__((visibility("default"))) int ref_hooked_scaling_list(OhScalingList *out)
{
    if (H.scaling_on)
        *out = H.scaling;
    return H.scaling_on;
}

/* cross-component prediction.  hls_cross_component_pred (hevc.c:1186-1200) gets res_scale_val from these two
 * CABAC elements; here the elements are parsed as always, the value goes to the recording tables (oh_tables_cross) and the
 * decoder itself is told "0", so its own additions (hevc.c:1319-1331, 1352-1364, hevc_cabac.c:1942-1947) add nothing: the chroma
 * block reaches transform_add with its own residual only and the engine adds the scaled luma residual. */
void oh_tables_cross(int res_scale_val);
static int hooked_log2_res_scale_abs(HEVCContext *s, int idx)
{
    const int a = ff_hevc_log2_res_scale_abs(s, idx);
    const int neg = a ? ff_hevc_res_scale_sign_flag(s, idx) : 0;
    oh_tables_cross(a ? (1 << (a - 1)) * (1 - 2 * neg) : 0);
    return 0;
}
#define ff_hevc_log2_res_scale_abs(s, idx) hooked_log2_res_scale_abs(s, idx)

/* SURVEY §8 f2 on real streams: with ref_hooked_bs_from_motion(1) the work lists carry what ff_hevc_deblocking_boundary_strengths
 * READS (motion field, cbf_luma, where it was called and for which block size) instead of the grids it writes, and the engine /
 * the checker derive the boundary strengths.  The call sites (hevc.c:1578, 1607, 2400, 2484) go through this wrapper. */
static void hooked_boundary_strengths(HEVCContext *s, int x0, int y0, int log2_size)
{
    if (g_bs_from_motion && g_bs_call) {                   /* every min-TB cell of the block carries its size (OhBsInputs.call_log2) */
        const int l = s->sps->log2_min_tb_size, n = 1 << (log2_size - l), w = s->sps->min_tb_width, h = s->sps->min_tb_height;
        for (int j = y0 >> l; j < (y0 >> l) + n && j < h; j++)
            for (int i = x0 >> l; i < (x0 >> l) + n && i < w; i++)
                g_bs_call[(size_t)j * w + i] = (uint8_t)log2_size;
    }
    ff_hevc_deblocking_boundary_strengths(s, x0, y0, log2_size);
}
#define ff_hevc_deblocking_boundary_strengths(s, x, y, l) hooked_boundary_strengths(s, x, y, l)
__attribute__((visibility("default"))) void ref_hooked_bs_from_motion(int on) { g_bs_from_motion = on; }

#ifdef OH_WITH_ENGINE
/* ============================ the engine behind the decoder (drop-in library) ============================ */
#include <pthread.h>
#include "../include/ohevc_hip.h"

/* Engine pictures are keyed by the HOST buffer of the decoder's frame (plane 0 origin): libavcodec's frame pool hands the same
 * buffers out again and again, a buffer holds one picture at a time, and the frame libOpenHevcGetOutput exposes keeps its buffer
 * referenced until the next libOpenHevcDecode — so "the engine picture of this host buffer" is the picture the caller is asking
 * for even after the DPB slot has moved on (hevc_refs.c:45-65 ff_hevc_unref_frame). */
static struct {
    OhEngine *e;
    int failed;
    struct { const uint8_t *base; size_t span; int id; OhPicParams p; } pic[256];
    int n;
    pthread_mutex_t lock;
} E = { .lock = PTHREAD_MUTEX_INITIALIZER };
static __thread HEVCContext *tls_s;           /* the context whose access unit this thread is decoding (hooked_decode_frame) */
static __thread int tls_md5_plane;            /* calc_md5 calls seen for the picture (0..2) */
static __thread uint8_t tls_md5[48];
/* OHEVC_HOOK_TIMING=1: where the wall time of the decode callback goes (printed when the decoder is closed) */
#include <time.h>
double hook_now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + ts.tv_nsec * 1e-9; }
static struct { double decode, finish, submit, fetch, setup, turn; int pictures; } HT;

/* hand-over in DECODE order: with frame threads a picture may finish parsing before the one it references; the engine runs work
 * lists in the order they are submitted (that order IS the dependency order on its stream), so a worker waits for its turn */
static struct { pthread_mutex_t mu; pthread_cond_t cv; unsigned long long next, turn; } SEQ = { PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, 0 };
static unsigned long long next_picture_seq(void)
{
    pthread_mutex_lock(&SEQ.mu);
    const unsigned long long v = SEQ.next++;
    pthread_mutex_unlock(&SEQ.mu);
    return v;
}
static void seq_wait_turn(unsigned long long seq)
{
    pthread_mutex_lock(&SEQ.mu);
    while (SEQ.turn != seq)
        pthread_cond_wait(&SEQ.cv, &SEQ.mu);
    pthread_mutex_unlock(&SEQ.mu);
}
static void seq_done(void)
{
    pthread_mutex_lock(&SEQ.mu);
    SEQ.turn++;
    pthread_cond_broadcast(&SEQ.cv);
    pthread_mutex_unlock(&SEQ.mu);
}

static OhEngine *the_engine(void)
{
    if (!E.e && !E.failed) {
        const char *dev = getenv("OHEVC_DEVICE");
        if (oh_engine_create(&E.e, dev ? atoi(dev) : 0) != OH_OK) {
            fprintf(stderr, "libopenhevc_hip: no MI355X engine (there is no CPU fallback)\n");
            E.e = NULL; E.failed = 1;
        }
    }
    return E.e;
}
static int same_geometry(const OhPicParams *a, const OhPicParams *b)
{
    return a->width == b->width && a->height == b->height && a->bit_depth == b->bit_depth && a->chroma_format_idc == b->chroma_format_idc;
}
/* the engine picture of a decoder frame (created on first sight, re-created when the buffer comes back with another geometry) */
static int engine_pic_of(const uint8_t *base, int linesize, const OhPicParams *p)
{
    OhEngine *e = the_engine();
    if (!e || !base) return -1;
    int slot = -1;
    for (int i = 0; i < E.n; i++)
        if (E.pic[i].base == base) { slot = i; break; }
    if (slot >= 0 && same_geometry(&E.pic[slot].p, p))
        return E.pic[slot].id;
    if (slot < 0) {
        /* a NEW buffer: whatever the map still holds inside its range belongs to buffers the frame pool has freed since */
        const uint8_t *lo = base, *hi = lo + (size_t)linesize * (size_t)p->height;
        for (int i = 0; i < E.n; ) {
            if (E.pic[i].base < hi && E.pic[i].base + E.pic[i].span > lo) {
                oh_pic_free(e, E.pic[i].id);
                E.pic[i] = E.pic[--E.n];
            } else
                i++;
        }
        if (E.n == (int)(sizeof(E.pic) / sizeof(E.pic[0]))) { fprintf(stderr, "libopenhevc_hip: more than %d host frame buffers in use\n", E.n); return -1; }
        slot = E.n++;
    } else {
        oh_pic_free(e, E.pic[slot].id);
    }
    int id = -1;
    if (oh_pic_alloc(e, p, &id) != OH_OK) { fprintf(stderr, "libopenhevc_hip: %s\n", oh_engine_last_error(e)); E.n -= (slot == E.n - 1); return -1; }
    E.pic[slot].base = base; E.pic[slot].span = (size_t)linesize * (size_t)p->height; E.pic[slot].id = id; E.pic[slot].p = *p;
    return id;
}

/* end of an access unit on the thread that decoded it: the picture's work list -> engine (asynchronous).  Idempotent. */
static int finish_and_submit(HEVCContext *s)
{
    int cur = -1, poc = 0, bad = 0;
    if (!H.open || H.s != s)
        return 0;
    const unsigned long long seq = H.seq;
    const double t_f0 = hook_now();
    const OhFrame *f = ref_hooked_finish(&cur, &poc, &bad);
    HT.finish += hook_now() - t_f0;
    { const double t_w0 = hook_now();
    seq_wait_turn(seq);                                       /* every picture that was started takes its turn, submitted or not */
    HT.turn += hook_now() - t_w0; }
    if (!f) {
        seq_done();
        return 0;
    }
    const double t_s0 = hook_now();
    HT.pictures++;
    if (bad) { seq_done(); fprintf(stderr, "libopenhevc_hip: %d table-slot calls of the picture (poc %d) could not be turned into work-list items\n", bad, poc); return -1; }
    pthread_mutex_lock(&E.lock);
    OhEngine *e = the_engine();
    int rc = e ? 0 : -1;
    OhFrame g = *f;
    if (!rc) {
        g.cur_pic = engine_pic_of(H.cur_base, H.cur_ls, &f->p);
        if (g.cur_pic < 0) rc = -1;
        for (int r = 0; r < OH_MAX_REFS && !rc; r++) {
            g.ref_pics[r] = -1;
            if (f->ref_pics[r] >= 0 && r < H.n_refs) {
                g.ref_pics[r] = engine_pic_of(H.ref_base[r], H.ref_ls[r], &f->p);
                if (g.ref_pics[r] < 0) rc = -1;
            }
        }
    }
    if (!rc && H.ilr_slot >= 0) {
        /* SHVC: the inter-layer reference picture is the base layer's picture of this access unit (submitted a moment ago, decode order
         * is execution order) resampled to this layer's size — the reference's up-sampling slots (hevcdsp_template.c:1834-2438), which
         * it drives per CTB from ff_upsample_block (hevc_filter.c:1370-1426); here ONE launch pair over the whole picture, before the
         * work list whose motion compensation reads it */
        int bl = -1;
        for (int i = 0; i < E.n; i++)
            if (E.pic[i].base == H.bl_base) bl = E.pic[i].id;
        if (bl < 0 || g.ref_pics[H.ilr_slot] < 0) { fprintf(stderr, "libopenhevc_hip: picture poc %d: the base layer's picture is not in the engine\n", poc); rc = -1; }
        else if (oh_pic_upsample(e, g.ref_pics[H.ilr_slot], bl, &H.up) != OH_OK) { fprintf(stderr, "libopenhevc_hip: picture poc %d: %s\n", poc, oh_engine_last_error(e)); rc = -1; }
    }
    if (!rc && H.scaling_on && g.sparse) g.scaling = &H.scaling;
    if (!rc && oh_frame_submit(e, &g) != OH_OK) { fprintf(stderr, "libopenhevc_hip: picture poc %d: %s\n", poc, oh_engine_last_error(e)); rc = -1; }
    H.last_engine_pic = rc ? -1 : g.cur_pic;
    pthread_mutex_unlock(&E.lock);
    seq_done();
    HT.submit += hook_now() - t_s0;
    oh_rec_recycle(H.rec);                                    /* the lists are in the engine's staging buffer: clear the maps on this worker's own time */
    return rc;
}

/* decode_checksum_sei (hevc.c:4146-4169): the decoder hashes s->ref->frame's planes — host memory nothing was decoded into.  Its
 * three calc_md5() calls (hevc.c:4623-4638) keep their place; the av_md5_sum inside them is replaced: it hands out the digest the
 * GPU computed over the engine's picture (md5.hip: oh_pics_md5, 48 bytes over PCIe instead of the picture), so the decoder's own
 * "Correct MD5 (poc, plane)" / "Incorrect MD5" lines judge what the engine decoded. */
static void hooked_md5_final(uint8_t *dst)
{
    if (tls_md5_plane == 0) {
        memset(tls_md5, 0, sizeof(tls_md5));
        if (tls_s && finish_and_submit(tls_s) == 0 && H.last_engine_pic >= 0) {
            pthread_mutex_lock(&E.lock);
            if (oh_pics_md5(E.e, &H.last_engine_pic, 1, tls_md5) != OH_OK)
                fprintf(stderr, "libopenhevc_hip: %s\n", oh_engine_last_error(E.e));
            pthread_mutex_unlock(&E.lock);
        }
    }
    memcpy(dst, tls_md5 + 16 * tls_md5_plane, 16);
    tls_md5_plane = (tls_md5_plane + 1) % 3;
}

/* the samples of the frame libOpenHevcDecode released for output, into the host planes the wrapper exposes: `out` is the wrapper's
 * AVFrame (data[] = the DPB frame's planes moved to the conformance window's origin, hevc_refs.c:248-254; width / height = the
 * cropped size) */
__attribute__((visibility("default"))) int oh_hooked_fetch_output(AVFrame *out)
{
    int rc = -1;
    const double t_g0 = hook_now();
    pthread_mutex_lock(&E.lock);
    int hit = -1;
    for (int i = 0; i < E.n && hit < 0; i++)                  /* the buffer itself (no window) ... */
        if (E.pic[i].base == out->data[0]) hit = i;
    for (int i = 0; i < E.n && hit < 0; i++) {                /* ... or a position inside one (the conformance window's origin) */
        const ptrdiff_t off = out->data[0] - E.pic[i].base;
        if (off >= 0 && (size_t)off < E.pic[i].span) hit = i;
    }
    OhDownload *dl = NULL;
    ptrdiff_t strides[3] = { out->linesize[0], out->linesize[1], out->linesize[2] };
    uint8_t *planes[3] = { out->data[0], out->data[1], out->data[2] };
    if (hit >= 0 && E.e && out->linesize[0] > 0) {
        const ptrdiff_t off = out->data[0] - E.pic[hit].base;
        const OhPicParams *p = &E.pic[hit].p;
        const int ps = p->bit_depth > 8;
        const int y = (int)(off / out->linesize[0]), x = (int)(off % out->linesize[0]) >> ps;
        const OhWindow win = { x, p->width - x - out->width, y, p->height - y - out->height };
        if (win.right < 0 || win.bottom < 0)
            fprintf(stderr, "libopenhevc_hip: output window %dx%d+%d+%d leaves the %dx%d picture\n", out->width, out->height, x, y, p->width, p->height);
        else if (oh_pic_download_start(E.e, E.pic[hit].id, &win, &dl) != OH_OK)     /* the copies are enqueued behind the picture's batch ... */
            fprintf(stderr, "libopenhevc_hip: %s\n", oh_engine_last_error(E.e));
    }
    OhEngine *eng = E.e;
    pthread_mutex_unlock(&E.lock);
    /* ... and waited for WITHOUT the engine lock: the frame-thread workers keep handing their pictures over meanwhile */
    if (dl) {
        const int drc = oh_download_finish(eng, dl, planes, strides);
        if (drc != OH_OK) fprintf(stderr, "libopenhevc_hip: fetching the output picture failed (%d)\n", drc);
        else rc = 0;
    }
    HT.fetch += hook_now() - t_g0;
    return rc;
}
/* libOpenHevcStartDecoder: the engine (HIP start-up, streams, kernels' attributes) comes up with the decoder, not with its first picture */
__attribute__((visibility("default"))) int oh_hooked_engine_open(void)
{
    pthread_mutex_lock(&E.lock);
    const int ok = the_engine() != NULL;
    pthread_mutex_unlock(&E.lock);
    return ok ? 0 : -1;
}
__attribute__((visibility("default"))) void oh_hooked_engine_close(void)
{
    if (getenv("OHEVC_HOOK_TIMING") && HT.pictures)
        fprintf(stderr, "libopenhevc_hip timing, ms per picture over %d pictures: decode callback (host decoder incl. recording slots) %.2f, "
                        "finish (side arrays + intra schedule) %.2f, hand-over to the engine %.2f, output fetch %.2f; frame threads: callback entry -> "
                        "ff_thread_finish_setup (serial across workers) %.2f, waiting for the hand-over turn %.2f\n", HT.pictures,
                HT.decode * 1e3 / HT.pictures, HT.finish * 1e3 / HT.pictures, HT.submit * 1e3 / HT.pictures, HT.fetch * 1e3 / HT.pictures,
                HT.setup * 1e3 / HT.pictures, HT.turn * 1e3 / HT.pictures);
    if (getenv("OHEVC_HOOK_TIMING") && HT.pictures)
        fprintf(stderr, "   of the serial stretch: callback entry -> ff_hevc_set_new_ref %.2f (NAL split, slice header, frame-start memsets), ff_hevc_set_new_ref %.2f (frame buffers), "
                        "callback entry -> ff_hevc_frame_rps %.2f, frame_rps + recorder begin + binding %.2f\n",
                HT2.to_newref * 1e3 / HT.pictures, HT2.newref * 1e3 / HT.pictures, HT2.to_rps * 1e3 / HT.pictures, HT2.bind * 1e3 / HT.pictures);
    memset(&HT2, 0, sizeof(HT2));
    if (getenv("OHEVC_HOOK_TIMING") && HT.pictures && E.e) {
        double ms[OH_N_HOST_TIMES]; uint64_t calls[OH_N_HOST_TIMES];
        static const char *nm[OH_N_HOST_TIMES] = { "upload", "upload: count loops", "upload: arena", "upload: wait for a staging buffer", "upload: memcpy to pinned",
                                                   "upload: enqueue", "execute", "execute: wait for the preparation", "release" };
        if (oh_engine_host_times(E.e, ms, calls, OH_N_HOST_TIMES, 0) == OH_OK) {
            fprintf(stderr, "   engine host time, ms per picture:");
            for (int i = 0; i < OH_N_HOST_TIMES; i++) fprintf(stderr, " %s %.2f;", nm[i], ms[i] / HT.pictures);
            fprintf(stderr, " bytes over PCIe per picture %.2f MB\n", (double)oh_engine_upload_bytes(E.e, 0) / HT.pictures / 1e6);
        }
    }
    if (getenv("OHEVC_HOOK_TIMING") && HT.pictures && E.e) {
        uint64_t m[6];
        if (oh_engine_memory(E.e, m) == OH_OK)
            fprintf(stderr, "   engine memory after %d pictures: %llu work-list arenas alive (%.1f MB, %llu of them free in the pool), %llu pinned staging buffers (%.1f MB), "
                            "%llu lists awaiting a deferred free, %d engine pictures\n", HT.pictures, (unsigned long long)m[0], m[1] / 1e6, (unsigned long long)m[2],
                    (unsigned long long)m[3], m[4] / 1e6, (unsigned long long)m[5], E.n);
    }
    memset(&HT, 0, sizeof(HT));
    pthread_mutex_lock(&SEQ.mu); SEQ.next = SEQ.turn = 0; pthread_mutex_unlock(&SEQ.mu);
    pthread_mutex_lock(&E.lock);
    if (E.e) { oh_engine_sync(E.e); oh_engine_destroy(E.e); }
    E.e = NULL; E.n = 0; E.failed = 0;
    pthread_mutex_unlock(&E.lock);
}
__attribute__((visibility("default"))) int oh_hooked_engine_sync(void)
{
    pthread_mutex_lock(&E.lock);
    const int rc = E.e ? oh_engine_sync(E.e) : 0;
    if (rc) fprintf(stderr, "libopenhevc_hip: %s\n", oh_engine_last_error(E.e));
    pthread_mutex_unlock(&E.lock);
    return rc;
}
#define av_md5_sum(dst, src, len)     hooked_md5_final(dst)
#endif /* OH_WITH_ENGINE */

#define ff_hevc_dsp_init(c, bd)   do { ff_hevc_dsp_init(c, bd);  ff_hevcdsp_init_hip((void *)(c), bd); } while (0)
#define ff_hevc_pred_init(c, bd)  do { ff_hevc_pred_init(c, bd); ff_hevcpred_init_hip((void *)(c), bd); } while (0)
#define ff_videodsp_init(c, bd)   do { ff_videodsp_init(c, bd);  ff_videodsp_init_hip((void *)(c), bd); } while (0)
#define ff_hevc_frame_rps(s)      frame_rps_and_bind(s)
/* The in-loop filter drivers are dropped (passes 4-5 run on the GPU) — but with FRAME threads they also carry the row progress the
 * next picture's thread waits for (hevc_filter.c:1040-1050 ff_thread_report_progress).  With recording slots nobody on the host
 * reads reference SAMPLES any more, so the motion-compensation waits (hevc.c:1951-1958 hevc_await_progress) fall away; what a later
 * picture still reads on the host is the collocated MOTION FIELD (hevc_mvs.c:262 temporal_luma_motion_vector waits for row y), and a
 * CTB row's motion field is complete when its last CTB has been parsed: that is reported here. */
static void hooked_row_progress(HEVCContext *s, int x_ctb, int y_ctb, int ctb_size)
{
    if ((s->threads_type & FF_THREAD_FRAME) && s->ref && x_ctb >= s->sps->width - ctb_size)
        ff_thread_report_progress(&s->ref->tf, y_ctb + ctb_size, 0);
}
#define ff_hevc_hls_filters(s, x, y, c) hooked_row_progress(s, x, y, c)
#define ff_hevc_hls_filter(s, x, y, c)  hooked_row_progress(s, x, y, c)
#define ff_thread_await_progress(tf, y, field) ((void)0)
#ifdef OH_WITH_ENGINE
#define ff_hevc_set_new_ref(s, f, poc) timed_set_new_ref(s, f, poc)
#define ff_thread_finish_setup(avctx) do { HT.setup += hook_now() - tls_t_entry; ff_thread_finish_setup(avctx); } while (0)
#endif

#include "libavcodec/hevc.c"


#ifdef OH_WITH_ENGINE
/* The AVCodec's decode callback (hevc.c:4555-4571 `.decode = hevc_decode_frame`), followed by the hand-over of the picture the
 * access unit held: on whichever thread libavcodec runs the callback.  Installed when the library is loaded. */
static int hooked_decode_frame(AVCodecContext *avctx, void *data, int *got_output, AVPacket *avpkt)
{
    tls_s = avctx->priv_data;
    h_idx = tls_s->decoder_id == 1;
    tls_md5_plane = 0;
    const double t_d0 = hook_now();
    tls_t_entry = t_d0;
    const int ret = hevc_decode_frame(avctx, data, got_output, avpkt);
    HT.decode += hook_now() - t_d0;
    if (finish_and_submit(avctx->priv_data) < 0)
        return AVERROR_EXTERNAL;
    return ret;
}
__attribute__((constructor)) static void install_engine_hooks(void)
{
    ff_hevc_decoder.decode = hooked_decode_frame;
    const char *b = getenv("OHEVC_BS_FROM_MOTION");          /* boundary strengths derived on the GPU from the decoder's motion field (bs_kernel) */
    if (b && atoi(b)) g_bs_from_motion = 1;
}
#endif
