/*
 * ohevc_recorder.h — host-side work-item recorder (C ABI, no GPU dependency).
 *
 * This is what the reference's CTU loop talks to instead of executing pixels: every call site
 * that today invokes a HEVCDSPContext / HEVCPredContext slot appends
 * one fixed-size item here.  oh_rec_finish() returns the OhFrame the engine (ohevc_hip.h) or the
 * oracle consumes.  The recorder owns all arrays; they stay valid until the next oh_rec_begin().
 *
 *   reference call site                                   recorder call
 *   hls_prediction_unit -> luma_mc_uni/bi, chroma_mc_uni/bi oh_rec_pu()          hevc.c:2103-2153
 *   ff_hevc_hls_residual_coding -> idct[], transform_add[] oh_rec_tu()          hevc_cabac.c:1868-1949
 *   hls_transform_unit -> hpc.intra_pred[]                 oh_rec_intra()       hevc.c:1215-1417
 *   hls_pcm_sample -> put_pcm                              oh_rec_tu(OH_TU_PCM) hevc.c:1587-1640
 *   ff_hevc_deblocking_boundary_strengths (stays host)     oh_rec_vertical_bs()/oh_rec_horizontal_bs()
 *   hls_coding_unit qp_y_tab / is_pcm writes               oh_rec_qp_y_tab()/oh_rec_is_pcm()
 *   hls_sao_param                                          oh_rec_sao()         hevc.c:1112-1181
 */
#ifndef OHEVC_RECORDER_H
#define OHEVC_RECORDER_H

#include "ohevc_frame.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OhRecorder OhRecorder;

OhRecorder *oh_rec_create(const OhPicParams *p);
const OhPicParams *oh_rec_params(const OhRecorder *r);
void        oh_rec_destroy(OhRecorder *r);

/* start a new picture: clears item lists and zeroes the BS grids / is_pcm like hevc_frame_start
 * does (hevc.c:3207-3210; is_pcm is cleared too — the reference's leak of stale flags across
 * pictures, SURVEY.md §7, is a host-side matter of what the caller writes into it). */
void oh_rec_begin(OhRecorder *r, int cur_pic, const int32_t *ref_pics, int n_ref_pics);
/* optional, once the work list oh_rec_finish() returned has been consumed (uploaded / copied): clears the per-picture maps NOW so that
 * the next oh_rec_begin() does not have to (frame threads: oh_rec_begin runs inside the stretch that is serial across the workers).
 * The lists the OhFrame points at stay valid until that oh_rec_begin(); the maps it points at (BS grids, QP, PCM, SAO, deblock) do not. */
void oh_rec_recycle(OhRecorder *r);

/* inter PU; returns 0 or -1 on invalid geometry.  wp may be NULL (default weighting). */
int oh_rec_pu(OhRecorder *r, int x, int y, int w, int h, int ref0, int mv0x, int mv0y,
              int ref1, int mv1x, int mv1y, const OhWeights *wp);

/* residual block: copies the N*N coefficients; returns the TU index (for oh_rec_intra) */
uint32_t oh_rec_tu(OhRecorder *r, int c_idx, int x, int y, int log2_size, int kind, int flags,
                   const int16_t *coeffs);

/* same block handed over as quantised levels (OH_TUF_SPARSE, ohevc_frame.h): pairs[i] = pos | (uint16_t)level << 16.
 * The dense pool keeps a slot for the block (the residual lands there), its content is not read. */
uint32_t oh_rec_tu_sparse(OhRecorder *r, int c_idx, int x, int y, int log2_size, int kind, int flags,
                          int qp, int matrix_id, int n, const uint32_t *pairs);
/* cross-component prediction (hevc.c:1319-1365): chroma block tu_c (index from oh_rec_tu*, recorded with its own coefficients
 * or, when cbf is 0, as an OH_TU_BYPASS block of zeros) takes (res_scale_val * residual of luma block tu_y) >> 3 on top */
int oh_rec_tu_cross(OhRecorder *r, uint32_t tu_c, uint32_t tu_y, int res_scale_val);
/* boundary strengths on the GPU: instead of filling the BS grids through oh_rec_bs*(), hand over the maps
 * ff_hevc_deblocking_boundary_strengths() reads (caller-owned until the frame is submitted; ohevc_frame.h: OhBsInputs) */
void oh_rec_bs_inputs(OhRecorder *r, const OhBsInputs *in);
/* same, with maps owned by the recorder: allocated on first use, zeroed at the first call after oh_rec_begin(); the caller fills them
 * through the returned struct's pointers (cast away the const) and sets loop_filter_across_tiles (default 1) */
OhBsInputs *oh_rec_bs_maps(OhRecorder *r);
/* scaling lists of the picture (zeroed by oh_rec_create; only read when a block names a matrix) */
OhScalingList *oh_rec_scaling_list(OhRecorder *r);

/* intra block; tu = index returned by oh_rec_tu for the same block or OH_NO_COEFF.
 * Computes the block's dependency level from the levels of the neighbours it reads. */
int oh_rec_intra(OhRecorder *r, int c_idx, int x, int y, int log2_size, int mode, int avail, uint32_t tu);
/* the same, returning the block's index for oh_rec_intra_attach_tu (OH_NO_COEFF on failure) — what a caller needs when several
 * threads append to one recorder (the reference's slice / wavefront threads): every appending entry point takes the recorder's lock */
uint32_t oh_rec_intra_idx(OhRecorder *r, int c_idx, int x, int y, int log2_size, int mode, int avail, uint32_t tu);

/* two-step form for callers that learn the residual after the prediction (the table slots):
 * record with tu = OH_NO_COEFF, then attach.  index = value of oh_rec_n_intra() before the record. */
uint32_t oh_rec_n_intra(const OhRecorder *r);
int oh_rec_intra_attach_tu(OhRecorder *r, uint32_t intra_index, uint32_t tu);

/* side arrays the caller fills in place (sizes: oh_bs_size / oh_qp_tab_size / min_pu grid / CTBs) */
uint8_t      *oh_rec_vertical_bs(OhRecorder *r);
uint8_t      *oh_rec_horizontal_bs(OhRecorder *r);
int8_t       *oh_rec_qp_y_tab(OhRecorder *r);
uint8_t      *oh_rec_is_pcm(OhRecorder *r);
uint8_t      *oh_rec_is_intra(OhRecorder *r);   /* min-PU map of intra CUs (tab_mvf pred_flag), zeroed by oh_rec_begin; only
                                                 consumed when OhPicParams.constrained_intra_pred */
OhDeblockCtb *oh_rec_deblock(OhRecorder *r);
OhSaoCtb     *oh_rec_sao(OhRecorder *r);

/* ---- slices and tiles ------------------------------------------------------------------------------------------------
 * The passes never see slice headers; what they need arrives derived: OhSaoCtb.edge_flags (the unfilterable CTB edges of
 * sao_filter_CTB, hevc_filter.c:206-252), the boundary-strength grids already gated at slice / tile boundaries
 * (hevc_filter.c:819-824, 857-862) or OhBsInputs.ctb_flags, and the intra candidate flags.  A caller with several slices or
 * tiles fills these per-CTB maps (raster order; the reference's s->tab_slice_address, s->filter_slice_edges, pps->tile_id
 * looked up through ctb_addr_rs_to_ts, and the slice's slice_deblocking_filter_disabled_flag) and oh_rec_finish() derives the
 * SAO edge flags and, with oh_rec_bs_maps(), the BS flags from them; oh_rec_avail() then also requires the neighbour to lie
 * in the same slice and tile (6.4.1; hevc.c:2636-2641).  Without a call to oh_rec_ctb_maps() the picture is one slice, one tile. */
typedef struct OhCtbMaps {
    int32_t *slice_addr;            /* s->tab_slice_address[ctb]: raster address of the first CTB of the slice (hevc.c:2600)   */
    uint8_t *filter_slice_edges;    /* slice_loop_filter_across_slices_enabled_flag of the CTB's slice (hevc.c:2679)           */
    uint8_t *deblock_disabled;      /* slice_deblocking_filter_disabled_flag: no boundary strengths are derived (hevc.c:1577)  */
    int32_t *tile_id;               /* pps->tile_id[ctb_addr_rs_to_ts[ctb]]                                                    */
    int32_t  tiles_enabled;         /* pps->tiles_enabled_flag                                                                 */
    int32_t  loop_filter_across_tiles;   /* pps->loop_filter_across_tiles_enabled_flag                                         */
} OhCtbMaps;
/* OhFrame.sao_pending for a picture decoded in tile scan by ONE thread (hls_decode_entry, hevc.c:2643-2697: ff_hevc_hls_filters after
 * every CTB in decoding order, the bottom-right CTB's ff_hevc_hls_filter at the end): replays the calls over the decoding order the
 * tile ids imply and notes, for each CTB's sao_filter_CTB, whether deblocking_filter_CTB(cx + 2, cy) and (cx + 2, cy + 1) — the
 * calls that filter the horizontal chroma edges of the right neighbour's columns (hevc_filter.c:526-530) — had run.  out: ctbs bytes */
void oh_sao_pending_driver(const int32_t *tile_id, int ctb_width, int ctb_height, uint8_t *out);
/* recorder-owned maps, reset at the first call after oh_rec_begin() (slice 0, tile 0, filtering across slices on) */
OhCtbMaps *oh_rec_ctb_maps(OhRecorder *r);
/* the maps if the current / last finished picture used them, else NULL (does not switch them on) */
const OhCtbMaps *oh_rec_ctb_maps_in_use(const OhRecorder *r);

/* lc->slice_or_tiles_{left,up}_boundary and the slice's across-slices flag of one CTB as OH_BSF_* bits (hevc.c:2619-2637) */
static inline int oh_ctb_bs_flags(const OhCtbMaps *m, int ctb_width, int rs)
{
    const int x = rs % ctb_width, y = rs / ctb_width, in_slice = rs - m->slice_addr[rs];
    int tl = 0, tu = 0, sl, su;
    if (m->tiles_enabled) {
        tl = x > 0 && m->tile_id[rs] != m->tile_id[rs - 1];
        sl = x > 0 && m->slice_addr[rs] != m->slice_addr[rs - 1];
        tu = y > 0 && m->tile_id[rs] != m->tile_id[rs - ctb_width];
        su = y > 0 && m->slice_addr[rs] != m->slice_addr[rs - ctb_width];
    } else {
        sl = in_slice <= 0;
        su = in_slice < ctb_width;
    }
    return (su ? OH_BSF_UP_SLICE : 0) | (tu ? OH_BSF_UP_TILE : 0) | (sl ? OH_BSF_LEFT_SLICE : 0) | (tl ? OH_BSF_LEFT_TILE : 0) |
           (m->filter_slice_edges[rs] ? OH_BSF_ACROSS_SLICES : 0);
}

/* the unfilterable edges of one CTB for the SAO pass, OhSaoCtb.edge_flags layout (hevc_filter.c:206-252) */
static inline int oh_ctb_sao_edge_flags(const OhCtbMaps *m, int ctb_width, int ctb_height, int rs)
{
    const int x = rs % ctb_width, y = rs / ctb_width;
    const int lfase = m->filter_slice_edges[rs], no_tile = m->tiles_enabled && !m->loop_filter_across_tiles;
    if (!no_tile && lfase)
        return 0;
    const int e0 = x == 0, e1 = y == 0, e2 = x == ctb_width - 1, e3 = y == ctb_height - 1;
#define OH_SL_(d) (!lfase && m->slice_addr[rs] != m->slice_addr[rs + (d)])
#define OH_TL_(d) (no_tile && m->tile_id[rs] != m->tile_id[rs + (d)])
    const int lt = !e0 && OH_TL_(-1), rt = !e2 && OH_TL_(1), ut = !e1 && OH_TL_(-ctb_width), bt = !e3 && OH_TL_(ctb_width);
    int f = 0;
    if (!e0 && (OH_SL_(-1) || lt)) f |= 1;
    if (!e2 && (OH_SL_(1) || rt)) f |= 2;
    if (!e1 && (OH_SL_(-ctb_width) || ut)) f |= 4;
    if (!e3 && (OH_SL_(ctb_width) || bt)) f |= 8;
    if (!e0 && !e1 && (OH_SL_(-ctb_width - 1) || lt || ut)) f |= 16;
    if (!e1 && !e2 && (OH_SL_(-ctb_width + 1) || rt || ut)) f |= 32;
    if (!e2 && !e3 && (OH_SL_(ctb_width + 1) || rt || bt)) f |= 64;
    if (!e0 && !e3 && (OH_SL_(ctb_width - 1) || lt || bt)) f |= 128;
#undef OH_SL_
#undef OH_TL_
    return f;
}

/* sort intra items into dependency levels and expose the picture's work list.  Returns NULL when an allocation failed while
 * the picture was recorded (the recording calls then returned -1 / OH_NO_COEFF and dropped their item; the table slots, which
 * cannot report anything, leave it to this call — the decoder treats the picture as lost, like a decode error). */
const OhFrame *oh_rec_finish(OhRecorder *r);

/* resolved intra candidate flags for a block, from the recorder's own "already reconstructed"
 * map (single slice / single tile: 6.4.1 z-scan availability == decoded earlier and inside the
 * picture; hevc_mvs.c:41-58 + hevcpred_template.c:100-109).  x,y,size in LUMA samples.
 * A caller that has the reference's lc->na and min_tb_addr_zs at hand passes those instead. */
int  oh_rec_avail(const OhRecorder *r, int x_luma, int y_luma, int w_luma, int h_luma);
void oh_rec_mark_decoded(OhRecorder *r, int x_luma, int y_luma, int w_luma, int h_luma);

#ifdef __cplusplus
}
#endif
#endif
