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
