/*
 * ohevc_synth.h — synthetic "decoded syntax" generator (C ABI, host only).
 *
 * There is no HEVC bitstream, encoder or conformance stream in the build environment
 * (SURVEY.md §8c), and entropy decoding stays on the host in the reference anyway.  A synthetic
 * stream is therefore generated one level below the bitstream: for each picture a random but
 * LEGAL coding quadtree (CU / PU / TU partitioning, prediction modes, motion vectors, dense
 * dequantised coefficient blocks, QP map, boundary strengths, SAO parameters) is produced and
 * pushed through the recorder API exactly as the reference's CTU loop would
 * (include/ohevc_recorder.h).  Fixed seed => identical work lists everywhere.
 */
#ifndef OHEVC_SYNTH_H
#define OHEVC_SYNTH_H

#include "ohevc_recorder.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OhSynthParams {
    uint64_t seed;
    int32_t  slice_type;        /* 0 = I, 1 = P (list 0 only), 2 = B                           */
    int32_t  n_refs;            /* usable entries of ref_pics (1..16) for inter pictures       */
    int32_t  intra_pct;         /* % of CUs coded intra in P/B pictures                        */
    int32_t  skip_pct;          /* % of inter CUs without residual (skip / rqt_root_cbf = 0)   */
    int32_t  bi_pct;            /* % of eligible PUs bi-predicted (B pictures)                 */
    int32_t  frac_mv_pct;       /* % of MVs with a fractional part                             */
    int32_t  mv_range;          /* |mv| bound in quarter samples (may point outside the picture) */
    int32_t  cbf_pct;           /* % of transform blocks with coded coefficients               */
    int32_t  weighted_pct;      /* % of PUs using explicit weighted prediction                 */
    int32_t  split_pct;         /* % chance to split a CU / TU one level further               */
    int32_t  qp_base, qp_var;   /* CU QP = qp_base +- qp_var                                   */
    int32_t  sao_pct;           /* % of CTBs with SAO (band or edge) per component             */
    int32_t  tskip_pct;         /* % of 4x4 blocks using transform_skip                        */
    int32_t  pcm_pct;           /* % of CUs coded PCM (needs pcm_loop_filter_disable to matter)*/
    int32_t  bypass_pct;        /* % of CUs with cu_transquant_bypass (if enabled in params)   */
    int32_t  vary_deblock_offsets; /* 1: per-CTB beta/tc offsets differ (multi-slice quirks)   */
    int32_t  sparse_pct;        /* % of transform blocks handed over as quantised levels (OH_TUF_SPARSE) */
    int32_t  scaling_list;      /* 1: random scaling lists, blocks name their matrix            */
    int32_t  ccp_pct;           /* 4:4:4 only: % of transform units with cross-component prediction */
    int32_t  bs_from_motion;    /* 1: no finished BS grids; the motion field, cbf map, call sizes and CTB flags go to the engine (OhBsInputs) */
    /* slices and tiles (0 / 1 everywhere = one slice, one tile: the streams of the older fixtures are unchanged) */
    int32_t  n_slices;          /* > 1: that many slices at random CTB addresses (raster scan, no tiles)                      */
    int32_t  tile_cols, tile_rows; /* > 1: uniform tile grid; see OH_SYNTH_SLICE_PER_TILE                                       */
    int32_t  slice_knobs;       /* OH_SYNTH_* bits                                                                          */
} OhSynthParams;
enum { OH_SYNTH_NO_LF_ACROSS_SLICES = 1,   /* about half of the slices get slice_loop_filter_across_slices_enabled_flag = 0   */
       OH_SYNTH_NO_LF_ACROSS_TILES = 2,    /* pps->loop_filter_across_tiles_enabled_flag = 0                                  */
       OH_SYNTH_DEBLOCK_OFF_SLICES = 4,    /* about a third of the slices get slice_deblocking_filter_disabled_flag = 1       */
       OH_SYNTH_SLICE_PER_TILE = 8,        /* every tile is its own slice (else one slice covers all tiles)                   */
       OH_SYNTH_SLICE_OFFSETS = 16 };      /* every slice draws its own beta / tc offsets                                     */

/* sensible defaults for a mid-QP picture of the given slice type */
void oh_synth_defaults(OhSynthParams *sp, int slice_type, uint64_t seed);

/* Generates one picture into the recorder (calls oh_rec_begin ... oh_rec_finish) and returns
 * the finished work list (owned by the recorder). */
const OhFrame *oh_synth_picture(OhRecorder *rec, const OhSynthParams *sp, int cur_pic,
                                const int32_t *ref_pics, int n_ref_pics);

#ifdef __cplusplus
}
#endif
#endif
