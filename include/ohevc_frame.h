/*
 * ohevc_frame.h — the picture work list ("frame command buffer").
 *
 * This is the hand-off format between the host CTU loop (openHEVC's hevc.c, which keeps
 * doing entropy decode, MV derivation, dequant and boundary-strength derivation) and the
 * MI355X block-reconstruction passes.  Every table call the reference makes per block
 *  is RECORDED as one fixed-size item; the GPU replays a whole picture
 * (or a batch of pictures) per pass:
 *
 *     pass 1  inter prediction   OhPu[]      <- put_hevc_{qpel,epel}*        hevc.c:1641-1949, 2103-2153
 *     pass 2  residual           OhTu[]      <- idct[] / transform_add[]     hevc_cabac.c:1868-1949
 *     pass 3  intra (wavefront)  OhIntra[]   <- hpc.intra_pred[]             hevc.c:1215..1417, hevcpred_template.c:30-344
 *     pass 4  deblock V, H       BS/QP grids <- deblocking_filter_CTB        hevc_filter.c:345-581
 *     pass 5  SAO                OhSaoCtb[]  <- sao_filter_CTB               hevc_filter.c:197-322
 *
 * Plain C, fixed-width integers, no pointers inside items: the same bytes are consumed by the
 * CPU oracle (oracle/), by the HIP engine (openhevc_amd/csrc/) and produced by the synthetic
 * stream generator (openhevc_amd/synth/).  Names follow the HEVC domain (CTB, TU, PU, BS).
 */
#ifndef OHEVC_FRAME_H
#define OHEVC_FRAME_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OH_MAX_REFS   16      /* DPB slots a picture may reference (hevc.h: MAX_REFS 16) */
#define OH_NO_REF     0xFF
#define OH_NO_WP      0xFFFFu
#define OH_NO_COEFF   0xFFFFFFFFu

/* ---- sequence / picture parameters the hot path needs (subset of SPS/PPS, hevc.h:729-920) ---- */
typedef struct OhPicParams {
    int32_t width, height;            /* luma samples (sps->width/height)                      */
    int32_t bit_depth;                /* 8, 9, 10 or 12 (luma == chroma, README.md:16)         */
    int32_t chroma_format_idc;        /* 0 mono, 1 4:2:0, 2 4:2:2, 3 4:4:4                     */
    int32_t log2_ctb_size;            /* 4..6                                                  */
    int32_t log2_min_cb_size;         /* >= 3                                                  */
    int32_t log2_min_tb_size;         /* >= 2                                                  */
    int32_t log2_min_pu_size;         /* log2_min_cb_size - 1 (hevc_ps.c)                      */
    int32_t pcm_loop_filter_disable;  /* sps->pcm_enabled_flag && pcm.loop_filter_disable_flag */
    int32_t transquant_bypass_enable; /* pps->transquant_bypass_enable_flag                    */
    int32_t strong_intra_smoothing;   /* sps->sps_strong_intra_smoothing_enable_flag           */
    int32_t intra_smoothing_disabled; /* sps->spsRext.intra_smoothing_disabled_flag            */
    int32_t cb_qp_offset, cr_qp_offset; /* pps offsets used by chroma_tc (hevc_filter.c:62-89) */
    int32_t sao_enabled;              /* sps->sao_enabled                                      */
    int32_t deblock_enabled;          /* 0: skip pass 4 entirely (all BS would be 0)           */
    int32_t constrained_intra_pred;   /* pps->constrained_intra_pred_flag (hevcpred_template.c:116-249)  */
    int32_t reserved[3];
} OhPicParams;

/* ---- pass 1: inter prediction.  One item = one PU rectangle, already cut to <= 64x64 luma. ---- */
typedef struct OhPu {
    uint16_t x, y;                /* luma position of the rectangle (hevc.c:2103: x0,y0)            */
    uint8_t  w, h;                /* luma size, 4..64, multiples of 4 (8x4 / 4x8 are the smallest)  */
    uint8_t  ref[2];              /* slot in OhFrame.ref_pics per list, OH_NO_REF = list unused     */
    int16_t  mv[2][2];            /* [list][x,y] quarter-luma-sample units (MvField.mv)             */
    uint16_t wp;                  /* index into OhFrame.wp[], OH_NO_WP = default weighting          */
    uint16_t reserved;
} OhPu;                           /* 20 bytes */

/* explicit weighted prediction parameters for one (ref_idx_l0, ref_idx_l1) pair (hevc.c:1767-1773) */
typedef struct OhWeights {
    int16_t w[2][3];              /* [list][plane]  luma_weight_l0/1, chroma_weight_l0/1[..][0..1]  */
    int16_t o[2][3];              /* [list][plane]  offsets, in 8-bit units (scaled by BD-8 inside) */
    uint8_t log2_denom[2];        /* [0] luma_log2_weight_denom, [1] chroma_log2_weight_denom       */
    uint8_t reserved[2];
} OhWeights;                      /* 28 bytes */

/* ---- pass 2: residual.  One item = one coded transform block of one plane. ---- */
enum OhTuKind {
    OH_TU_IDCT   = 0,   /* hevcdsp.idct[log2-2] / idct_dc (DC-only is the same arithmetic)         */
    OH_TU_DST4   = 1,   /* hevcdsp.idct_4x4_luma (intra 4x4 luma)                                  */
    OH_TU_SKIP   = 2,   /* hevcdsp.transform_skip                                                  */
    OH_TU_BYPASS = 3,   /* cu_transquant_bypass: coefficients are the residual                     */
    OH_TU_PCM    = 4    /* put_pcm: `coeffs` hold final samples, no prediction is added            */
};
enum OhTuFlags {
    OH_TUF_ADD_NOW   = 1,  /* prediction already complete after pass 1 (inter CU): add in pass 2  */
    OH_TUF_RDPCM     = 2,  /* hevcdsp.transform_rdpcm after skip / on bypass                      */
    OH_TUF_RDPCM_VER = 4,  /* rdpcm mode 1 (vertical)                                             */
    OH_TUF_ROTATE    = 8,  /* transform_skip_rotation (4x4): coefficients reversed before skip    */
    OH_TUF_SPARSE    = 16, /* coefficients come as quantised (position, level) pairs, see OhFrame.sparse */
    OH_TUF_CROSS     = 32, /* chroma block with cross-component prediction (4:4:4 range extension, hevc.c:1319-1365,
                              hevc_cabac.c:1942-1947): residual += (res_scale_val * luma residual) >> 3, see OhFrame.tu_cross */
    OH_TUF_KEEP_RES  = 64  /* luma block whose residual a OH_TUF_CROSS block reads: kept in the residual pool even when added at once */
};

/* Sparse residual hand-off (the step upstream of the inverse transform): instead of a dense
 * de-quantised N x N block the host passes what residual_coding parsed — the non-zero LEVELS — and de-quantisation
 * (hevc_cabac.c:1478-1494, 1818-1841) runs on the GPU.  Record of one block in OhFrame.sparse (uint32 words):
 *   word 0      n | qp << 16 | matrix_id << 24      n = number of pairs (<= N*N); qp = the block's final QP incl.
 *                                                   qp_bd_offset (0..75); matrix_id 0..5 = 3*(!intra) + c_idx into
 *                                                   OhFrame.scaling, OH_FLAT_MATRIX = flat 16 (also for
 *                                                   transform-skipped blocks > 4x4, hevc_cabac.c:1485)
 *   word 1..n   pos | (uint16_t)level << 16         pos = y*N + x
 * Kinds: IDCT, DST4, SKIP (bypass and PCM blocks carry no quantised levels and stay dense). */
#define OH_FLAT_MATRIX 0xffu
typedef struct OhScalingList {    /* hevc.h:722-727 */
    uint8_t sl[4][6][64];
    uint8_t sl_dc[2][6];
} OhScalingList;
typedef struct OhTu {
    uint16_t x, y;                /* position in samples of plane c_idx                             */
    uint8_t  c_idx;               /* 0 Y, 1 Cb, 2 Cr                                                */
    uint8_t  log2_size;           /* 2..5                                                           */
    uint8_t  kind;                /* enum OhTuKind                                                  */
    uint8_t  flags;               /* enum OhTuFlags                                                 */
    uint32_t coeff_off;           /* first int16 of the dense N*N block in OhFrame.coeffs           */
} OhTu;                           /* 12 bytes */

/* ---- pass 3: intra prediction, executed in dependency levels. ---- */
enum OhIntraAvail {               /* resolved candidate flags (hevcpred_template.c:100-109)         */
    OH_AV_BOTTOM_LEFT = 1, OH_AV_LEFT = 2, OH_AV_UP_LEFT = 4, OH_AV_UP = 8, OH_AV_UP_RIGHT = 16
};
typedef struct OhIntra {
    uint16_t x, y;                /* position in samples of plane c_idx                             */
    uint8_t  c_idx;
    uint8_t  log2_size;           /* 2..5                                                           */
    uint8_t  mode;                /* 0 planar, 1 DC, 2..34 angular                                  */
    uint8_t  avail;               /* enum OhIntraAvail bits                                         */
    uint32_t tu;                  /* index of this block's OhTu (residual added right after the
                                     prediction), OH_NO_COEFF when cbf == 0                        */
} OhIntra;                        /* 12 bytes */

/* Pass 3 is scheduled as a CTU WAVEFRONT (the GPU form of the reference's WPP rows,
 * hevc.c:2751-2832): a CTU that contains intra blocks may start once the CTUs whose samples its
 * blocks read (left, up-left, up, up-right) are done — that is its LEVEL; inside the CTU the
 * blocks are ordered in SUB-LEVELS (block b's sub-level = 1 + max sub-level of the same-CTU
 * blocks it reads).  One workgroup reconstructs one CTU, sub-level after sub-level. */
typedef struct OhIntraCtu {
    uint32_t sub_first;           /* first entry of this CTU in OhFrame.sub_start                   */
    uint16_t n_sub;               /* number of sub-levels                                           */
    uint16_t ctu;                 /* raster index of the CTU                                        */
} OhIntraCtu;                     /* 8 bytes */

/* ---- pass 4: deblocking side arrays ---- */
typedef struct OhDeblockCtb { int8_t beta_offset, tc_offset; } OhDeblockCtb;   /* hevc.h:1083 */

/* ---- pass 5: SAO parameters per CTB (hevc.h:514-523, only what the filter reads) ---- */
typedef struct OhSaoCtb {
    int16_t offset_val[3][5];     /* SaoOffsetVal, [plane][0..4], [0] is always 0                   */
    uint8_t band_position[3];
    uint8_t eo_class[3];          /* 0 horiz, 1 vert, 2 135deg, 3 45deg                             */
    uint8_t type_idx[3];          /* 0 off, 1 band, 2 edge                                          */
    uint8_t edge_flags;           /* non-filterable CTB edges (slice/tile), hevc_filter.c:206-252:
                                     bit0 left, bit1 right (vert_edge[0..1]); bit2 up, bit3 bottom
                                     (horiz_edge[0..1]); bits 4..7 diag_edge[0..3]                  */
} OhSaoCtb;                       /* 40 bytes */

/* ---- one picture's work list (host pointers; the engine copies them at submit) ---- */
typedef struct OhFrame {
    OhPicParams p;
    int32_t  cur_pic;                 /* picture id reconstructed by this work list                */
    int32_t  ref_pics[OH_MAX_REFS];   /* picture ids OhPu.ref[] indexes                             */

    uint32_t n_pu;      const OhPu      *pu;
    uint32_t n_wp;      const OhWeights *wp;

    uint32_t n_tu;      const OhTu      *tu;
    uint64_t n_coeff;   const int16_t   *coeffs;

    uint32_t n_intra;   const OhIntra    *intra;      /* sorted by (CTU level, CTU, sub-level): any
                                                         prefix is closed under dependencies      */
    uint32_t n_ictu;    const OhIntraCtu *ictu;       /* CTUs holding intra blocks, by CTU level  */
    uint32_t n_sub;     const uint32_t   *sub_start;  /* n_sub+1 offsets into intra[]: the blocks of
                                                         sub-level s of ictu[k] are
                                                         [sub_start[k.sub_first+s], sub_start[k.sub_first+s+1]) */
    uint32_t n_levels;  const uint32_t   *level_start;/* n_levels+1 offsets into ictu[]           */

    /* (width>>2) x (height>>2) grids, index (x + y*bs_width)>>2 exactly as hevc_filter.c:388,487;
       bs_size bytes each, bs_size >= the reference's padded allocation (hevc.c:170-171)          */
    uint32_t bs_size;   const uint8_t *vertical_bs, *horizontal_bs;
    const int8_t       *qp_y_tab;     /* min_cb_width x min_cb_height (hevc.c:158)                */
    const uint8_t      *is_pcm;       /* min_pu_width x min_pu_height, may be NULL (hevc.c:147)   */
    const OhDeblockCtb *deblock;      /* ctb_width x ctb_height                                   */
    const OhSaoCtb     *sao;          /* ctb_width x ctb_height, may be NULL when !sao_enabled    */
    const uint8_t      *is_intra;     /* min_pu_width x min_pu_height: 1 where the covering CU is intra (tab_mvf[].pred_flag ==
                                         PF_INTRA, incl. PCM CUs); required when constrained_intra_pred, else may be NULL */
    /* sparse residual hand-off (OH_TUF_SPARSE blocks): all NULL / 0 when every block is dense */
    uint32_t n_sparse;  const uint32_t *sparse;      /* records, uint32 words                                     */
    const uint32_t     *tu_sparse;    /* per OhTu: word offset of its record in sparse[], OH_NO_COEFF for dense blocks */
    const OhScalingList *scaling;     /* scaling lists in use (sps/pps), NULL when every block uses the flat matrix */
    /* cross-component prediction: per OhTu, for OH_TUF_CROSS blocks `luma TU index | (res_scale_val & 0xff) << 24`
     * (res_scale_val = +-1, 2, 4, 8 as a signed byte), OH_NO_COEFF otherwise; NULL when the picture has none */
    const uint32_t     *tu_cross;
    /* boundary strengths derived on the GPU: when set, vertical_bs / horizontal_bs may be NULL and the
     * engine computes both grids from these maps, bit-exact with ff_hevc_deblocking_boundary_strengths (hevc_filter.c:584-941) */
    const struct OhBsInputs *bs_in;
    /* 16x16 CTBs with horizontally subsampled chroma only, else ignored.  In that configuration the reference's CTB driver runs a
     * CTB's SAO before the horizontal chroma edges reached the first chroma column of its right neighbour: per CTB,
     * bit 0 = the edges of the CTB's own row were still pending there, bit 1 = those of the CTB row below (oh_sao_pending_driver
     * simulates ff_hevc_hls_filters / ff_hevc_hls_filter, hevc_filter.c:1027-1064, over the picture's decoding order).
     * NULL: CTBs decoded in raster order, one thread — the engine derives the same bits in closed form. */
    const uint8_t      *sao_pending;  /* ctb_width x ctb_height, may be NULL */
    uint32_t flags;                   /* OH_FRAME_*: how the arrays are held (0: ordinary host memory, byte grids) */
} OhFrame;
/* OhFrame.flags.
 * OH_FRAME_PINNED     every array the list points at lies in memory from oh_host_alloc() (ohevc_hip.h): the GPU pulls them over PCIe
 *                     straight from where they lie (one kernel, prep.hip prep_pull) — no staging copy on the host, no DMA request
 *                     per array.  They must stay untouched until the list's copy has completed (oh_frames_execute of the list, or
 *                     any wait on the engine, is behind it).  This is the hand-over of a recorder that writes its lists into blocks
 *                     the engine lent it.
 * OH_FRAME_BS_PACKED  vertical_bs / horizontal_bs hold the strengths FOUR TO THE BYTE (entry i in bits 2 (i & 3) of byte i >> 2,
 *                     (bs_size + 3) / 4 bytes each): the form they travel in and live in on the GPU (a strength is 0..2).  A
 *                     recorder packs them while it copies the decoder's byte grids (oh_pack_bs). */
enum { OH_FRAME_PINNED = 1, OH_FRAME_BS_PACKED = 2 };
static inline void oh_pack_bs(uint8_t *dst, const uint8_t *src, size_t n)
{
    size_t i = 0;
    for (; i + 4 <= n; i += 4)
        dst[i >> 2] = (uint8_t)((src[i] & 3) | (src[i + 1] & 3) << 2 | (src[i + 2] & 3) << 4 | (src[i + 3] & 3) << 6);
    if (i < n) {
        unsigned v = 0;
        for (size_t k = i; k < n; k++) v |= (unsigned)(src[k] & 3) << (2 * (k & 3));
        dst[i >> 2] = (uint8_t)v;
    }
}

/* one entry of the reference's motion field = MvField as compiled (TEST_MV_POC defined, hevc.h:73, 1032-1041): 24 bytes, compared
 * as a whole by boundary_strength()'s memcmp (hevc_filter.c:600), padding included */
typedef struct OhMvField {
    int16_t  mv[2][2];            /* Mv mv[2]: x, y in quarter samples                        */
    int32_t  poc[2];              /* POC of the reference picture of each list                */
    uint32_t pred_flag;           /* PF_INTRA 0, PF_L0 1, PF_L1 2, PF_BI 3                    */
    uint8_t  ref_idx[2];
    uint8_t  pad[2];
} OhMvField;

/* what ff_hevc_deblocking_boundary_strengths() reads, as whole-picture maps (hevc_filter.c:805-941) */
typedef struct OhBsInputs {
    const OhMvField *mvf;         /* min_pu_width x min_pu_height: s->ref->tab_mvf                                                  */
    const uint8_t   *cbf_luma;    /* min_tb_width x min_tb_height: s->cbf_luma (hevc.c:1566-1575)                                   */
    const uint8_t   *call_log2;   /* min_tb_width x min_tb_height, EVERY cell a call's block covers: log2 size of the block the function was called for
                                     (the transform unit, hevc.c:1578, or the whole coding block, :1607 :2400 :2484); 0 = never
                                     called there (slice_deblocking_filter_disabled_flag): both grids stay 0                        */
    const uint8_t   *ctb_flags;   /* ctb_width x ctb_height: OH_BSF_* of the CTB's slice / position (hevc.c:2636-2637)              */
    int32_t loop_filter_across_tiles;          /* pps->loop_filter_across_tiles_enabled_flag                                        */
} OhBsInputs;
enum { OH_BSF_UP_SLICE = 1, OH_BSF_UP_TILE = 2,       /* lc->slice_or_tiles_up_boundary   */
       OH_BSF_LEFT_SLICE = 4, OH_BSF_LEFT_TILE = 8,   /* lc->slice_or_tiles_left_boundary << 2 */
       OH_BSF_ACROSS_SLICES = 16 };                   /* s->sh.slice_loop_filter_across_slices_enabled_flag */

/* ---- derived geometry helpers (all integer, shared by every consumer) ---- */
static inline int oh_hshift(const OhPicParams *p, int c) { return c && (p->chroma_format_idc == 1 || p->chroma_format_idc == 2); }
static inline int oh_vshift(const OhPicParams *p, int c) { return c && p->chroma_format_idc == 1; }
static inline int oh_ctb_width(const OhPicParams *p)  { return (p->width  + (1 << p->log2_ctb_size) - 1) >> p->log2_ctb_size; }
static inline int oh_ctb_height(const OhPicParams *p) { return (p->height + (1 << p->log2_ctb_size) - 1) >> p->log2_ctb_size; }
static inline int oh_min_cb_width(const OhPicParams *p)  { return p->width  >> p->log2_min_cb_size; }
static inline int oh_min_cb_height(const OhPicParams *p) { return p->height >> p->log2_min_cb_size; }
static inline int oh_min_pu_width(const OhPicParams *p)  { return p->width  >> p->log2_min_pu_size; }
static inline int oh_min_pu_height(const OhPicParams *p) { return p->height >> p->log2_min_pu_size; }
/* the reference's padded BS allocation: max of hevc.c:170 and :171 so one size serves both grids */
static inline uint32_t oh_bs_size(const OhPicParams *p)
{
    uint32_t bw = (uint32_t)p->width >> 2, bh = (uint32_t)p->height >> 2;
    uint32_t a = (bw + 4u * (1u << oh_hshift(p, 1))) * bh;
    uint32_t b = bw * (bh + 4u * (1u << oh_vshift(p, 1)));
    return (a > b ? a : b) + 64u;
}
/* qp_y_tab is allocated one row/column larger than the picture (hevc.c:118-119: +1) */
static inline uint32_t oh_qp_tab_size(const OhPicParams *p)
{
    return (uint32_t)(oh_min_cb_width(p) + 1) * (uint32_t)(oh_min_cb_height(p) + 1);
}

/* Layout of one picture buffer ("half", see ohevc_hip.h): planes back to back, every row and
 * every plane padded to 256 bytes.  Returns the half's size in bytes; stride[] in SAMPLES,
 * offset[] in bytes.  Shared by the engine, the tests and the multi-GPU exchange code. */
static inline uint64_t oh_pic_half_layout(const OhPicParams *p, int32_t stride[3], uint64_t offset[3])
{
    uint64_t bpp = p->bit_depth > 8 ? 2 : 1, total = 0;
    for (int c = 0; c < (p->chroma_format_idc ? 3 : 1); c++) {
        uint64_t w = (uint64_t)(p->width >> oh_hshift(p, c)), h = (uint64_t)(p->height >> oh_vshift(p, c));
        uint64_t row = (w * bpp + 255) / 256 * 256;
        stride[c] = (int32_t)(row / bpp);
        offset[c] = total;
        total += (row * h + 255) / 256 * 256;
    }
    return total;
}

/* ---- SHVC inter-layer up-sampling ------------------------------------------------
 * Parameters of the resampling of a base-layer picture into the enhancement layer's geometry: the
 * reference's UpsamplInf (hevc.h:347-357) plus the scaled reference layer window (HEVCWindow,
 * hevc.h:384-389) the slots receive as `Enhscal`. */
enum { OH_UP_DEFAULT = 0, OH_UP_X2 = 1, OH_UP_X1_5 = 2, OH_UP_SNR = 3 };      /* hevc.h:340-345 */
typedef struct OhUpsample {
    int32_t add_x_lum, add_y_lum, scale_x_lum, scale_y_lum;
    int32_t add_x_cr, add_y_cr, scale_x_cr, scale_y_cr;
    int32_t idx;                                   /* OH_UP_* : which slot variant the reference picks */
    int32_t win_left, win_right, win_top, win_bottom;
} OhUpsample;

/* hevc.c:446-501 (set_sps of an enhancement layer): sizes are the layers' luma sizes inside their windows */
static inline void oh_upsample_setup(OhUpsample *u, int width_bl, int height_bl, int width_el_pic, int height_el_pic,
                                     int win_left, int win_right, int win_top, int win_bottom, int phase_align_flag)
{
    const int phase_xc = 0, phase_yc = 1, phase_x = phase_align_flag << 1, phase_y = phase_align_flag << 1;
    const int height_el = height_el_pic - win_bottom - win_top, width_el = width_el_pic - win_left - win_right;
    u->win_left = win_left; u->win_right = win_right; u->win_top = win_top; u->win_bottom = win_bottom;
    u->scale_x_lum = ((width_bl << 16) + (width_el >> 1)) / width_el;
    u->scale_y_lum = ((height_bl << 16) + (height_el >> 1)) / height_el;
    u->add_x_lum = ((phase_x * u->scale_x_lum + 2) >> 2) + (1 << 11);
    u->add_y_lum = ((phase_y * u->scale_y_lum + 2) >> 2) + (1 << 11);
    u->add_x_cr = (((phase_xc + phase_align_flag) * u->scale_x_lum + 2) >> 2) + (1 << 11);
    u->add_y_cr = (((phase_yc + phase_align_flag) * u->scale_y_lum + 2) >> 2) + (1 << 11);
    u->scale_x_cr = u->scale_x_lum;
    u->scale_y_cr = u->scale_y_lum;
    if (u->scale_x_lum == 65536 && u->scale_y_lum == 65536) u->idx = OH_UP_SNR;
    else if (u->scale_x_lum == 32768 && u->scale_y_lum == 32768) u->idx = OH_UP_X2;
    else if (u->scale_x_lum == 43691 && u->scale_y_lum == 43691) u->idx = OH_UP_X1_5;
    else u->idx = OH_UP_DEFAULT;
}

#ifdef __cplusplus
}
#endif
#endif /* OHEVC_FRAME_H */
