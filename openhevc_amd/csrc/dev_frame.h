/*
 * dev_frame.h — device-side view of one picture's work list (lives in HBM; kernels receive a
 * pointer and read the fields with scalar loads).
 */
#ifndef OHEVC_DEV_FRAME_H
#define OHEVC_DEV_FRAME_H

#include <stdint.h>
#include "../../include/ohevc_frame.h"

struct DevPlanes {
    void   *p[3];
    int32_t stride[3];            /* in samples */
    int32_t w[3], h[3];
};

/* one <=8x8 block of one plane of a PU: the unit a quarter wave (16 lanes) interpolates.  The lists
 * are compacted: ref[0] / mv[0] is the first list the PU uses, ref[1] the second or OH_NO_REF. */
struct DevMcJob {
    uint16_t x, y;                /* position in the plane, samples (even)                      */
    uint8_t  w, h;                /* size in the plane, 2..8 (even)                             */
    uint8_t  ref[2];              /* slots in DevFrame.refs                                     */
    int16_t  mv[2][2];            /* quarter-luma-sample units, same order as ref[]             */
    uint16_t wp;                  /* index into DevFrame.wp, OH_NO_WP = default weighting       */
    uint8_t  c_idx;
    uint8_t  flags;               /* OH_MCF_*                                                   */
};                                /* 20 bytes */
enum { OH_MCF_FROM_L1 = 1 };      /* uni-prediction whose only list is list 1 (selects the weights) */

/* LDS layout of the intra CTU kernel.  The sample area (uint16 units) holds per plane `main`: hc rows of
 * rs = wc + 4 entries, sample (x,y) at [y*rs + x + 4] (column -1 at +3), then per plane `top`: row -1,
 * sample x at [x + 4], x in [-1, 2*wc).  The host resolves block offsets into it (DevIntra.cm_off/top_off);
 * the per-launch arrays behind it (descriptors, sub-level table, residual, per-wave edges) are sized by
 * the host from the level's CTUs (OhIntraLaunch). */
#define OH_CTU_MAX 64
#define OH_MAX_CTU_BLOCKS 768                      /* 64x64 4:4:4 all 4x4 */
#define OH_INTRA_WAVE_LDS 576                      /* per-wave edge arrays: 132 ints for one block or 4 x 36 ints for four
                                                      <=8x8 blocks in 16-lane slots (intra.hip: IntraLds) */
struct OhCtuAreas { uint32_t main[3], top[3], total; };
static __host__ __device__ inline OhCtuAreas oh_ctu_areas(int log2_ctb, int chroma_format_idc)
{
    OhCtuAreas a;
    uint32_t off = 0;
    const int np = chroma_format_idc ? 3 : 1, ctb = 1 << log2_ctb;
    for (int c = 0; c < 3; c++) {
        const int hs = c && (chroma_format_idc == 1 || chroma_format_idc == 2), vs = c && chroma_format_idc == 1;
        a.main[c] = off;
        if (c < np) off += (uint32_t)((ctb >> vs) * ((ctb >> hs) + 4));
    }
    for (int c = 0; c < 3; c++) {
        const int hs = c && (chroma_format_idc == 1 || chroma_format_idc == 2);
        a.top[c] = off;
        if (c < np) off += (uint32_t)(2 * (ctb >> hs) + 8);
    }
    a.total = off;
    return a;
}
struct OhIntraLaunch {                             /* one wavefront level of a batch of pictures = one launch */
    uint32_t level;                                /* index into DevFrame.lvl_start                           */
    uint32_t off_items, off_sub, off_small, off_res, off_wave, lds_bytes;   /* byte offsets into the dynamic LDS block */
    uint32_t waves;                                /* waves per workgroup (CTU)                               */
    uint32_t staged;                               /* 1: the CTUs' residual spans are staged in LDS (all contiguous, and the launch is one the chip holds at once);
                                                      0: every block fetches its residual from the pool in HBM, a sub-level ahead (intra.hip: slots_prepare) */
    uint32_t phases;                               /* sub-level s is finished by the waves with wave % phases == s % phases; >= 2, divides waves */
};

enum { OH_IF_FILTER = 1, OH_IF_STRONG_CAND = 2, OH_IF_EDGE = 4,              /* DevIntra.flags bits 0..2 */
       OH_IF_CIP = 8, OH_IF_CIP_CORNER = 128 };                              /* constrained intra pred: slow path; corner sample intra */
enum { OH_IC_PLANAR = 0, OH_IC_DC, OH_IC_ANG_V, OH_IC_ANG_H, OH_IC_PURE_V, OH_IC_PURE_H };   /* flags >> 4 */

/* intra block as the kernel wants it: OhIntra plus everything that only depends on the block's
 * geometry and mode, resolved once on the host at upload (engine.hip) */
struct DevIntra {
    uint16_t x, y;                /* position in the plane (HBM store address)                  */
    uint8_t  c_idx, log2_size, mode, avail;
    uint32_t res_off;             /* int16 offset into DevFrame.res, OH_NO_COEFF when cbf == 0  */
    uint16_t cm_off;              /* LDS: index of the block's sample (0,0), in uint16 from CtuLds::main */
    uint16_t top_off;             /* LDS: index of top[0] (row above the block); top[-1] is at -1 */
    uint16_t rs;                  /* LDS row stride of the plane                                */
    uint8_t  tr_size, bl_size;    /* samples inside the picture above-right / below-left (:111-114) */
    int8_t   angle;               /* intraPredAngle, 0 for planar / DC                          */
    uint8_t  flags;               /* OH_IF_* | (OH_IC_* << 4)                                   */
    int16_t  inv_angle;           /* invAngle when the side projection is needed, else 0        */
    uint32_t res_lds;             /* offset of the residual inside the staged span (res_off - res_lo) */
    uint16_t cip_left, cip_top;   /* constrained intra pred: bit k = the 4-sample group k of the left column / top row
                                     belongs to an intra CU (IS_INTRA(-1, 4k) / IS_INTRA(4k, -1), hevcpred_template.c:37-41) */
};

/* OhIntraCtu plus the span of the residual pool its blocks use (staged in LDS when it fits) */
struct DevIntraCtu {
    uint32_t sub_first;
    uint16_t n_sub, ctu;
    uint32_t res_lo, res_cnt;     /* int16 elements; res_cnt == 0: blocks read their residual from HBM */
    uint32_t item0, n_items;      /* the CTU's blocks: intra[item0 .. item0 + (n_items & 0xffff)) (= sub_start range); n_items >> 16: the
                                     samples they cover / 64 (all planes): a CTU that is mostly inter writes back its blocks, not its rectangle */
    int16_t  bx0, bx1, by0, by1;  /* CTU-local luma rectangle [x0,x1) x [y0,y1) that covers every sample a block of the
                                     CTU reads (its row above and column to the left, up to 2n): what gets staged */
};

/* transform block as uploaded: OhTu plus the word offset of its sparse record (OH_TUF_SPARSE) */
struct DevTu { OhTu t; uint32_t sparse_off; };     /* 16 bytes */

/* cross-component prediction of one chroma block (OH_TUF_CROSS): residual_c += (scale * residual_y) >> 3 */
struct DevCross { uint16_t x, y; uint8_t c_idx, log2_size, flags; int8_t scale; uint32_t res_c, res_y; };    /* 16 bytes */

/* what the preparation kernels (prep.hip) report back to the host: the first violation found in the work list, and what
 * sizes the launches (the host reads it — a few hundred bytes, copied to pinned memory behind the kernels — before the first
 * execute of the list) */
enum { OH_PE_OK = 0, OH_PE_PU = 1, OH_PE_TU = 2, OH_PE_INTRA = 3, OH_PE_INTRA_TABLES = 4 };
struct DevLevelStat {                 /* one intra wavefront level: what sizes the launch that runs it */
    uint32_t n_ctu, max_items, max_sub, max_res, staged, pad;
    uint64_t sum_items, sum_sub;
};                                    /* 40 bytes */
struct DevSummary {
    uint32_t err, err_item;           /* OH_PE_*, index of the offending item */
    uint32_t tu_cnt[4], n_cross, n_levels;
    uint32_t intra_area64, max_passes;/* samples the intra blocks cover / 64 (all planes); wave passes of the heaviest schedule entry */
    /* DevLevelStat[n_levels] follows */
};
/* DevFrame.ctu_aux[k] */
enum { OH_AUX_PASSES = 0xffff,        /* wave passes of the entry: groups of four <= 8x8 blocks + the bigger blocks */
       OH_AUX_DEP_SHIFT = 16,         /* bits 16..19: a block of the entry gathers samples of the left / up-left / up / up-right CTU */
       OH_AUX_WAITS = 1u << 29,       /* ctu_wait[] of the entry holds at least one entry */
       OH_AUX_AWAITED = 1u << 30,     /* another entry waits for this one: it must publish its samples (release) and set ctu_done */
       OH_AUX_RES_SCATTERED = 1u << 31 };   /* its residual blocks do not lie together in the pool: not stageable in LDS */
/* kernel-side failures latched in OhEngine's error word (reported by oh_engine_sync and everything that waits for the stream) */
enum { OH_KE_OK = 0, OH_KE_ROW_TIMEOUT = 1, OH_KE_DAG_TIMEOUT = 2 };
struct OhPrepCounts { uint32_t n_pu, n_mc_luma, n_mc_chroma, n_tu, n_intra, n_sub, n_ictu, n_levels; };

struct DevFrame {
    OhPicParams pp;
    DevPlanes   cur;              /* reconstruction / deblock buffer of the current picture   */
    DevPlanes   out;              /* SAO output (== cur when SAO is disabled)                 */
    DevPlanes   refs[OH_MAX_REFS];/* final planes of the reference pictures                   */

    const OhPu      *pu;
    const DevMcJob  *mc_luma, *mc_chroma;
    const OhWeights *wp;
    const DevTu     *tu;              /* sorted by size: blocks of log2 size 2+k are tu[tu_first[k] .. +tu_cnt[k]) */
    uint32_t        tu_first[4], tu_cnt[4];
    const int16_t   *coeffs;
    const uint32_t  *sparse;          /* records of the OH_TUF_SPARSE blocks (ohevc_frame.h), may be null            */
    const DevCross  *cross;           /* cross-component prediction blocks, n_cross of them                          */
    uint32_t        n_cross;
    const OhScalingList *scaling;     /* may be null: flat matrices only                                            */
    int16_t         *res;             /* residual pool (deferred adds of intra blocks)        */
    const DevIntra  *intra;
    const DevIntraCtu *ictu;          /* CTUs with intra blocks in wavefront order             */
    const uint32_t  *sub_start;       /* sub-level ranges into intra[]                         */
    const uint32_t  *sub_small;       /* per sub-level: its first sub_small[] blocks are <= 8x8 (slot path)      */
    const uint32_t  *lvl_start;       /* wavefront level ranges into ictu[]                    */
    const uint8_t   *vbs, *hbs;       /* boundary strengths of the 4-sample edge segments, FOUR to the byte: entry i in bits 2 (i & 3) of byte i >> 2 */
    const int8_t    *qp;
    const uint8_t   *is_pcm;          /* may be null                                          */
    const OhDeblockCtb *db;
    const OhSaoCtb  *sao;             /* may be null                                          */
    uint32_t n_pu, n_mc_luma, n_mc_chroma, n_tu, n_intra;
    uint64_t *dbg;                    /* diagnostic builds only (OH_STAMPS), null otherwise       */
    /* the raw lists as handed over (include/ohevc_frame.h) — inputs of the preparation kernels (prep.hip), which write the
     * lists above (mc_luma .. lvl_start) — and their scratch */
    const OhTu       *tu_raw;
    const OhIntra    *intra_raw;
    const OhIntraCtu *ictu_raw;
    const uint32_t   *tu_sparse, *tu_cross;   /* per OhTu, may be null */
    const uint8_t    *is_intra;               /* min-PU map, constrained intra pred only */
    uint32_t n_ictu, n_sub, n_levels, n_wp, n_sparse, ref_ok;      /* ref_ok: bit i = reference slot i holds a usable picture */
    uint32_t coeffs_present;
    uint64_t n_coeff;
    const uint32_t *pu_off;           /* [2][n_pu + 1]: first luma / chroma block of every PU (counted by the host while sizing the lists) */
    uint8_t  *tu_keep;                /* [n_tu]: luma block whose residual a cross-component block reads */
    uint32_t *tu_cursor;              /* [0..3] blocks per size, [4..7] scatter cursors, [8] cross-component blocks, [9] their cursor */
    uint32_t *intra_perm;             /* [n_intra]: position of block i after the <= 8x8-first partition of its sub-level */
    uint32_t *sub_small_w;            /* = sub_small, writable */
    uint32_t *ctu_seen;               /* [CTBs]: index + 1 of the schedule entry the CTU heads (one at most), 0: no intra block in it */
    uint32_t *row_progress;           /* [CTB rows]: CTUs of the row finished by intra_rows_kernel */
    uint32_t *ctu_aux;                /* [n_ictu]: OH_AUX_*: wave passes of the entry, the neighbour CTUs its blocks read, flags */
    /* the schedule as a dependency graph (intra.hip: intra_dag_kernel / intra_direct_kernel run a whole picture in ONE launch):
     * entry k starts once the entries ctu_wait[4k .. 4k+3] (left, up-left, up, up-right neighbour CTUs of a LOWER level; ~0: none)
     * have set their ctu_done word; written by prep_intra_wait, ctu_done cleared before every execution */
    uint32_t *ctu_wait;               /* [n_ictu][4] */
    uint32_t *ctu_done;               /* [n_ictu] */
    uint32_t *ctu_lvl;                /* [n_ictu]: wavefront level of the entry (scratch of the preparation) */
    uint32_t *ctu_order;              /* [n_ictu]: dispatch order of intra_direct_kernel: entries on dependency chains first (prep_intra_order) */
    uint32_t *err_word;               /* pinned host memory of the engine: [0] OH_KE_* of the first kernel-side failure, [1] cur_pic, [2] entry / row */
    int32_t   cur_pic_id;             /* for the error report */
    void     *summary;                /* DevSummary + DevLevelStat[n_levels] */
    void     *summary_host;           /* pinned host copy, written by prep_finish (no D2H copy on the stream) */
    uint32_t *zero_ptr; uint32_t zero_words;      /* scratch the preparation starts from cleared (prep_clear) */
    /* 16x16 CTBs with horizontally subsampled chroma only (else null): the first chroma column of every CTB on the two rows of
     * every horizontal chroma edge as it was BEFORE the horizontal-edge pass — the reference's CTB driver lets the SAO of the
     * left neighbour read exactly that (deblock.hip: oh_sao_stale_*) */
    uint16_t *sao_stale;              /* [plane - 1][edge row / 8][CTB column][p0, q0] */
    const uint8_t *sao_pending;       /* OhFrame.sao_pending (pictures not decoded in raster order), or null: closed form in sao.hip */
};
static __host__ __device__ inline bool oh_sao_stale_config(const OhPicParams *p) { return p->log2_ctb_size == 4 && (p->chroma_format_idc == 1 || p->chroma_format_idc == 2); }
static __host__ __device__ inline size_t oh_sao_stale_index(const OhPicParams *p, int plane, int edge_row8, int ctb_col)
{
    const int ctbw = (p->width + 15) >> 4, hc = p->height >> (p->chroma_format_idc == 1);
    return (((size_t)(plane - 1) * (size_t)((hc >> 3) + 1) + (size_t)edge_row8) * (size_t)ctbw + (size_t)ctb_col) * 2;
}

/* one plane of the SHVC up-sampling pass (kernel argument) */
struct OhUpPlane {
    const void *src; int32_t sstride, w_bl, h_bl;
    void *dst;       int32_t dstride, w_el, h_el;
    int32_t left, right_end_h, right_end_v, top, bottom_end;
    int32_t scale_x, add_x, scale_y, add_y, y_bias;
};

/* a batch of mutually independent pictures of one geometry: every pass is ONE launch over all of them
 * (kernel argument; the picture is picked by a grid dimension) */
#define OH_MAX_BATCH 32
struct OhBatch { const DevFrame *f[OH_MAX_BATCH]; };

#endif
