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

/* one <=16x16 luma tile of a PU: the unit one wave interpolates */
struct DevTile {
    uint32_t pu;                  /* index into DevFrame.pu                                  */
    uint8_t  ox, oy, w, h;        /* offset inside the PU and size, luma samples             */
};

/* intra block as the kernel wants it: OhIntra with the TU index resolved to its residual offset */
struct DevIntra {
    uint16_t x, y;
    uint8_t  c_idx, log2_size, mode, avail;
    uint32_t res_off;             /* int16 offset into DevFrame.res, OH_NO_COEFF when cbf == 0  */
};

/* OhIntraCtu plus the span of the residual pool its blocks use (staged in LDS when it fits) */
struct DevIntraCtu {
    uint32_t sub_first;
    uint16_t n_sub, ctu;
    uint32_t res_lo, res_cnt;     /* int16 elements; res_cnt == 0: blocks read their residual from HBM */
};

struct DevFrame {
    OhPicParams pp;
    DevPlanes   cur;              /* reconstruction / deblock buffer of the current picture   */
    DevPlanes   out;              /* SAO output (== cur when SAO is disabled)                 */
    DevPlanes   refs[OH_MAX_REFS];/* final planes of the reference pictures                   */

    const OhPu      *pu;
    const DevTile   *tiles;
    const OhWeights *wp;
    const OhTu      *tu;
    const int16_t   *coeffs;
    int16_t         *res;             /* residual pool (deferred adds of intra blocks)        */
    const DevIntra  *intra;
    const DevIntraCtu *ictu;          /* CTUs with intra blocks in wavefront order             */
    const uint32_t  *sub_start;       /* sub-level ranges into intra[]                         */
    const uint8_t   *vbs, *hbs;
    const int8_t    *qp;
    const uint8_t   *is_pcm;          /* may be null                                          */
    const OhDeblockCtb *db;
    const OhSaoCtb  *sao;             /* may be null                                          */
    uint32_t n_pu, n_tiles, n_tu, n_intra;
    uint64_t *dbg;                    /* diagnostic builds only (OH_STAMPS), null otherwise       */
};

#endif
