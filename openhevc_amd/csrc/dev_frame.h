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
    const OhIntra   *intra;
    const OhIntraCtu *ictu;          /* CTUs with intra blocks in wavefront order             */
    const uint32_t  *sub_start;       /* sub-level ranges into intra[]                         */
    const uint8_t   *vbs, *hbs;
    const int8_t    *qp;
    const uint8_t   *is_pcm;          /* may be null                                          */
    const OhDeblockCtb *db;
    const OhSaoCtb  *sao;             /* may be null                                          */
    uint32_t n_pu, n_tiles, n_tu, n_intra;
};

#endif
