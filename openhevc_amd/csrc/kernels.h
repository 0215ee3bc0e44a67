/* kernels.h — launchers of kernels.hip (internal to libohevc_hip.so) */
#ifndef OHEVC_KERNELS_H
#define OHEVC_KERNELS_H

#include <hip/hip_runtime.h>
#include "dev_frame.h"

extern "C" {
int  ohk_init(void);
void ohk_inter(const DevFrame *df, const OhPicParams *p, uint32_t n_luma, uint32_t n_chroma, hipStream_t st);
void ohk_residual(const DevFrame *df, const OhPicParams *p, uint32_t n_tu, hipStream_t st);
void ohk_intra_level(const DevFrame *df, const OhPicParams *p, const OhIntraLaunch *l, hipStream_t st);
void ohk_deblock(const DevFrame *df, const OhPicParams *p, int horiz, hipStream_t st);
void ohk_sao(const DevFrame *df, const OhPicParams *p, hipStream_t st);
}
#endif
