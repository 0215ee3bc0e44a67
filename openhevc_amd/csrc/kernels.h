/* kernels.h — launchers of the per-pass kernel files (mc / residual / intra / deblock / sao / upsample .hip) (internal to libohevc_hip.so) */
#ifndef OHEVC_KERNELS_H
#define OHEVC_KERNELS_H

#include <hip/hip_runtime.h>
#include "dev_frame.h"

/* one MD5 chain: a plane's packed rows (md5.hip) */
struct OhMd5Job {
    const void *base;            /* first sample of the plane */
    uint32_t    pitch;           /* bytes between rows */
    uint32_t    row_bytes;       /* bytes hashed per row */
    uint32_t    rows;
    uint32_t    pad;
};

extern "C" {
int  ohk_init(void);
void ohk_md5(const OhMd5Job *jobs, int n, void *digests, hipStream_t st);
void ohk_inter(const OhBatch *B, int n, const OhPicParams *p, uint32_t max_luma, uint32_t max_chroma, hipStream_t st);
void ohk_residual(const OhBatch *B, int n, const OhPicParams *p, const uint32_t max_cnt[4], hipStream_t st);
void ohk_cross(const OhBatch *B, int n, const OhPicParams *p, uint32_t max_cross, hipStream_t st);
void ohk_intra_level(const OhBatch *B, int n, const OhPicParams *p, const OhIntraLaunch *l, uint32_t max_ctu, hipStream_t st);
void ohk_intra_rows(const OhBatch *B, int n, const OhPicParams *p, const OhIntraLaunch *l, uint32_t spin_limit, hipStream_t st);
void ohk_intra_dag_reset(const OhBatch *B, int n, uint32_t max_ictu, uint32_t *tickets, hipStream_t st);
void ohk_intra_dag(const OhBatch *B, int n, const OhPicParams *p, const OhIntraLaunch *l, uint32_t max_ictu, uint32_t *ticket,
                   uint32_t spin_limit, hipStream_t st);
void ohk_intra_direct(const OhBatch *B, int n, const OhPicParams *p, uint32_t max_ictu, uint32_t *ticket, uint32_t spin_limit, hipStream_t st);
void ohk_deblock(const OhBatch *B, int n, const OhPicParams *p, int horiz, hipStream_t st);
void ohk_upsample_plane(const OhUpPlane *a, int taps, int tw, int th, const uint32_t *list, int n_list, hipStream_t st);
void ohk_sao(const OhBatch *B, int n, const OhPicParams *p, hipStream_t st);
/* one array of a page-locked work list: where it lies on the host, where it goes in the arena */
struct OhPullSeg { const void *src; void *dst; uint64_t bytes; };
void ohk_pull(const OhPullSeg *segs, int nseg, size_t total_bytes, hipStream_t st);
void ohk_prepare(const OhBatch *B, int nb, const OhPrepCounts *max_counts, uint32_t max_mc_runs, uint32_t max_cross, hipStream_t st);
void ohk_bs_derive(const OhPicParams *p, const void *mvf, const void *cbf, const void *call_log2, const void *ctb_flags,
                   int across_tiles, void *vbs, void *hbs, hipStream_t st);
}
#endif
