/*
 * ohevc_hip.h — C ABI of the MI355X block-reconstruction engine (libohevc_hip.so).
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.  The engine owns
 * the decoded pictures in HBM (the DPB lives on the GPU), replays recorded work lists
 * (ohevc_frame.h) as whole-picture passes and hands pictures back on request:
 *
 *   reference side (kept intact)                       engine entry point
 *   ------------------------------------------------   -----------------------------------------
 *   set_sps -> ff_hevc_dsp_init/ff_hevc_pred_init       oh_engine_create        hevc.c:421-423
 *   ff_hevc_set_new_ref -> ff_thread_get_buffer         oh_pic_alloc            hevc_refs.c:75-147
 *   hls_slice_data end / frame done                     oh_frame_submit         hevc.c:3017-3090
 *   libOpenHevcDecode before exposing a picture         oh_engine_sync          openHevcWrapper.c:130-153
 *   libOpenHevcGetOutput(Cpy), calc_md5                 oh_pic_download         openHevcWrapper.c:338-398, hevc.c:4146-4169
 *   ff_hevc_output_frame crop + GetOutputCpy            oh_pic_download_window  hevc_refs.c:248-254, openHevcWrapper.c:353-398
 *   ff_hevc_unref_frame                                 oh_pic_free             hevc_refs.c:45-65
 *
 * All functions return 0 on success and a negative OH_E_* code otherwise; the table slots of the
 * reference cannot fail (void returns), so recording never reports errors —
 * they surface here, at submit / sync.  There is NO CPU fallback: every entry point fails with
 * OH_E_HIP when no gfx950 device is usable.
 *
 * Threading: an engine is a single-submitter object (one HIP stream, one picture table); calls on the same engine
 * must be serialised by the caller (the reference's frame threads each record into their own OhRecorder and hand the
 * finished work list to the thread that owns the engine).  Different engines are independent.
 */
#ifndef OHEVC_HIP_H
#define OHEVC_HIP_H

#include <stddef.h>
#include <stdint.h>
#include "ohevc_frame.h"

#ifdef __cplusplus
extern "C" {
#endif

enum {
    OH_OK = 0,
    OH_E_HIP = -1,        /* HIP runtime error (see oh_engine_last_error) */
    OH_E_ARG = -2,        /* invalid argument / inconsistent work list    */
    OH_E_NOMEM = -3,
    OH_E_UNSUPPORTED = -4 /* outside what the path covers (e.g. up-sampling of >8-bit pictures) */
};

enum OhPass {             /* indices into oh_engine_pass_times()          */
    OH_PASS_INTER = 0, OH_PASS_RESIDUAL, OH_PASS_INTRA, OH_PASS_DEBLOCK_V, OH_PASS_DEBLOCK_H,
    OH_PASS_SAO, OH_N_PASSES
};

typedef struct OhEngine   OhEngine;
typedef struct OhDevFrame OhDevFrame;

int  oh_engine_create(OhEngine **out, int device);
/* page-locked host memory: a work list whose arrays ALL lie in blocks from oh_host_alloc() may carry OH_FRAME_PINNED (ohevc_frame.h) and
 * is then copied to the GPU by DMA from where it lies (no staging copy on the host thread) */
void *oh_host_alloc(size_t bytes);
void  oh_host_free(void *p);
/* same, but every kernel and copy is enqueued on the caller's stream (hipStream_t passed as
 * void*; e.g. torch.cuda.current_stream().cuda_stream) so that RCCL collectives issued by the
 * caller on that stream are ordered with the engine's passes.  The stream is not destroyed. */
int  oh_engine_create_on_stream(OhEngine **out, int device, void *hip_stream);
void oh_engine_destroy(OhEngine *e);
const char *oh_engine_last_error(const OhEngine *e);
int  oh_engine_sync(OhEngine *e);                       /* wait for everything enqueued so far */

/* pictures (device resident).  ids are small integers, stable until oh_pic_free */
int oh_pic_alloc(OhEngine *e, const OhPicParams *p, int *pic_id);
int oh_pic_free(OhEngine *e, int pic_id);
/* A picture is two buffers of oh_pic_bytes()/2 bytes each: the reconstruction/deblock planes
 * ("half 0") and the SAO output planes ("half 1"); layout inside a half: oh_pic_half_layout().
 * oh_pic_wrap builds a picture over caller-owned device memory (two 256-byte aligned buffers, e.g.
 * rows of a torch uint8 tensor) so that finished reference pictures of several GPUs sit
 * contiguously and can be handed to one RCCL all-gather without a copy;
 * oh_pic_final_half tells which half holds the finished picture, oh_pic_set_final_half marks a
 * half as finished after the caller filled it (broadcast / all-gather receive side). */
size_t oh_pic_bytes(const OhPicParams *p);
int oh_pic_wrap(OhEngine *e, const OhPicParams *p, void *half0, void *half1, size_t half_bytes, int *pic_id);
int oh_pic_final_half(OhEngine *e, int pic_id);                 /* 0, 1 or a negative error */
int oh_pic_set_final_half(OhEngine *e, int pic_id, int half);
/* planes: tightly described by byte strides, sample type uint8_t (8 bit) or uint16_t (>8 bit) */
int oh_pic_upload(OhEngine *e, int pic_id, const uint8_t *const planes[3], const ptrdiff_t strides[3]);
int oh_pic_download(OhEngine *e, int pic_id, uint8_t *const planes[3], const ptrdiff_t strides[3]);

/* output side: the picture inside its conformance window, packed — what ff_hevc_output_frame's
 * plane-pointer offsets (hevc_refs.c:248-254) followed by libOpenHevcGetOutputCpy's row copies (openHevcWrapper.c:353-398)
 * hand to the application.  Plane c receives ((height - top - bottom) >> vshift) rows of ((width - left - right) >> hshift)
 * samples starting at (left >> hshift, top >> vshift); strides[] are the destination pitches in bytes (>= the row size).
 * The copy goes through a pinned staging buffer of the engine. */
typedef struct OhWindow { int32_t left, right, top, bottom; } OhWindow;     /* luma samples, as HEVCWindow after hevc_ps.c scaled it */
int oh_pic_download_window(OhEngine *e, int pic_id, const OhWindow *win, uint8_t *const planes[3], const ptrdiff_t strides[3]);
/* The same fetch in two halves, for a decoder whose threads share one engine behind a lock (an engine is driven by one thread at a
 * time): oh_pic_download_start — under that lock, microseconds — enqueues the device-to-host copies behind the batch that finished the
 * picture; oh_download_finish may then run on ANY thread WITHOUT the lock, while other threads hand pictures over: it waits for the
 * copies, moves the rows into the caller's planes (OHEVC_FETCH_THREADS copy helpers, default 3, plus the calling thread) and returns
 * the staging buffer.  Every started download must be finished (also to release it: planes = NULL gives OH_E_ARG after the wait).
 * With frame threads this takes the fetch of a released picture off the decoder's serial path (openHevcWrapper.c:338-398 is called
 * between two libOpenHevcDecode calls while the workers keep decoding). */
typedef struct OhDownload OhDownload;
int oh_pic_download_start(OhEngine *e, int pic_id, const OhWindow *win, OhDownload **out);
int oh_download_finish(OhEngine *e, OhDownload *d, uint8_t *const planes[3], const ptrdiff_t strides[3]);

/* MD5 of the three planes of n finished pictures computed on the GPU — the digests of the decoded-picture-hash SEI (hevc.c:4146-4162,
 * calc_md5 hevc.c:4623-4638: whole coded planes, rows packed, little-endian samples).  digests: n x 3 x 16 bytes.  Waits for the engine. */
int oh_pics_md5(OhEngine *e, const int *pic_ids, int n, uint8_t *digests);

/* SHVC inter-layer reference picture: resample the finished base-layer picture src_pic into
 * the enhancement-layer picture dst_pic, bit-exact with the reference's whole-picture slot
 * HEVCDSPContext.upsample_base_layer_frame (hevcdsp_template.c:2164-2438, call site hevc.c:3241).
 * 8-bit 4:2:0 only, like that routine (OH_E_UNSUPPORTED otherwise).  u: oh_upsample_setup(). */
int oh_pic_upsample(OhEngine *e, int dst_pic, int src_pic, const OhUpsample *u);
/* The same resampling for a LIST of CTBs of the enhancement-layer picture (raster addresses for CTBs of 1 << log2_ctb_size luma
 * samples; the rest of dst_pic is left as it is): the granularity of the reference's default build, which up-samples a CTB when
 * a prediction unit first reads the inter-layer reference there (ACTIVE_PU_UPSAMPLING hevc.h:117, ff_upsample_block
 * hevc_filter.c:1370-1426, is_upsampled[]).  A decoder collects the CTBs its picture's inter-layer PUs touch and issues one call
 * before the picture's work list.  Same samples as the reference's CTB path wherever that path and its whole-picture slot agree:
 * no scaled reference layer offsets and no phase alignment (tests/test_upsample_vs_ref.py); offsets are refused
 * (OH_E_UNSUPPORTED — use oh_pic_upsample).  The motion-field half of that path (ff_upscale_mv_block, hevc_filter.c:1311-1368)
 * feeds merge / AMVP derivation (hevc_mvs.c), host work by SURVEY §8, and stays with the host decoder. */
int oh_pic_upsample_ctbs(OhEngine *e, int dst_pic, int src_pic, const OhUpsample *u, int log2_ctb_size, const uint32_t *ctb_addrs, int n);

/* work lists.  OhFrame.cur_pic / ref_pics[] hold engine picture ids.
 * upload copies every array to HBM (after it returns the host arrays may be reused);
 * execute enqueues passes 1-5 on the engine stream and may be called repeatedly on the same
 * device frame (the coefficient pool is never modified). */
int oh_frame_upload(OhEngine *e, const OhFrame *f, OhDevFrame **out);
/* n work lists at once: the host part and the copy per list, ONE set of preparation launches for all of them (what a batch of
 * independent pictures that will run as one oh_frames_execute should use).  All or nothing: on error no list stays uploaded. */
int oh_frames_upload(OhEngine *e, const OhFrame *const *fs, int n, OhDevFrame **out);
int oh_frame_execute(OhEngine *e, OhDevFrame *df);
/* n mutually independent pictures (none is a reference of another one; same OhPicParams): every pass
 * is ONE launch over all of them, which is how pictures of independent sequences / GOPs (the reference's
 * frame threads, pthread_frame.c) fill the GPU while each picture's own dependency chain is short of it */
int oh_frames_execute(OhEngine *e, OhDevFrame *const *dfs, int n);      /* n == 0: nothing to do, OH_OK */
int oh_frame_free(OhEngine *e, OhDevFrame *df);          /* waits for the engine stream, then frees */
/* same without the wait: legal right after the last oh_frame(s)_execute of df was ENQUEUED — the device memory is recycled in
 * stream order (a decoder that uploads, executes and forgets one work list per picture never blocks on the GPU) */
int oh_frame_release(OhEngine *e, OhDevFrame *df);
/* the boundary-strength grids of an uploaded work list as the deblock pass will read them: the ones handed over, or — with
 * OhFrame.bs_in — the ones the engine derived from the motion field at upload (hevc_filter.c:584-941).
 * bytes: size of each destination, at most oh_bs_size() is copied */
int oh_frame_download_bs(OhEngine *e, OhDevFrame *df, uint8_t *vbs, uint8_t *hbs, size_t bytes);
int oh_frame_submit(OhEngine *e, const OhFrame *f);     /* upload + execute + release in stream order: nothing of the list stays behind */
/* what the engine holds for work lists: out[0] device arenas alive (pooled or holding a list), [1] their bytes, [2] of them free in the pool,
 * [3] pinned staging buffers, [4] their bytes, [5] work lists waiting for a deferred free.  A decoder that submits and forgets one list
 * per picture sees all of them level off after a few pictures, however long the stream. */
int oh_engine_memory(OhEngine *e, uint64_t out[6]);

/* per-pass device time of the executes since the last reset, measured with HIP events on the
 * engine stream (enable = 1 costs two event records per pass; enable = 2 additionally brackets every
 * launch of the intra pass, see oh_engine_intra_launch_times).  ms[] and launches[] hold OH_N_PASSES
 * entries: accumulated milliseconds and number of timed executes. */
int oh_engine_profile(OhEngine *e, int enable);
int oh_engine_pass_times(OhEngine *e, double *ms, uint64_t *executes, int reset);
/* the intra pass is one launch per CTU-wavefront level: summed per-launch device time and launch
 * count (events bracket every launch in profile mode); read after oh_engine_pass_times() */
int oh_engine_intra_launch_times(OhEngine *e, double *ms, uint64_t *launches, int reset);

/* where the HOST time of the hand-over path went since the last reset: accumulated wall milliseconds and calls per OH_HT_* slot
 * (nested: OH_HT_UPLOAD contains the OH_HT_UPLOAD_* parts, OH_HT_EXECUTE contains OH_HT_EXECUTE_WAIT_PREP) */
enum OhHostTime { OH_HT_UPLOAD = 0, OH_HT_UPLOAD_COUNT, OH_HT_UPLOAD_ARENA, OH_HT_UPLOAD_STAGE_WAIT, OH_HT_UPLOAD_MEMCPY, OH_HT_UPLOAD_ENQUEUE,
                  OH_HT_EXECUTE, OH_HT_EXECUTE_WAIT_PREP, OH_HT_RELEASE, OH_N_HOST_TIMES };
int oh_engine_host_times(OhEngine *e, double *ms, uint64_t *calls, int n, int reset);
uint64_t oh_engine_upload_bytes(OhEngine *e, int reset);      /* bytes of work lists copied host -> device since the last reset */

/* the stream everything is enqueued on (hipStream_t as void*), for callers that need to order
 * their own work (RCCL broadcasts of reference pictures) against the engine */
void *oh_engine_stream(OhEngine *e);
/* device address / geometry of a picture's FINAL planes (valid after the frame that writes it was
 * executed), for zero-copy exchange between GPUs; stride in samples */
int oh_pic_device_planes(OhEngine *e, int pic_id, void *planes[3], int32_t stride[3], int32_t width[3], int32_t height[3]);

/* diagnostics only: in-kernel cycle stamps of a -DOH_STAMPS build (tools/intra_stamps.py) */
int oh_debug_read(OhEngine *e, uint64_t *out, size_t n_u64);

#ifdef __cplusplus
}
#endif
#endif
