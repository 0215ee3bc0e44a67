/*
 * ref_wrapper_hip_unit.c — the wrapper half of the drop-in library oracle/_ref/libopenhevc_hip.so.
 *
 * The reference's gpac/modules/openhevc_dec/openHevcWrapper.c compiled where it lies (the #include below; nothing copied), with the
 * one change SURVEY.md §8(b) "where the GPU sync must sit" asks for, made with the preprocessor: before libOpenHevcGetOutput /
 * libOpenHevcGetOutputCpy (openHevcWrapper.c:338-398) expose the frame libOpenHevcDecode released, its samples are brought from the
 * engine's picture in HBM into the host planes the wrapper's AVFrame points at (oh_hooked_fetch_output: conformance window and all —
 * the one strided device-to-pinned copy per plane of oh_pic_download_window); libOpenHevcClose also closes the engine.  Every other
 * entry point (Init, StartDecoder, Decode, GetPictureInfo[Cpy], CopyExtraData, the Set* family, Flush, FlushSVC, Version) is the
 * reference's code unchanged: the library exports exactly the 18 symbols of openHevcWrapper.h:79-98.
 *
 * Built from reference sources, so by the rules of this repository it lives under oracle/_ref/ (container-built, git-ignored,
 * shipped to the GPU box as a binary) — but it is not a checker: it is INTEGRATION.md executed, the library a user of the
 * reference links instead of libLibOpenHevcWrapper to get the MI355X engine under the unchanged API.
 */
#include "libavcodec/avcodec.h"

int  oh_hooked_fetch_output(AVFrame *out);       /* ref_hooked_unit.c, engine build */
int  oh_hooked_engine_open(void);
void oh_hooked_engine_close(void);
int  oh_hooked_engine_sync(void);

#define libOpenHevcGetOutput    oh_host_GetOutput
#define libOpenHevcGetOutputCpy oh_host_GetOutputCpy
#define libOpenHevcDecode       oh_host_Decode
#define libOpenHevcStartDecoder oh_host_StartDecoder
#define libOpenHevcClose        oh_host_Close
#include "openHevcWrapper.c"
#undef libOpenHevcGetOutput
#undef libOpenHevcGetOutputCpy
#undef libOpenHevcDecode
#undef libOpenHevcStartDecoder
#undef libOpenHevcClose

static const void *g_fetched;                    /* the output frame whose samples are already in its host planes */

static void fetch(OpenHevc_Handle openHevcHandle)
{
    OpenHevcWrapperContexts *ctxs = (OpenHevcWrapperContexts *)openHevcHandle;
    AVFrame *picture = ctxs->wraper[ctxs->display_layer]->picture;
    if (!picture->data[0] || g_fetched == picture->data[0])
        return;
    if (oh_hooked_fetch_output(picture) == 0)
        g_fetched = picture->data[0];
}

int libOpenHevcStartDecoder(OpenHevc_Handle openHevcHandle)
{
    OpenHevcWrapperContexts *ctxs = (OpenHevcWrapperContexts *)openHevcHandle;
    const int rc = oh_host_StartDecoder(openHevcHandle);
    for (int i = 0; rc == 1 && i < ctxs->nb_decoders; i++)
        if ((ctxs->wraper[i]->c->active_thread_type & FF_THREAD_FRAME) && (ctxs->wraper[i]->c->active_thread_type & FF_THREAD_SLICE)) {
            /* frame threads: every worker records its own picture (per-thread binding, ordered hand-over); slice / wavefront threads:
             * the workers adopt the picture in flight.  Both at once: a slice worker could not tell WHICH picture in flight is its
             * own (the table slots carry no context): INTEGRATION.md 7b */
            fprintf(stderr, "libopenhevc_hip: frame AND slice threads together are not supported by the recording hooks (thread type 1 or 2, not 4)\n");
            return -1;
        }
    if (rc == 1 && oh_hooked_engine_open() != 0)
        return -1;                               /* no MI355X: there is no CPU fallback behind this library */
    return rc;
}

int libOpenHevcDecode(OpenHevc_Handle openHevcHandle, const unsigned char *buff, int au_len, int64_t pts)
{
    g_fetched = NULL;                            /* the frame exposed so far is released by this call (openHevcWrapper.h: valid until the next Decode) */
    const int got = oh_host_Decode(openHevcHandle, buff, au_len, pts);
    if (got == 0 && au_len == 0 && oh_hooked_engine_sync() != 0)      /* flushed: nothing is left to output, and nothing is left running */
        return -1;
    return got;
}

int libOpenHevcGetOutput(OpenHevc_Handle openHevcHandle, int got_picture, OpenHevc_Frame *openHevcFrame)
{
    if (got_picture)
        fetch(openHevcHandle);
    return oh_host_GetOutput(openHevcHandle, got_picture, openHevcFrame);
}

int libOpenHevcGetOutputCpy(OpenHevc_Handle openHevcHandle, int got_picture, OpenHevc_Frame_cpy *openHevcFrame)
{
    if (got_picture)
        fetch(openHevcHandle);
    return oh_host_GetOutputCpy(openHevcHandle, got_picture, openHevcFrame);
}

void libOpenHevcClose(OpenHevc_Handle openHevcHandle)
{
    oh_host_Close(openHevcHandle);
    oh_hooked_engine_close();
}
