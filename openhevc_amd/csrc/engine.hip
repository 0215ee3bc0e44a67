/*
 * engine.hip — host side of the MI355X engine: device-resident pictures (the DPB lives in HBM),
 * work-list upload, pass scheduling on one HIP stream, per-pass event timing.
 * C ABI in include/ohevc_hip.h.  No CPU fallback exists: without a usable device every entry
 * point returns OH_E_HIP.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ohevc_hip.h"
#include "dev_frame.h"
#include "kernels.h"

namespace {

struct Pic {
    bool        used = false;
    OhPicParams p{};
    void       *base = nullptr;          /* one allocation: planes A (recon/deblock) then B (SAO out) */
    bool        owned = true;            /* false: caller-owned memory (oh_pic_wrap)                   */
    void       *a[3] = {}, *b[3] = {};
    int32_t     stride[3] = {}, w[3] = {}, h[3] = {};
    bool        final_b = false;         /* which buffer holds the finished picture */
    uint32_t    gen = 0;                 /* bumped whenever the id is (re)installed: uploaded work lists remember it */
    uint64_t    done_seq = 0;            /* the batch that last reconstructed the picture (OhEngine::batch_ev ring); 0: written by something else, or never */
};

struct EventSet { hipEvent_t ev[OH_N_PASSES + 1]; int n_frames = 1; };

} // namespace

struct OhDevFrame {
    void      *arena = nullptr;
    size_t     arena_bytes = 0;
    DevFrame  *d = nullptr;
    OhPicParams p{};
    uint32_t   n_mc_luma = 0, n_mc_chroma = 0, n_tu = 0, n_intra = 0;
    uint32_t   tu_cnt[4] = { 0, 0, 0, 0 };
    uint32_t   n_cross = 0;
    bool       has_sao = false;
    int        cur_pic = -1;      /* engine id of the picture the list reconstructs */
    uint32_t   cur_gen = 0;
    /* reference slots as uploaded: picture id, its generation, and which half DevFrame.refs[] points at — looked up again at
     * every execute (a reference finished or received AFTER the upload moves to its other half: oh_pic_set_final_half, SAO) */
    int        ref_id[OH_MAX_REFS];
    uint32_t   ref_gen[OH_MAX_REFS];
    uint8_t    ref_half[OH_MAX_REFS];
    uint16_t   ref_used = 0;      /* bit i: some PU predicts from slot i */
    hipEvent_t ready = nullptr;   /* recorded on the copy stream behind the work list's H2D copy, preparation kernels and summary */
    bool       waited = false;    /* the engine stream already waits for `ready` */
    void      *sum_host = nullptr;/* pinned: DevSummary + DevLevelStat[n_levels] as the preparation kernels left them */
    size_t     sum_bytes = 0;
    bool       sum_pooled = false, summary_read = false;
    void      *sum_dev = nullptr; /* the summary in the arena */
    OhPrepCounts cnt{};           /* sizes of the preparation launches */
    uint32_t   prep_err = 0, n_levels = 0;
    uint32_t   intra_area64 = 0, max_passes = 0;   /* from the summary: samples of the intra blocks / 64; wave passes of the heaviest CTU */
    const struct OhEngine *owner = nullptr;   /* picture ids and arenas belong to one engine */
    struct Level {                        /* one wavefront level: what sizes the launch that runs it */
        uint32_t n_ctu, max_items, max_sub, max_res;
        bool     staged;                  /* every CTU has its residual span contiguous (stageable in LDS) */
        uint64_t sum_items, sum_sub;
    };
    std::vector<Level> levels;
};

/* helper threads for the one host copy of the hand-over (the work list into a pinned staging buffer): the lists of a 4K picture are
 * ~4 MB, 0.22 ms for one thread — most of what the hand-over costs the decoder's thread.  The calling thread keeps a share. */
struct CopyJob { char *dst; const char *src; size_t n; bool pack; };
static void pack_bs(uint8_t *dst, const uint8_t *src, size_t n);
struct CopyPool {
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv, done_cv;
    const std::vector<CopyJob> *jobs = nullptr;
    int pending = 0;
    uint64_t gen = 0;
    bool stop = false;
    static void run_share(const std::vector<CopyJob> &jobs, int share, int shares)
    {
        for (size_t i = (size_t)share; i < jobs.size(); i += (size_t)shares) {
            const CopyJob &j = jobs[i];
            if (j.pack) pack_bs((uint8_t *)j.dst, (const uint8_t *)j.src, j.n);
            else memcpy(j.dst, j.src, j.n);
        }
    }
    void worker(int k)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::vector<CopyJob> *my;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                my = jobs;
            }
            run_share(*my, k + 1, (int)th.size() + 1);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) done_cv.notify_one();
            }
        }
    }
    void start(int n) { for (int k = 0; k < n; k++) th.emplace_back(&CopyPool::worker, this, k); }
    void run(const std::vector<CopyJob> &j)                    /* returns when every job has been copied */
    {
        if (th.empty()) { run_share(j, 0, 1); return; }
        {
            std::lock_guard<std::mutex> lk(mu);
            jobs = &j; pending = (int)th.size(); gen++;
        }
        cv.notify_all();
        run_share(j, 0, (int)th.size() + 1);
        std::unique_lock<std::mutex> lk(mu);
        done_cv.wait(lk, [&] { return pending == 0; });
    }
    ~CopyPool()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv.notify_all();
        for (auto &t : th) t.join();
    }
};

struct OhEngine {
    int         device = 0;
    int         n_cu = 256;              /* compute units of the device */
    hipStream_t stream = nullptr;
    bool        own_stream = true;
    std::vector<Pic> pics;
    std::string err;
    int         profile = 0;             /* 0 off, 1 events between passes, 2 also around every intra launch */
    std::vector<EventSet> ev_pool, ev_pending;
    std::vector<hipEvent_t> lev_pool, lev_pending;   /* per-launch event pairs of the intra pass (profile mode) */
    double      intra_launch_ms = 0;
    uint64_t    intra_launches = 0;
    double      pass_ms[OH_N_PASSES] = {};
    uint64_t    executes = 0;
    std::vector<OhDevFrame *> deferred;
    uint32_t    pic_gen = 0;
    /* upload path: pinned staging buffers and device arenas are recycled (hipHostMalloc / hipMalloc cost milliseconds);
     * a staging buffer is busy until the H2D copy that reads it has passed `done` */
    struct Stage { void *p; size_t bytes; hipEvent_t done; bool busy; };
    std::vector<Stage> stages;
    /* work lists travel on their own stream so that the copy of picture n+1 overlaps the passes of picture n; the engine stream
     * waits for a list's `ready` event before the first kernel that reads it, and a recycled arena is not overwritten before
     * the engine stream has passed the event recorded when it was released */
    hipStream_t copy_stream = nullptr;
    struct Arena { void *p; size_t bytes; hipEvent_t free_ev; };
    std::vector<Arena> arenas;               /* free device arenas */
    uint64_t    arenas_alive = 0, arena_bytes_alive = 0;      /* device arenas allocated and not freed (pooled or holding a work list): oh_engine_memory */
    std::vector<hipEvent_t> sync_events;     /* pool of timing-disabled events (ready / free_ev) */
    std::vector<void *> sum_pool;            /* pinned OH_SUMMARY_BLOCK-byte blocks */
    double      host_ms[OH_N_HOST_TIMES] = {};   /* where the host time of the hand-over path goes (oh_engine_host_times) */
    uint64_t    host_calls[OH_N_HOST_TIMES] = {};
    uint64_t    up_bytes = 0;                    /* bytes of work lists sent over PCIe since the last reset */
    uint64_t   *dbg = nullptr;           /* diagnostics (OHEVC_STAMPS=1 + a -DOH_STAMPS build) */
    /* what a kernel can tell the host when it cannot go on (the table slots and the passes have no error channel: hevcdsp.h's slots
     * return void): four words of pinned host memory, [0] OH_KE_* of the first failure, [1] picture id, [2] schedule entry / CTB
     * row; read by everything that waits for the stream (kernel_error) */
    /* a ring of events, one behind every executed batch: a download of a finished picture waits for ITS batch (on the download
     * stream), not for everything enqueued since — a decoder fetches the picture it outputs while the passes of the pictures it
     * submitted later keep running */
    enum { BATCH_RING = 64 };
    hipEvent_t  batch_ev[BATCH_RING] = {};
    uint64_t    batch_seq = 0;
    hipStream_t dl_stream = nullptr;
    /* ticket counters of the one-launch intra forms: a ring of pairs (direct, staged) in HBM; a batch uses the next pair, cleared on
     * the stream in front of its launches (a pair comes round again 128 batches later: long after its launch has drained) */
    enum { TICKET_RING = 128, TICKET_WORDS = 2 * OH_MAX_BATCH * 32 };      /* per batch: (direct, staged) x pictures, a cache line each (intra.hip: OH_TICKET_STRIDE) */
    uint32_t   *tickets = nullptr;
    uint64_t    ticket_seq = 0;
    CopyPool   *copiers = nullptr;      /* created with the first hand-over (OHEVC_COPY_THREADS helpers, default 2) */
    /* output fetch (oh_pic_download_start / oh_download_finish): its own pinned buffers and copy helpers, usable while another thread
     * drives the engine */
    std::mutex  dl_mu, dl_copy_mu;
    std::vector<Stage *> dl_stages;
    CopyPool   *dl_copiers = nullptr;   /* created with the first fetch (OHEVC_FETCH_THREADS helpers, default 3) */
    std::vector<CopyJob> dl_jobs;
    double      dl_wait_ms = 0, dl_copy_ms = 0;     /* OHEVC_FETCH_TIMING: where the time of the fetches went (printed when the engine is destroyed) */
    uint64_t    dl_count = 0;
    std::vector<CopyJob> copy_jobs;
    uint32_t   *kerr = nullptr;
    uint32_t    spin_limit = 1u << 22;   /* polls (with s_sleep between them, ~1 s in all) before a waiting workgroup gives up; OHEVC_SPIN_LIMIT */
};

#define HIPCHK(e, call)                                                                           \
    do {                                                                                          \
        hipError_t rc_ = (call);                                                                  \
        if (rc_ != hipSuccess) {                                                                  \
            char buf_[512];                                                                       \
            snprintf(buf_, sizeof(buf_), "%s failed: %s (%s:%d)", #call, hipGetErrorString(rc_), __FILE__, __LINE__); \
            (e)->err = buf_;                                                                      \
            return OH_E_HIP;                                                                      \
        }                                                                                         \
    } while (0)

#define FAIL(e, code, ...)                                                                        \
    do {                                                                                          \
        char buf_[512];                                                                           \
        snprintf(buf_, sizeof(buf_), __VA_ARGS__);                                                \
        (e)->err = buf_;                                                                          \
        return (code);                                                                            \
    } while (0)

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

/* after a wait on the engine's stream: did a kernel latch a failure (intra.hip: dag_latch_error)?  Reported once, then cleared. */
static int kernel_error(OhEngine *e)
{
    if (!e->kerr || !e->kerr[0])
        return OH_OK;
    const uint32_t code = e->kerr[0], pic = e->kerr[1], where = e->kerr[2];
    e->kerr[0] = 0;
    FAIL(e, OH_E_HIP, code == OH_KE_ROW_TIMEOUT
             ? "intra pass (CTB rows in one launch): picture %u, CTB row %u gave up waiting for the row above; the picture's samples are not valid"
             : "intra pass (one launch per picture): picture %u, schedule entry %u gave up waiting for a neighbour CTU; the picture's samples are not valid",
         pic, where);
}
struct HostTimer {                       /* adds the scope's wall time to one slot of OhEngine::host_ms */
    OhEngine *e; int slot; std::chrono::steady_clock::time_point t0;
    HostTimer(OhEngine *e_, int slot_);
    ~HostTimer();
};
enum { OH_MAX_STAGES = 48, OH_SUMMARY_BLOCK = 32768 };             /* pinned staging buffers per engine before the host is made to wait */

HostTimer::HostTimer(OhEngine *e_, int slot_) : e(e_), slot(slot_), t0(std::chrono::steady_clock::now()) {}
HostTimer::~HostTimer()
{
    e->host_ms[slot] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    e->host_calls[slot]++;
}

extern "C" uint64_t oh_engine_upload_bytes(OhEngine *e, int reset)
{
    if (!e)
        return 0;
    const uint64_t v = e->up_bytes;
    if (reset) e->up_bytes = 0;
    return v;
}

extern "C" int oh_engine_host_times(OhEngine *e, double *ms, uint64_t *calls, int n, int reset)
{
    if (!e || n < 0)
        return OH_E_ARG;
    for (int i = 0; i < n && i < OH_N_HOST_TIMES; i++) {
        if (ms) ms[i] = e->host_ms[i];
        if (calls) calls[i] = e->host_calls[i];
    }
    if (reset)
        for (int i = 0; i < OH_N_HOST_TIMES; i++) { e->host_ms[i] = 0; e->host_calls[i] = 0; }
    return OH_OK;
}

static int engine_create(OhEngine **out, int device, hipStream_t ext, bool use_ext)
{
    if (!out)
        return OH_E_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        fprintf(stderr, "ohevc_hip: no usable HIP device (count=%d, requested %d); there is no CPU fallback\n", n, device);
        return OH_E_HIP;
    }
    OhEngine *e = new OhEngine();
    e->device = device;
    bool ok = hipSetDevice(device) == hipSuccess;
    if (ok && use_ext) {
        e->stream = ext;
        e->own_stream = false;
    } else if (ok) {
        ok = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) == hipSuccess;
    }
    if (ok)
        ok = hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking) == hipSuccess;     /* (highest priority for this stream: measured, no difference) */
    if (!ok || ohk_init() != 0) {
        fprintf(stderr, "ohevc_hip: device %d initialisation failed\n", device);
        delete e;
        return OH_E_HIP;
    }
    if (hipDeviceGetAttribute(&e->n_cu, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || e->n_cu <= 0)
        e->n_cu = 256;
    if (hipHostMalloc((void **)&e->kerr, 4 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
        fprintf(stderr, "ohevc_hip: no pinned memory for the kernels' error word\n");
        delete e;
        return OH_E_HIP;
    }
    memset(e->kerr, 0, 4 * sizeof(uint32_t));
    if (hipMalloc((void **)&e->tickets, (size_t)OhEngine::TICKET_WORDS * OhEngine::TICKET_RING * sizeof(uint32_t)) != hipSuccess) {
        fprintf(stderr, "ohevc_hip: no device memory for the ticket counters\n");
        delete e;
        return OH_E_HIP;
    }
    if (const char *sl = getenv("OHEVC_SPIN_LIMIT"))          /* tests: make a waiting workgroup give up at once */
        e->spin_limit = (uint32_t)std::max(1l, atol(sl));
    if (getenv("OHEVC_STAMPS")) {
        const size_t bytes = (16 + 4000 * 16) * sizeof(uint64_t);
        if (hipMalloc((void **)&e->dbg, bytes) == hipSuccess)
            (void)hipMemset(e->dbg, 0, bytes);
    }
    *out = e;
    return OH_OK;
}

/* diagnostics: copies the in-kernel stamp records (see intra.hip, OH_STAMPS) and clears them */
extern "C" int oh_debug_read(OhEngine *e, uint64_t *out, size_t n_u64)
{
    if (!e || !e->dbg || !out)
        return OH_E_ARG;
    const size_t total = 16 + 4000 * 16;
    if (n_u64 > total) n_u64 = total;
    HIPCHK(e, hipStreamSynchronize(e->stream));
    HIPCHK(e, hipMemcpy(out, e->dbg, n_u64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIPCHK(e, hipMemset(e->dbg, 0, total * sizeof(uint64_t)));
    return OH_OK;
}

/* page-locked host memory for work lists handed over with OH_FRAME_PINNED (the GPU reads them by DMA where they lie) */
extern "C" void *oh_host_alloc(size_t bytes)
{
    void *p = nullptr;
    return hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}
extern "C" void oh_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" int oh_engine_create(OhEngine **out, int device) { return engine_create(out, device, nullptr, false); }
extern "C" int oh_engine_create_on_stream(OhEngine **out, int device, void *hip_stream)
{
    return engine_create(out, device, (hipStream_t)hip_stream, true);
}

extern "C" int oh_engine_memory(OhEngine *e, uint64_t out[6])
{
    if (!e || !out)
        return OH_E_ARG;
    uint64_t sb = 0;
    for (auto &c : e->stages) sb += c.bytes;
    out[0] = e->arenas_alive; out[1] = e->arena_bytes_alive; out[2] = e->arenas.size();
    out[3] = e->stages.size(); out[4] = sb; out[5] = e->deferred.size();
    return OH_OK;
}

extern "C" const char *oh_engine_last_error(const OhEngine *e) { return e ? e->err.c_str() : "no engine"; }
extern "C" void *oh_engine_stream(OhEngine *e) { return e ? (void *)e->stream : nullptr; }

static hipEvent_t sync_event_get(OhEngine *e)
{
    hipEvent_t ev = nullptr;
    if (!e->sync_events.empty()) { ev = e->sync_events.back(); e->sync_events.pop_back(); return ev; }
    return hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess ? ev : nullptr;
}
static void sync_event_put(OhEngine *e, hipEvent_t ev)
{
    if (!ev) return;
    if (e->sync_events.size() < 4096) e->sync_events.push_back(ev); else (void)hipEventDestroy(ev);
}

/* arenas go back to the engine's pool.  in_flight: kernels enqueued on the engine stream may still read the arena — an event
 * recorded there now tells the copy stream when the next work list may overwrite it. */
static void free_dev_frame(OhEngine *e, OhDevFrame *df, bool in_flight = false)
{
    if (!df)
        return;
    if (e && df->ready)
        sync_event_put(e, df->ready);
    if (df->sum_host) {
        if (e && df->sum_pooled && e->sum_pool.size() < 4096) e->sum_pool.push_back(df->sum_host);
        else (void)hipHostFree(df->sum_host);
    }
    if (df->arena) {
        if (e && e->arenas.size() < 1024) {
            hipEvent_t fe = nullptr;
            if (in_flight && (fe = sync_event_get(e)) != nullptr && hipEventRecord(fe, e->stream) != hipSuccess) {
                sync_event_put(e, fe);
                fe = nullptr;
            }
            if (in_flight && !fe)
                (void)hipStreamSynchronize(e->stream);           /* no event to be had: wait instead */
            e->arenas.push_back({ df->arena, df->arena_bytes, fe });
        } else {
            if (in_flight) (void)hipStreamSynchronize(e->stream);
            (void)hipFree(df->arena);
            if (e) { e->arenas_alive--; e->arena_bytes_alive -= df->arena_bytes; }
        }
    }
    delete df;
}

extern "C" int oh_engine_sync(OhEngine *e)
{
    if (!e)
        return OH_E_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (OhDevFrame *df : e->deferred)
        free_dev_frame(e, df);
    e->deferred.clear();
    return kernel_error(e);
}

extern "C" void oh_engine_destroy(OhEngine *e)
{
    if (!e)
        return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    if (e->copy_stream) (void)hipStreamSynchronize(e->copy_stream);
    for (OhDevFrame *df : e->deferred)
        free_dev_frame(e, df);
    for (Pic &p : e->pics)
        if (p.used && p.base && p.owned)
            (void)hipFree(p.base);
    for (auto &s : e->ev_pool)
        for (auto &ev : s.ev) (void)hipEventDestroy(ev);
    for (auto &ev : e->lev_pool) (void)hipEventDestroy(ev);
    for (auto &ev : e->lev_pending) (void)hipEventDestroy(ev);
    for (auto &s : e->ev_pending)
        for (auto &ev : s.ev) (void)hipEventDestroy(ev);
    for (auto &c : e->stages) { (void)hipEventDestroy(c.done); (void)hipHostFree(c.p); }
    for (auto &a : e->arenas) { if (a.free_ev) (void)hipEventDestroy(a.free_ev); (void)hipFree(a.p); }
    for (auto &ev : e->sync_events) (void)hipEventDestroy(ev);
    for (void *b : e->sum_pool) (void)hipHostFree(b);
    if (e->dl_count && getenv("OHEVC_FETCH_TIMING"))
        fprintf(stderr, "ohevc engine: %llu output fetches, per fetch %.2f ms waiting for the picture's passes and its device-to-host copy, %.2f ms moving the rows into the caller's planes\n",
                (unsigned long long)e->dl_count, e->dl_wait_ms / e->dl_count, e->dl_copy_ms / e->dl_count);
    delete e->copiers;
    delete e->dl_copiers;
    for (auto *c : e->dl_stages) { (void)hipEventDestroy(c->done); (void)hipHostFree(c->p); delete c; }
    if (e->kerr) (void)hipHostFree(e->kerr);
    if (e->tickets) (void)hipFree(e->tickets);
    for (auto &ev : e->batch_ev) if (ev) (void)hipEventDestroy(ev);
    if (e->dl_stream) { (void)hipStreamSynchronize(e->dl_stream); (void)hipStreamDestroy(e->dl_stream); }
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->own_stream)
        (void)hipStreamDestroy(e->stream);
    delete e;
}

/* ---------------- pictures ---------------- */
static int check_params(OhEngine *e, const OhPicParams *p)
{
    if (!p || p->width <= 0 || p->height <= 0 || p->width > 16384 || p->height > 16384)
        FAIL(e, OH_E_ARG, "bad picture size");
    if (p->bit_depth != 8 && p->bit_depth != 9 && p->bit_depth != 10 && p->bit_depth != 12)
        FAIL(e, OH_E_UNSUPPORTED, "bit depth %d not supported (8/9/10/12: what the reference's wrapper can hand out, openHevcWrapper.c)", p->bit_depth);
    if (p->chroma_format_idc < 0 || p->chroma_format_idc > 3)
        FAIL(e, OH_E_ARG, "chroma_format_idc %d out of range", p->chroma_format_idc);
    if (p->log2_ctb_size < 4 || p->log2_ctb_size > 6 || p->log2_min_cb_size < 3 || p->log2_min_cb_size > p->log2_ctb_size ||
        p->log2_min_tb_size < 2 || p->log2_min_tb_size > 5 || p->log2_min_pu_size != p->log2_min_cb_size - 1)
        FAIL(e, OH_E_ARG, "bad block size parameters");
    if ((uint64_t)oh_ctb_width(p) * (uint64_t)oh_ctb_height(p) > 65535u)
        FAIL(e, OH_E_UNSUPPORTED, "%d x %d picture with %d x %d CTBs has more than 65535 CTBs: the intra schedule (OhIntraCtu.ctu) counts them in 16 bits; use larger CTBs",
             p->width, p->height, 1 << p->log2_ctb_size, 1 << p->log2_ctb_size);
    if (p->width % (1 << p->log2_min_cb_size) || p->height % (1 << p->log2_min_cb_size))
        FAIL(e, OH_E_ARG, "picture size must be a multiple of the minimum CB size");
    return OH_OK;
}

/* layout of one picture allocation: half 0 = planes A, half 1 = planes B */
static size_t pic_layout(const OhPicParams *p, Pic *pic, size_t off[6])
{
    uint64_t o[3] = { 0, 0, 0 };
    const size_t half = (size_t)oh_pic_half_layout(p, pic->stride, o);
    for (int c = 0; c < (p->chroma_format_idc ? 3 : 1); c++) {
        pic->w[c] = p->width >> oh_hshift(p, c);
        pic->h[c] = p->height >> oh_vshift(p, c);
        off[c] = (size_t)o[c];
        off[3 + c] = half + (size_t)o[c];
    }
    return 2 * half;
}

extern "C" size_t oh_pic_bytes(const OhPicParams *p)
{
    Pic tmp;
    size_t off[6];
    return p ? pic_layout(p, &tmp, off) : 0;
}

static int pic_install(OhEngine *e, const OhPicParams *p, void *half0, void *half1, bool owned, int *pic_id)
{
    Pic pic;
    size_t off[6];
    pic.used = true;
    pic.p = *p;
    pic.owned = owned;
    pic.gen = ++e->pic_gen;
    size_t half = pic_layout(p, &pic, off) / 2;
    pic.base = half0;
    for (int c = 0; c < (p->chroma_format_idc ? 3 : 1); c++) {
        pic.a[c] = (char *)half0 + off[c];
        pic.b[c] = (char *)half1 + (off[3 + c] - half);
    }
    size_t id = 0;
    while (id < e->pics.size() && e->pics[id].used)
        id++;
    if (id == e->pics.size())
        e->pics.push_back(pic);
    else
        e->pics[id] = pic;
    *pic_id = (int)id;
    return OH_OK;
}

extern "C" int oh_pic_alloc(OhEngine *e, const OhPicParams *p, int *pic_id)
{
    if (!e || !pic_id)
        return OH_E_ARG;
    int rc = check_params(e, p);
    if (rc)
        return rc;
    HIPCHK(e, hipSetDevice(e->device));
    void *mem = nullptr;
    HIPCHK(e, hipMalloc(&mem, oh_pic_bytes(p)));
    return pic_install(e, p, mem, (char *)mem + oh_pic_bytes(p) / 2, true, pic_id);
}

extern "C" int oh_pic_wrap(OhEngine *e, const OhPicParams *p, void *half0, void *half1, size_t half_bytes, int *pic_id)
{
    if (!e || !pic_id || !half0 || !half1)
        return OH_E_ARG;
    int rc = check_params(e, p);
    if (rc)
        return rc;
    if (half_bytes < oh_pic_bytes(p) / 2 || ((uintptr_t)half0 & 255) || ((uintptr_t)half1 & 255))
        FAIL(e, OH_E_ARG, "oh_pic_wrap: each half needs %zu bytes, 256-byte aligned", oh_pic_bytes(p) / 2);
    return pic_install(e, p, half0, half1, false, pic_id);
}

static Pic *get_pic(OhEngine *e, int id)
{
    if (id < 0 || (size_t)id >= e->pics.size() || !e->pics[id].used)
        return nullptr;
    return &e->pics[id];
}

extern "C" int oh_pic_free(OhEngine *e, int pic_id)
{
    if (!e)
        return OH_E_ARG;
    Pic *p = get_pic(e, pic_id);
    if (!p)
        FAIL(e, OH_E_ARG, "oh_pic_free: unknown picture %d", pic_id);
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (p->owned)
        HIPCHK(e, hipFree(p->base));
    *p = Pic();
    return OH_OK;
}

extern "C" int oh_pic_final_half(OhEngine *e, int pic_id)
{
    if (!e)
        return OH_E_ARG;
    Pic *p = get_pic(e, pic_id);
    return p ? (p->final_b ? 1 : 0) : OH_E_ARG;
}

extern "C" int oh_pic_set_final_half(OhEngine *e, int pic_id, int half)
{
    if (!e)
        return OH_E_ARG;
    Pic *p = get_pic(e, pic_id);
    if (!p || (half != 0 && half != 1))
        FAIL(e, OH_E_ARG, "oh_pic_set_final_half: bad picture or half");
    p->final_b = half == 1;
    p->done_seq = 0;                      /* filled from outside (an exchange): a download stays behind the whole engine stream */
    return OH_OK;
}

/* SHVC inter-layer reference: upsample_base_layer_frame (hevcdsp_template.c:2164-2438, called at hevc.c:3241) */
static OhEngine::Stage *stage_acquire(OhEngine *e, size_t bytes);

/* SHVC up-sampling of tiles of the enhancement-layer picture: ctbs == nullptr: the whole picture (the reference's whole-picture
 * slot, hevc.c:3241); else the listed CTBs (raster addresses, CTB size 1 << log2_ctb) — the on-demand granularity of the
 * reference's default build (ff_upsample_block, hevc_filter.c:1370-1426: a CTB is up-sampled when a PU first predicts from it) */
static int upsample_tiles(OhEngine *e, int dst_pic, int src_pic, const OhUpsample *u, int log2_ctb, const uint32_t *ctbs, int n_ctbs, const char *who)
{
    if (!e || !u)
        return OH_E_ARG;
    Pic *el = get_pic(e, dst_pic), *bl = get_pic(e, src_pic);
    if (!el || !bl || el == bl)
        FAIL(e, OH_E_ARG, "%s: bad picture ids", who);
    el->done_seq = 0;
    if (el->p.bit_depth != 8 || bl->p.bit_depth != 8 || el->p.chroma_format_idc != 1 || bl->p.chroma_format_idc != 1)
        FAIL(e, OH_E_UNSUPPORTED, "%s: the reference's up-sampler is written for 8-bit 4:2:0 (byte edge buffers, shift 12)", who);
    const int w_el = el->p.width, h_el = el->p.height, w_bl = bl->p.width, h_bl = bl->p.height;
    if (u->win_left < 0 || u->win_right < 0 || u->win_top < 0 || u->win_bottom < 0 || u->win_left + u->win_right >= w_el ||
        u->win_top + u->win_bottom >= h_el || u->scale_x_lum <= 0 || u->scale_y_lum <= 0 || u->scale_x_cr <= 0 || u->scale_y_cr <= 0)
        FAIL(e, OH_E_ARG, "%s: bad window / scale", who);
    if (u->scale_x_lum > 65536 || u->scale_y_lum > 65536 || u->scale_x_cr > 65536 || u->scale_y_cr > 65536)
        FAIL(e, OH_E_UNSUPPORTED, "%s: the enhancement layer is smaller than the base layer (scale > 1): not a spatial-scalability configuration", who);
    HIPCHK(e, hipSetDevice(e->device));
    const int tile = ctbs ? 1 << log2_ctb : 64;
    const uint32_t *dlist = nullptr;
    OhEngine::Stage *sg = nullptr;
    if (ctbs) {
        if (log2_ctb < 4 || log2_ctb > 6 || n_ctbs < 0)
            FAIL(e, OH_E_ARG, "%s: CTB size / count", who);
        if (!n_ctbs)
            return OH_OK;
        const uint32_t n_ctb = (uint32_t)(((w_el + tile - 1) / tile) * ((h_el + tile - 1) / tile));
        for (int i = 0; i < n_ctbs; i++)
            if (ctbs[i] >= n_ctb)
                FAIL(e, OH_E_ARG, "%s: CTB address %u of %u", who, ctbs[i], n_ctb);
        sg = stage_acquire(e, (size_t)n_ctbs * sizeof(uint32_t));       /* pinned and mapped: the kernels read the list there */
        if (!sg)
            FAIL(e, OH_E_NOMEM, "%s: no staging buffer", who);
        memcpy(sg->p, ctbs, (size_t)n_ctbs * sizeof(uint32_t));
        dlist = (const uint32_t *)sg->p;
    }
    void *const *src = bl->final_b ? bl->b : bl->a;
    OhUpPlane a;
    /* luma: BL rows = min(BL height, EL height) (:2220); x clipped to [left, right_end] inclusive (:2223) */
    a.src = src[0]; a.sstride = bl->stride[0]; a.w_bl = w_bl; a.h_bl = h_bl <= h_el ? h_bl : h_el;
    a.dst = el->a[0]; a.dstride = el->stride[0]; a.w_el = w_el; a.h_el = h_el;
    a.left = u->win_left; a.right_end_h = w_el - u->win_right; a.right_end_v = w_el - u->win_right;
    a.top = u->win_top; a.bottom_end = h_el - u->win_bottom;
    a.scale_x = u->scale_x_lum; a.add_x = u->add_x_lum; a.scale_y = u->scale_y_lum; a.add_y = u->add_y_lum; a.y_bias = 0;
    ohk_upsample_plane(&a, 8, tile, tile, dlist, n_ctbs, e->stream);
    /* chroma: BL rows = max(BL height, EL chroma height) >> 1 (:2317-2320); x clipped to [left, right_end - 1] (:2324);
     * the vertical position carries the -4 of :2384.  A CTB's chroma tile has the same index in a grid of half-size tiles. */
    const int wc_el = w_el >> 1, hc_el = h_el >> 1;
    for (int c = 1; c <= 2; c++) {
        a.src = src[c]; a.sstride = bl->stride[c]; a.w_bl = w_bl >> 1; a.h_bl = (h_bl > hc_el ? h_bl : hc_el) >> 1;
        if (a.h_bl > bl->h[c]) a.h_bl = bl->h[c];
        a.dst = el->a[c]; a.dstride = el->stride[c]; a.w_el = wc_el; a.h_el = hc_el;
        a.left = u->win_left >> 1; a.right_end_v = wc_el - (u->win_right >> 1); a.right_end_h = a.right_end_v - 1;
        a.top = u->win_top >> 1; a.bottom_end = hc_el - (u->win_bottom >> 1);
        a.scale_x = u->scale_x_cr; a.add_x = u->add_x_cr; a.scale_y = u->scale_y_cr; a.add_y = u->add_y_cr; a.y_bias = 4;
        ohk_upsample_plane(&a, 4, tile >> 1, tile >> 1, dlist, n_ctbs, e->stream);
    }
    HIPCHK(e, hipGetLastError());
    if (sg) {
        HIPCHK(e, hipEventRecord(sg->done, e->stream));
        sg->busy = true;
    }
    el->final_b = false;                                   /* the resampled picture is a finished picture in half 0 */
    return OH_OK;
}

extern "C" int oh_pic_upsample(OhEngine *e, int dst_pic, int src_pic, const OhUpsample *u)
{
    return upsample_tiles(e, dst_pic, src_pic, u, 6, nullptr, 0, "oh_pic_upsample");
}

extern "C" int oh_pic_upsample_ctbs(OhEngine *e, int dst_pic, int src_pic, const OhUpsample *u, int log2_ctb_size, const uint32_t *ctb_addrs, int n)
{
    if (!ctb_addrs && n)
        return OH_E_ARG;
    /* the reference's block path positions by its block driver and its x2 / x1.5 slots ignore the phase: with scaled reference
     * layer offsets or phase alignment it produces OTHER samples than the whole-picture slot (tests/test_upsample_vs_ref.py
     * records both).  This entry point is the whole-picture arithmetic per CTB, so it stands for the block path only where the
     * reference's two paths agree. */
    if (u && (u->win_left || u->win_right || u->win_top || u->win_bottom))
        FAIL(e, OH_E_UNSUPPORTED, "oh_pic_upsample_ctbs: scaled reference layer offsets — the reference's CTB path and its whole-picture slot differ there; use oh_pic_upsample");
    static const uint32_t none = 0;
    return upsample_tiles(e, dst_pic, src_pic, u, log2_ctb_size, n ? ctb_addrs : &none, n, "oh_pic_upsample_ctbs");
}

extern "C" int oh_pic_upload(OhEngine *e, int pic_id, const uint8_t *const planes[3], const ptrdiff_t strides[3])
{
    if (!e || !planes || !strides)
        return OH_E_ARG;
    Pic *p = get_pic(e, pic_id);
    if (!p)
        FAIL(e, OH_E_ARG, "oh_pic_upload: unknown picture %d", pic_id);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t bpp = p->p.bit_depth > 8 ? 2 : 1;
    for (int c = 0; c < (p->p.chroma_format_idc ? 3 : 1); c++)
        HIPCHK(e, hipMemcpy2DAsync(p->a[c], (size_t)p->stride[c] * bpp, planes[c], (size_t)strides[c], (size_t)p->w[c] * bpp,
                                   (size_t)p->h[c], hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    p->final_b = false;
    p->done_seq = 0;
    return OH_OK;
}

extern "C" int oh_pic_download(OhEngine *e, int pic_id, uint8_t *const planes[3], const ptrdiff_t strides[3])
{
    if (!e || !planes || !strides)
        return OH_E_ARG;
    Pic *p = get_pic(e, pic_id);
    if (!p)
        FAIL(e, OH_E_ARG, "oh_pic_download: unknown picture %d", pic_id);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t bpp = p->p.bit_depth > 8 ? 2 : 1;
    for (int c = 0; c < (p->p.chroma_format_idc ? 3 : 1); c++)
        HIPCHK(e, hipMemcpy2DAsync(planes[c], (size_t)strides[c], p->final_b ? p->b[c] : p->a[c], (size_t)p->stride[c] * bpp,
                                   (size_t)p->w[c] * bpp, (size_t)p->h[c], hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    { const int ke = kernel_error(e); if (ke) return ke; }     /* a kernel that gave up: these samples are not the picture */
    return OH_OK;
}

/* n boundary strengths (0..2, one per byte: hevc_filter.c's vertical_bs / horizontal_bs) -> (n + 3) / 4 bytes, entry i in bits
 * 2 (i & 3) of byte i >> 2 — the form the deblock pass reads.  Four bytes per multiply: the 2-bit fields land in the top byte. */
static void pack_bs(uint8_t *dst, const uint8_t *src, size_t n)
{
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
        uint32_t x;
        memcpy(&x, src + i, 4);
        dst[i >> 2] = (uint8_t)(((x & 0x03030303u) * 0x01041040u) >> 24);
    }
    if (i < n) {
        uint32_t v = 0;
        for (size_t k = i; k < n; k++) v |= (uint32_t)(src[k] & 3) << ((k & 3) * 2);
        dst[i >> 2] = (uint8_t)v;
    }
}

/* a pinned buffer of at least `bytes` whose previous copy has completed (uploads and downloads share the pool) */
static OhEngine::Stage *stage_acquire(OhEngine *e, size_t bytes)
{
    OhEngine::Stage *sg = nullptr;
    for (auto &c : e->stages) {
        if (c.busy && hipEventQuery(c.done) == hipSuccess)
            c.busy = false;
        if (!c.busy && c.bytes >= bytes && (!sg || c.bytes < sg->bytes))
            sg = &c;
    }
    if (sg)
        return sg;
    /* every buffer is busy: beyond OH_MAX_STAGES wait for one whose copy is furthest along instead of pinning more host memory
     * (this is what throttles a host that hands pictures over faster than PCIe and the passes take them) */
    if (e->stages.size() >= OH_MAX_STAGES) {
        for (auto &c : e->stages)
            if (c.busy && c.bytes >= bytes) {
                if (hipEventSynchronize(c.done) != hipSuccess)
                    return nullptr;
                c.busy = false;
                return &c;
            }
    }
    OhEngine::Stage c;
    c.bytes = align_up(bytes, (size_t)4 << 20); c.busy = false; c.p = nullptr; c.done = nullptr;
    if (hipHostMalloc(&c.p, c.bytes, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&c.done, hipEventDisableTiming) != hipSuccess) {
        if (c.p) (void)hipHostFree(c.p);
        return nullptr;
    }
    e->stages.push_back(c);
    return &e->stages.back();
}

/* Output side of the path: the conformance-window crop of ff_hevc_output_frame (hevc_refs.c:248-254:
 * plane pointers advanced by (left >> hshift, top >> vshift)) followed by libOpenHevcGetOutputCpy's packed row copies
 * (openHevcWrapper.c:353-398: `height >> vshift` rows of `(width >> hshift) << pixel_shift` bytes, width / height = the cropped
 * size).  One strided device-to-pinned copy per plane, one wait, then the rows go to the caller's pitches. */
/* The output fetch in two halves, for a decoder whose threads share ONE engine behind a lock (the drop-in library: frame-thread
 * workers hand pictures over while the application's thread fetches the one that was released):
 *   oh_pic_download_start   (under the caller's engine lock, microseconds) validates, takes a pinned DOWNLOAD staging buffer — a list
 *                           of its own, guarded by its own mutex — and enqueues the strided device-to-host copies behind the batch
 *                           that finished the picture (download stream), then records an event;
 *   oh_download_finish      (NO engine lock needed, any thread) waits for that event, copies the rows into the caller's planes with
 *                           the download copy helpers, and gives the staging buffer back.
 * oh_pic_download_window is the two in a row. */
struct OhDownload {
    OhEngine::Stage *sg;
    size_t row[3], rows[3], off[3];
    int np;
};

extern "C" int oh_pic_download_start(OhEngine *e, int pic_id, const OhWindow *win, OhDownload **out)
{
    if (!e || !win || !out)
        return OH_E_ARG;
    *out = nullptr;
    Pic *p = get_pic(e, pic_id);
    if (!p)
        FAIL(e, OH_E_ARG, "oh_pic_download_window: unknown picture %d", pic_id);
    const int W = p->p.width - win->left - win->right, H = p->p.height - win->top - win->bottom;
    if (win->left < 0 || win->right < 0 || win->top < 0 || win->bottom < 0 || W <= 0 || H <= 0)
        FAIL(e, OH_E_ARG, "oh_pic_download_window: window (%d,%d,%d,%d) leaves nothing of %dx%d", win->left, win->right, win->top, win->bottom,
             p->p.width, p->p.height);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t bpp = p->p.bit_depth > 8 ? 2 : 1;
    OhDownload d;
    d.np = p->p.chroma_format_idc ? 3 : 1;
    size_t total = 0;
    for (int c = 0; c < d.np; c++) {
        const int hs = oh_hshift(&p->p, c), vs = oh_vshift(&p->p, c);
        d.row[c] = (size_t)(W >> hs) * bpp; d.rows[c] = (size_t)(H >> vs);
        d.off[c] = total; total += align_up(d.row[c] * d.rows[c], 256);
    }
    {   /* a free download buffer that fits, else a new one (a decoder has one or two fetches in flight) */
        std::lock_guard<std::mutex> lk(e->dl_mu);
        d.sg = nullptr;
        for (auto *c : e->dl_stages)
            if (!c->busy && c->bytes >= total && (!d.sg || c->bytes < d.sg->bytes)) d.sg = c;
        if (!d.sg) {
            OhEngine::Stage *c = new OhEngine::Stage();
            c->bytes = align_up(total, (size_t)4 << 20); c->busy = false; c->p = nullptr; c->done = nullptr;
            if (hipHostMalloc(&c->p, c->bytes, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&c->done, hipEventDisableTiming) != hipSuccess) {
                if (c->p) (void)hipHostFree(c->p);
                delete c;
                FAIL(e, OH_E_NOMEM, "hipHostMalloc(%zu) failed", total);
            }
            e->dl_stages.push_back(c);
            d.sg = c;
        }
        d.sg->busy = true;
    }
    /* a picture a batch of this engine finished, and whose event is still in the ring: the copies run on the download stream behind
     * THAT batch; anything else (uploaded, up-sampled, received from another GPU, long ago): behind everything on the engine stream */
    hipStream_t dl = e->stream;
    if (p->done_seq && e->batch_seq - p->done_seq < OhEngine::BATCH_RING - 1) {
        if (!e->dl_stream) HIPCHK(e, hipStreamCreateWithFlags(&e->dl_stream, hipStreamNonBlocking));
        HIPCHK(e, hipStreamWaitEvent(e->dl_stream, e->batch_ev[p->done_seq % OhEngine::BATCH_RING], 0));
        dl = e->dl_stream;
    }
    hipError_t he = hipSuccess;
    for (int c = 0; c < d.np && he == hipSuccess; c++) {
        const int hs = oh_hshift(&p->p, c), vs = oh_vshift(&p->p, c);
        const uint8_t *src = (const uint8_t *)(p->final_b ? p->b[c] : p->a[c]) + ((size_t)(win->top >> vs) * p->stride[c] + (size_t)(win->left >> hs)) * bpp;
        he = hipMemcpy2DAsync((char *)d.sg->p + d.off[c], d.row[c], src, (size_t)p->stride[c] * bpp, d.row[c], d.rows[c], hipMemcpyDeviceToHost, dl);
    }
    if (he == hipSuccess) he = hipEventRecord(d.sg->done, dl);
    if (he != hipSuccess) {
        std::lock_guard<std::mutex> lk(e->dl_mu);
        d.sg->busy = false;
        FAIL(e, OH_E_HIP, "oh_pic_download_window: %s", hipGetErrorString(he));
    }
    *out = new OhDownload(d);
    return OH_OK;
}

extern "C" int oh_download_finish(OhEngine *e, OhDownload *d, uint8_t *const planes[3], const ptrdiff_t strides[3])
{
    if (!e || !d)
        return OH_E_ARG;
    int rc = OH_OK;
    if (!planes || !strides)
        rc = OH_E_ARG;
    for (int c = 0; c < d->np && !rc; c++)
        if (!planes[c] || (ptrdiff_t)d->row[c] > strides[c])
            rc = OH_E_ARG;
    const auto t_w0 = std::chrono::steady_clock::now();
    const hipError_t he = hipEventSynchronize(d->sg->done);    /* also when the arguments are bad: the buffer goes back only after its copy */
    const auto t_w1 = std::chrono::steady_clock::now();
    if (!rc && he != hipSuccess) rc = OH_E_HIP;
    if (!rc) rc = kernel_error(e);                             /* a kernel that gave up: these samples are not the picture */
    if (!rc) {
        std::lock_guard<std::mutex> lk(e->dl_copy_mu);         /* one fetch at a time uses the helpers */
        if (!e->dl_copiers) {
            const char *v = getenv("OHEVC_FETCH_THREADS");
            const int n = v ? atoi(v) : 3;
            e->dl_copiers = new CopyPool();
            e->dl_copiers->start(n < 0 ? 0 : (n > 15 ? 15 : n));
        }
        /* pieces of at most 1 MiB (packed planes) or single rows (pitched planes), dealt round-robin to the helpers and this thread */
        std::vector<CopyJob> &jobs = e->dl_jobs;
        jobs.clear();
        for (int c = 0; c < d->np; c++) {
            const char *src = (const char *)d->sg->p + d->off[c];
            if ((size_t)strides[c] == d->row[c]) {
                const size_t n = d->row[c] * d->rows[c], piece = (size_t)1 << 20;
                for (size_t o = 0; o < n; o += piece) jobs.push_back(CopyJob{ (char *)planes[c] + o, src + o, n - o < piece ? n - o : piece, false });
            } else {
                for (size_t y = 0; y < d->rows[c]; y++) jobs.push_back(CopyJob{ (char *)planes[c] + (ptrdiff_t)y * strides[c], src + y * d->row[c], d->row[c], false });
            }
        }
        e->dl_copiers->run(jobs);
        e->dl_wait_ms += std::chrono::duration<double, std::milli>(t_w1 - t_w0).count();
        e->dl_copy_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_w1).count();
        e->dl_count++;
    }
    {
        std::lock_guard<std::mutex> lk(e->dl_mu);
        d->sg->busy = false;
    }
    delete d;
    return rc;
}

extern "C" int oh_pic_download_window(OhEngine *e, int pic_id, const OhWindow *win, uint8_t *const planes[3], const ptrdiff_t strides[3])
{
    if (!e || !win || !planes || !strides)
        return OH_E_ARG;
    OhDownload *d = nullptr;
    const int rc = oh_pic_download_start(e, pic_id, win, &d);
    if (rc)
        return rc;
    const int rc2 = oh_download_finish(e, d, planes, strides);
    if (rc2 == OH_E_ARG)
        FAIL(e, OH_E_ARG, "oh_pic_download_window: a destination plane is missing or its pitch is smaller than a row");
    if (rc2 == OH_E_HIP)
        FAIL(e, OH_E_HIP, "oh_pic_download_window: the device-to-host copy failed");
    return rc2;
}

/* Picture hash on the GPU: the three plane digests of the reference's SEI check (hevc.c:4146-4162 over calc_md5,
 * hevc.c:4623-4638: the whole coded planes, sps->width x sps->height and the chroma sizes, packed rows) for n finished pictures in one
 * launch of one chain per (picture, plane) — md5.hip.  48 bytes per picture come back instead of the picture.  Waits for the engine
 * stream.  digests: n x 3 x 16 bytes (monochrome: planes 1, 2 zero). */
extern "C" int oh_pics_md5(OhEngine *e, const int *pic_ids, int n, uint8_t *digests)
{
    if (!e || n < 0 || (n && (!pic_ids || !digests)))
        return OH_E_ARG;
    if (!n)
        return OH_OK;
    for (int i = 0; i < n; i++)
        if (!get_pic(e, pic_ids[i]))
            FAIL(e, OH_E_ARG, "oh_pics_md5: unknown picture %d", pic_ids[i]);
    HIPCHK(e, hipSetDevice(e->device));
    const size_t jobs_bytes = align_up((size_t)n * 3 * sizeof(OhMd5Job), 256);
    OhEngine::Stage *sg = stage_acquire(e, jobs_bytes + (size_t)n * 48);
    if (!sg)
        FAIL(e, OH_E_NOMEM, "hipHostMalloc for %d picture hashes failed", n);
    OhMd5Job *jobs = (OhMd5Job *)sg->p;
    uint8_t *out = (uint8_t *)sg->p + jobs_bytes;
    int nj = 0;
    std::vector<int> slot((size_t)n * 3, -1);
    for (int i = 0; i < n; i++) {
        const Pic *p = get_pic(e, pic_ids[i]);
        const uint32_t bpp = p->p.bit_depth > 8 ? 2 : 1;
        for (int c = 0; c < (p->p.chroma_format_idc ? 3 : 1); c++) {
            OhMd5Job &j = jobs[nj];
            j.base = p->final_b ? p->b[c] : p->a[c];
            j.pitch = (uint32_t)p->stride[c] * bpp; j.row_bytes = (uint32_t)p->w[c] * bpp; j.rows = (uint32_t)p->h[c]; j.pad = 0;
            slot[(size_t)i * 3 + c] = nj++;
        }
    }
    ohk_md5(jobs, nj, out, e->stream);                        /* pinned host memory is mapped: the kernel reads the jobs and writes the digests there */
    HIPCHK(e, hipGetLastError());
    HIPCHK(e, hipStreamSynchronize(e->stream));
    { const int ke = kernel_error(e); if (ke) return ke; }     /* a kernel that gave up: these samples are not the picture */
    for (size_t k = 0; k < slot.size(); k++) {
        if (slot[k] >= 0) memcpy(digests + k * 16, out + (size_t)slot[k] * 16, 16);
        else memset(digests + k * 16, 0, 16);
    }
    return OH_OK;
}

extern "C" int oh_pic_device_planes(OhEngine *e, int pic_id, void *planes[3], int32_t stride[3], int32_t width[3], int32_t height[3])
{
    if (!e)
        return OH_E_ARG;
    Pic *p = get_pic(e, pic_id);
    if (!p)
        FAIL(e, OH_E_ARG, "unknown picture %d", pic_id);
    for (int c = 0; c < 3; c++) {
        planes[c] = p->final_b ? p->b[c] : p->a[c];
        stride[c] = p->stride[c]; width[c] = p->w[c]; height[c] = p->h[c];
    }
    return OH_OK;
}

/* ---------------- work lists ---------------- */
static void fill_planes(DevPlanes *dp, const Pic *p, bool use_b)
{
    for (int c = 0; c < 3; c++) {
        dp->p[c] = use_b ? p->b[c] : p->a[c];
        dp->stride[c] = p->stride[c];
        dp->w[c] = p->w[c];
        dp->h[c] = p->h[c];
    }
}

static bool same_geometry(const OhPicParams &a, const OhPicParams &b)
{
    return a.width == b.width && a.height == b.height && a.bit_depth == b.bit_depth && a.chroma_format_idc == b.chroma_format_idc;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * Hand-over of a work list.  The host copies the RAW lists (include/ohevc_frame.h, exactly as recorded) into one pinned
 * buffer, counts what sizes the device arena — blocks per PU, transform blocks per size: two light loops — and enqueues on the
 * copy stream:   H2D copy  ->  preparation kernels (prep.hip: validation of every index a pass kernel will follow, the
 * <= 8x8 MC block lists, the transform-size buckets, the intra block descriptors, the per-level launch statistics)  ->
 * boundary strengths from the motion field when the list carries bs_in (bs.hip)  ->  the summary back to pinned memory  ->
 * `ready`.  Nothing of it touches samples, so it overlaps the passes of the pictures before.  A malformed list is
 * reported by the first oh_frame(s)_execute that includes it (OH_E_ARG, before any of its passes is launched).
 * ------------------------------------------------------------------------------------------------------------------- */
static int check_host_side(OhEngine *e, const OhFrame *f, const Pic *cur, OhPrepCounts *cnt, uint32_t tu_cnt[4], uint32_t *n_cross,
                           bool *any_dense, uint16_t *ref_used, uint32_t *ref_ok, std::vector<uint32_t> &pu_off)
{
    const OhPicParams &p = f->p;
    *ref_ok = 0; *ref_used = 0; *n_cross = 0; *any_dense = false;
    for (int i = 0; i < OH_MAX_REFS; i++) {
        Pic *r = get_pic(e, f->ref_pics[i]);
        if (r && same_geometry(r->p, p) && r != cur)
            *ref_ok |= 1u << i;
    }
    if ((f->n_pu && !f->pu) || (f->n_wp && !f->wp) || (f->n_tu && !f->tu) || (f->n_intra && !f->intra))
        FAIL(e, OH_E_ARG, "a non-zero item count comes with a NULL array (pu / wp / tu / intra)");
    /* blocks per PU: sizes the MC block lists (prep_pu_scan repeats the sums on the GPU and validates every PU) */
    uint64_t nl = 0, nc = 0;
    const int hs = oh_hshift(&p, 1), vs = oh_vshift(&p, 1), two = p.chroma_format_idc ? 2 : 0;
    if ((uint64_t)f->n_pu * 2048 >= (1ull << 31))
        FAIL(e, OH_E_ARG, "PU list too long");
    pu_off.resize(2 * ((size_t)f->n_pu + 1));
    uint32_t *ol = pu_off.data(), *oc = ol + f->n_pu + 1;       /* running sums: where every PU's blocks start in the two lists */
    for (uint32_t i = 0; i < f->n_pu; i++) {
        const OhPu &pu = f->pu[i];
        ol[i] = (uint32_t)nl; oc[i] = (uint32_t)nc;
        nl += (uint64_t)(((pu.w + 7) >> 3) * ((pu.h + 7) >> 3));
        nc += (uint64_t)(two * ((((pu.w >> hs) + 7) >> 3) * (((pu.h >> vs) + 7) >> 3)));
        for (int l = 0; l < 2; l++)
            if (pu.ref[l] < OH_MAX_REFS) *ref_used |= (uint16_t)(1u << pu.ref[l]);
    }
    ol[f->n_pu] = (uint32_t)nl; oc[f->n_pu] = (uint32_t)nc;
    for (uint32_t i = 0; i < f->n_wp; i++)
        if (f->wp[i].log2_denom[0] > 7 || f->wp[i].log2_denom[1] > 7)
            FAIL(e, OH_E_ARG, "weights %u: log2 denominator out of range", i);
    tu_cnt[0] = tu_cnt[1] = tu_cnt[2] = tu_cnt[3] = 0;
    for (uint32_t i = 0; i < f->n_tu; i++) {                  /* launch sizes of the residual pass; is any block dense? */
        const OhTu &t = f->tu[i];
        tu_cnt[(t.log2_size - 2) & 3]++;
        *n_cross += (t.flags & OH_TUF_CROSS) != 0;
        *any_dense = *any_dense || !(t.flags & OH_TUF_SPARSE);
    }
    if (*any_dense && f->n_coeff && !f->coeffs)
        FAIL(e, OH_E_ARG, "dense transform blocks but coeffs[] is NULL");
    if (*n_cross && !f->tu_cross)
        FAIL(e, OH_E_ARG, "cross-component blocks without tu_cross[]");
    if (f->n_intra && p.constrained_intra_pred && !f->is_intra)
        FAIL(e, OH_E_ARG, "constrained_intra_pred without the is_intra map");
    if (f->n_intra) {
        /* the level table sizes the launches (grid = CTUs of the level): checked here; everything below it on the GPU */
        if (!f->level_start || !f->n_levels || !f->ictu || !f->n_ictu || !f->sub_start || !f->n_sub ||
            f->level_start[0] != 0 || f->level_start[f->n_levels] != f->n_ictu)
            FAIL(e, OH_E_ARG, "intra wavefront tables inconsistent");
        for (uint32_t l = 0; l < f->n_levels; l++)
            if (f->level_start[l] > f->level_start[l + 1])
                FAIL(e, OH_E_ARG, "intra level table not monotonic");
    }
    if (p.deblock_enabled) {
        const OhBsInputs *bi = f->bs_in;                      /* boundary strengths derived on the GPU instead of handed over */
        if (bi && (!bi->mvf || !bi->cbf_luma || !bi->call_log2 || !bi->ctb_flags))
            FAIL(e, OH_E_ARG, "bs_in: all four maps are required");
        if (bi && (p.log2_min_pu_size < 2 || p.log2_min_tb_size < 2))
            FAIL(e, OH_E_ARG, "bs_in: min PU / TB size below 4");
        if (bi) {
            const size_t n_cells = (size_t)(p.width >> p.log2_min_tb_size) * (p.height >> p.log2_min_tb_size);
            for (size_t i = 0; i < n_cells; i++)
                if (bi->call_log2[i] && (bi->call_log2[i] < p.log2_min_tb_size || bi->call_log2[i] > p.log2_ctb_size))
                    FAIL(e, OH_E_ARG, "bs_in: call_log2[%zu] = %d is not a block size of this picture", i, bi->call_log2[i]);
        }
        if ((!bi && (!f->vertical_bs || !f->horizontal_bs || f->bs_size < oh_bs_size(&p))) || !f->qp_y_tab || !f->deblock)
            FAIL(e, OH_E_ARG, "deblock side arrays missing or too small");
    }
    if ((p.pcm_loop_filter_disable || p.transquant_bypass_enable) && !f->is_pcm)
        FAIL(e, OH_E_ARG, "is_pcm map required when pcm loop-filter disable / transquant bypass is on");
    cnt->n_pu = f->n_pu; cnt->n_mc_luma = (uint32_t)nl; cnt->n_mc_chroma = (uint32_t)nc; cnt->n_tu = f->n_tu;
    cnt->n_intra = f->n_intra; cnt->n_sub = f->n_intra ? f->n_sub : 0; cnt->n_ictu = f->n_intra ? f->n_ictu : 0;
    cnt->n_levels = f->n_intra ? f->n_levels : 0;
    return OH_OK;
}

/* pinned block the summary of one work list lands in */
static void *summary_block_get(OhEngine *e, size_t bytes, bool *pooled)
{
    void *p = nullptr;
    *pooled = bytes <= OH_SUMMARY_BLOCK;
    if (*pooled && !e->sum_pool.empty()) { p = e->sum_pool.back(); e->sum_pool.pop_back(); return p; }
    return hipHostMalloc(&p, *pooled ? (size_t)OH_SUMMARY_BLOCK : bytes, hipHostMallocDefault) == hipSuccess ? p : nullptr;
}

/* the host part of one hand-over and its H2D copy; finish_uploads() enqueues the preparation kernels behind it */
static int upload_one(OhEngine *e, const OhFrame *f, OhDevFrame **out)
{
    *out = nullptr;
    int rc = check_params(e, &f->p);
    if (rc)
        return rc;
    Pic *cur = get_pic(e, f->cur_pic);
    if (!cur || !same_geometry(cur->p, f->p))
        FAIL(e, OH_E_ARG, "cur_pic %d is not an allocated picture of this geometry", f->cur_pic);
    static const bool timing = getenv("OHEVC_UPLOAD_TIMING") != nullptr;      /* diagnostic: where the host time of an upload goes */
    auto tnow = [] { return std::chrono::steady_clock::now(); };
    auto t_begin = tnow();
    OhPrepCounts cnt;
    uint32_t tu_cnt[4], n_cross = 0, ref_ok = 0;
    uint16_t ref_used = 0;
    bool any_dense = false;
    static thread_local std::vector<uint32_t> pu_off;
    HostTimer t_all(e, OH_HT_UPLOAD);
    { HostTimer t(e, OH_HT_UPLOAD_COUNT);
    rc = check_host_side(e, f, cur, &cnt, tu_cnt, &n_cross, &any_dense, &ref_used, &ref_ok, pu_off);
    }
    if (rc)
        return rc;
    auto t_valid = tnow();
    HIPCHK(e, hipSetDevice(e->device));

    const OhPicParams &p = f->p;
    const size_t n_ctb = (size_t)oh_ctb_width(&p) * oh_ctb_height(&p);
    const size_t n_pcm = (size_t)oh_min_pu_width(&p) * oh_min_pu_height(&p);
    const bool has_sao = p.sao_enabled && f->sao;
    const bool has_db = p.deblock_enabled != 0;

    /* arena: [copied: header, raw lists, side arrays, coefficient pool] [device only: prepared lists, scratch, residual pool] */
    struct Seg { const void *src; size_t bytes, off; size_t pack_n; };     /* pack_n != 0: src holds pack_n boundary strengths, one per byte */
    Seg seg[56];
    int ns = 0;
    size_t total = 0;
    auto add = [&](const void *src, size_t bytes) {
        if (ns >= 56) abort();                                  /* a segment was added without growing seg[] */
        seg[ns].src = src; seg[ns].bytes = bytes; seg[ns].off = total; seg[ns].pack_n = 0;
        total += align_up(bytes ? bytes : 1, 256);
        return ns++;
    };
    DevFrame hd;
    memset(&hd, 0, sizeof(hd));
    int s_hdr = add(&hd, sizeof(DevFrame));
    int s_pu = add(f->pu, (size_t)f->n_pu * sizeof(OhPu));
    int s_puoff = add(pu_off.data(), 2 * ((size_t)f->n_pu + 1) * sizeof(uint32_t));
    int s_wp = add(f->wp, (size_t)f->n_wp * sizeof(OhWeights));
    int s_tur = add(f->tu, (size_t)f->n_tu * sizeof(OhTu));
    int s_tusp = add(f->tu_sparse, f->tu_sparse ? (size_t)f->n_tu * sizeof(uint32_t) : 0);
    int s_tucr = add(f->tu_cross, f->tu_cross ? (size_t)f->n_tu * sizeof(uint32_t) : 0);
    int s_sparse = add(f->sparse, (size_t)(f->sparse ? f->n_sparse : 0) * sizeof(uint32_t));
    int s_scaling = add(f->scaling, f->scaling ? sizeof(OhScalingList) : 0);
    int s_inr = add(f->intra, (size_t)f->n_intra * sizeof(OhIntra));
    int s_ictur = add(cnt.n_ictu ? f->ictu : nullptr, (size_t)cnt.n_ictu * sizeof(OhIntraCtu));
    int s_lvl = add(cnt.n_levels ? f->level_start : nullptr, cnt.n_levels ? ((size_t)cnt.n_levels + 1) * sizeof(uint32_t) : 0);
    int s_sub = add(cnt.n_sub ? f->sub_start : nullptr, cnt.n_sub ? ((size_t)cnt.n_sub + 1) * sizeof(uint32_t) : 0);
    const bool cip = p.constrained_intra_pred && f->is_intra;
    int s_isin = add(cip ? f->is_intra : nullptr, cip ? n_pcm : 0);
    const OhBsInputs *bsi = has_db ? f->bs_in : nullptr;
    const size_t bs_bytes = bsi ? oh_bs_size(&p) : f->bs_size;
    /* the grids cross PCIe and live in HBM four strengths to the byte (0..2 each: 2 bits) — a megabyte less per 4K picture */
    const size_t bs_packed = (bs_bytes + 3) / 4;
    int s_vbs = add(has_db && !bsi ? f->vertical_bs : nullptr, has_db ? bs_packed : 0);    /* with bs_in: written by bs_kernel after the copy */
    int s_hbs = add(has_db && !bsi ? f->horizontal_bs : nullptr, has_db ? bs_packed : 0);
    const bool bs_packed_in = (f->flags & OH_FRAME_BS_PACKED) != 0;          /* the grids come four to the byte already */
    if (has_db && !bsi && !bs_packed_in) seg[s_vbs].pack_n = seg[s_hbs].pack_n = bs_bytes;
    const size_t n_mtb = (size_t)(p.width >> p.log2_min_tb_size) * (p.height >> p.log2_min_tb_size);
    int s_mvf = add(bsi ? bsi->mvf : nullptr, bsi ? n_pcm * sizeof(OhMvField) : 0);
    int s_cbf = add(bsi ? bsi->cbf_luma : nullptr, bsi ? n_mtb : 0);
    int s_call = add(bsi ? bsi->call_log2 : nullptr, bsi ? n_mtb : 0);
    int s_bsf = add(bsi ? bsi->ctb_flags : nullptr, bsi ? n_ctb : 0);
    int s_qp = add(has_db ? f->qp_y_tab : nullptr, has_db ? oh_qp_tab_size(&p) : 0);
    /* the PCM / bypass map is read by the deblock and SAO passes only under these two flags (hevc_filter.c:180, 337; deblock.hip, sao.hip):
     * without them its half megabyte per 4K picture stays on the host */
    const bool need_pcm = f->is_pcm && (p.pcm_loop_filter_disable || p.transquant_bypass_enable);
    int s_pcm = add(need_pcm ? f->is_pcm : nullptr, need_pcm ? n_pcm : 0);
    int s_db = add(has_db ? f->deblock : nullptr, has_db ? n_ctb * sizeof(OhDeblockCtb) : 0);
    int s_sao = add(has_sao ? f->sao : nullptr, has_sao ? n_ctb * sizeof(OhSaoCtb) : 0);
    const bool has_pend = has_sao && f->sao_pending && oh_sao_stale_config(&p);     /* tiled pictures: the driver order as bits per CTB */
    int s_pend = add(has_pend ? f->sao_pending : nullptr, has_pend ? n_ctb : 0);
    int s_coef = add(f->coeffs, (size_t)f->n_coeff * sizeof(int16_t));
    /* the dense pool is the last copied segment: when every block came as levels nothing of it crosses PCIe */
    const size_t copy_bytes = any_dense || !f->n_tu ? total : seg[s_coef].off;
    /* device only.  The first four are cleared before the preparation kernels run. */
    const size_t sum_bytes = sizeof(DevSummary) + (size_t)cnt.n_levels * sizeof(DevLevelStat);
    const size_t zero_off = total;
    int s_cursor = add(nullptr, 16 * sizeof(uint32_t));
    int s_sum = add(nullptr, sum_bytes);
    int s_seen = add(nullptr, cnt.n_intra ? n_ctb * sizeof(uint32_t) : 0);
    int s_keep = add(nullptr, f->n_tu);
    const size_t zero_bytes = total - zero_off;
    int s_mcl = add(nullptr, (size_t)cnt.n_mc_luma * sizeof(DevMcJob));
    int s_mcc = add(nullptr, (size_t)cnt.n_mc_chroma * sizeof(DevMcJob));
    int s_aux = add(nullptr, (size_t)cnt.n_ictu * sizeof(uint32_t));
    int s_tu = add(nullptr, (size_t)f->n_tu * sizeof(DevTu));
    int s_cross = add(nullptr, (size_t)n_cross * sizeof(DevCross));
    int s_intra = add(nullptr, (size_t)f->n_intra * sizeof(DevIntra));
    int s_ictu = add(nullptr, (size_t)cnt.n_ictu * sizeof(DevIntraCtu));
    int s_small = add(nullptr, (size_t)cnt.n_sub * sizeof(uint32_t));
    int s_perm = add(nullptr, (size_t)f->n_intra * sizeof(uint32_t));
    int s_rowp = add(nullptr, cnt.n_intra ? (size_t)oh_ctb_height(&p) * sizeof(uint32_t) : 0);
    int s_wait = add(nullptr, (size_t)cnt.n_ictu * 4 * sizeof(uint32_t));
    int s_done = add(nullptr, (size_t)cnt.n_ictu * sizeof(uint32_t));
    int s_clvl = add(nullptr, (size_t)cnt.n_ictu * sizeof(uint32_t));
    int s_cord = add(nullptr, (size_t)cnt.n_ictu * sizeof(uint32_t));
    const size_t res_off = total;
    total += align_up((size_t)(f->n_coeff ? f->n_coeff : 1) * sizeof(int16_t), 256);
    const bool stale_cfg = has_db && has_sao && oh_sao_stale_config(&p);       /* see DevFrame.sao_stale */
    const size_t stale_off = total;
    if (stale_cfg)
        total += align_up(oh_sao_stale_index(&p, 3, 0, 0) * sizeof(uint16_t), 256);

    auto t_lists = tnow();
    bool best_was_new = false;
    OhDevFrame *df = new OhDevFrame();
    {   /* a pooled arena that fits (within 2x), else a new one rounded up to 1 MiB */
        HostTimer t(e, OH_HT_UPLOAD_ARENA);
        /* oldest first (the pool is in release order): an arena released long ago has no pass left that reads it, so the copy
         * need not wait for the engine stream; the most recently released one would stall the copy stream behind the passes
         * of the batch that just let go of it */
        int best = -1;
        best_was_new = false;
        for (size_t i = 0; i < e->arenas.size() && best < 0; i++)
            if (e->arenas[i].bytes >= total && e->arenas[i].bytes <= 2 * total + (1u << 20))
                best = (int)i;
        if (best >= 0 && e->arenas[best].free_ev && hipEventQuery(e->arenas[best].free_ev) != hipSuccess && e->arenas.size() < 512)
            best = -1;                                      /* even the oldest fit is still in flight: a new arena beats a stall */
        if (best >= 0) {
            df->arena = e->arenas[best].p; df->arena_bytes = e->arenas[best].bytes;
            if (e->arenas[best].free_ev) {                  /* released while passes were in flight: the copy must stay behind them */
                (void)hipStreamWaitEvent(e->copy_stream, e->arenas[best].free_ev, 0);
                sync_event_put(e, e->arenas[best].free_ev);
            }
            e->arenas.erase(e->arenas.begin() + best);
        } else {
            best_was_new = true;
            df->arena_bytes = align_up(total, (size_t)1 << 20);
            if (hipMalloc(&df->arena, df->arena_bytes) != hipSuccess) {
                (void)hipStreamSynchronize(e->stream);
                for (auto &a : e->arenas) { sync_event_put(e, a.free_ev); (void)hipFree(a.p); e->arenas_alive--; e->arena_bytes_alive -= a.bytes; }      /* the pool may be what is in the way */
                e->arenas.clear();
                if (hipMalloc(&df->arena, df->arena_bytes) != hipSuccess) {
                    df->arena = nullptr;
                    delete df;
                    FAIL(e, OH_E_NOMEM, "hipMalloc(%zu) for the work list failed", total);
                }
            }
        }
    }
    if (best_was_new) { e->arenas_alive++; e->arena_bytes_alive += df->arena_bytes; }
    char *base = (char *)df->arena;
    hd.pp = p;
    fill_planes(&hd.cur, cur, false);
    fill_planes(&hd.out, cur, has_sao);
    for (int i = 0; i < OH_MAX_REFS; i++) {
        Pic *r = get_pic(e, f->ref_pics[i]);
        df->ref_id[i] = -1; df->ref_gen[i] = 0; df->ref_half[i] = 0;
        if (ref_ok >> i & 1) {
            fill_planes(&hd.refs[i], r, r->final_b);
            df->ref_id[i] = f->ref_pics[i]; df->ref_gen[i] = r->gen; df->ref_half[i] = r->final_b ? 1 : 0;
        }
    }
    df->ref_used = ref_used;
    df->cur_gen = cur->gen;
#define AT(T, s) ((T)(base + seg[s].off))
    hd.pu = AT(const OhPu *, s_pu);
    hd.mc_luma = AT(const DevMcJob *, s_mcl); hd.mc_chroma = AT(const DevMcJob *, s_mcc);
    hd.wp = AT(const OhWeights *, s_wp);
    hd.tu = AT(const DevTu *, s_tu); hd.tu_raw = AT(const OhTu *, s_tur);
    hd.tu_sparse = f->tu_sparse ? AT(const uint32_t *, s_tusp) : nullptr;
    hd.tu_cross = f->tu_cross ? AT(const uint32_t *, s_tucr) : nullptr;
    hd.sparse = f->sparse ? AT(const uint32_t *, s_sparse) : nullptr;
    hd.cross = AT(const DevCross *, s_cross);
    hd.scaling = f->scaling ? AT(const OhScalingList *, s_scaling) : nullptr;
    hd.coeffs = AT(const int16_t *, s_coef);
    hd.coeffs_present = f->coeffs != nullptr;
    hd.res = (int16_t *)(base + res_off);
    hd.sao_stale = stale_cfg ? (uint16_t *)(base + stale_off) : nullptr;
    hd.intra = AT(const DevIntra *, s_intra); hd.intra_raw = AT(const OhIntra *, s_inr);
    hd.ictu = AT(const DevIntraCtu *, s_ictu); hd.ictu_raw = AT(const OhIntraCtu *, s_ictur);
    hd.sub_start = AT(const uint32_t *, s_sub);
    hd.sub_small = AT(const uint32_t *, s_small); hd.sub_small_w = AT(uint32_t *, s_small);
    hd.lvl_start = AT(const uint32_t *, s_lvl);
    hd.is_intra = cip ? AT(const uint8_t *, s_isin) : nullptr;
    hd.vbs = AT(const uint8_t *, s_vbs); hd.hbs = AT(const uint8_t *, s_hbs);
    hd.qp = AT(const int8_t *, s_qp);
    hd.is_pcm = need_pcm ? AT(const uint8_t *, s_pcm) : nullptr;
    hd.db = AT(const OhDeblockCtb *, s_db);
    hd.sao = has_sao ? AT(const OhSaoCtb *, s_sao) : nullptr;
    hd.sao_pending = has_pend ? AT(const uint8_t *, s_pend) : nullptr;
    hd.pu_off = AT(const uint32_t *, s_puoff); hd.ctu_aux = AT(uint32_t *, s_aux); hd.tu_keep = AT(uint8_t *, s_keep); hd.tu_cursor = AT(uint32_t *, s_cursor);
    hd.intra_perm = AT(uint32_t *, s_perm); hd.ctu_seen = AT(uint32_t *, s_seen); hd.summary = AT(void *, s_sum);
    hd.row_progress = AT(uint32_t *, s_rowp);
    hd.ctu_wait = AT(uint32_t *, s_wait); hd.ctu_done = AT(uint32_t *, s_done); hd.ctu_lvl = AT(uint32_t *, s_clvl); hd.ctu_order = AT(uint32_t *, s_cord);
    hd.err_word = e->kerr; hd.cur_pic_id = f->cur_pic;
#undef AT
    hd.n_pu = f->n_pu; hd.n_mc_luma = cnt.n_mc_luma; hd.n_mc_chroma = cnt.n_mc_chroma; hd.n_tu = f->n_tu; hd.n_intra = f->n_intra;
    hd.n_ictu = cnt.n_ictu; hd.n_sub = cnt.n_sub; hd.n_levels = cnt.n_levels; hd.n_wp = f->n_wp; hd.n_sparse = f->sparse ? f->n_sparse : 0;
    hd.ref_ok = ref_ok; hd.n_coeff = f->n_coeff;
    hd.n_cross = n_cross;
    hd.zero_ptr = (uint32_t *)(base + zero_off); hd.zero_words = (uint32_t)(zero_bytes / 4);
    for (int k = 0, first = 0; k < 4; k++) { hd.tu_first[k] = (uint32_t)first; hd.tu_cnt[k] = tu_cnt[k]; first += (int)tu_cnt[k]; }
    hd.dbg = e->dbg;
    (void)s_hdr;

    /* stage everything in one host buffer -> one H2D copy; a list that lies in pinned memory (OH_FRAME_PINNED) is copied by DMA from
     * where it lies, segment by segment: only what this function made itself (the header, the PU block offsets) is staged */
    auto t_alloc = tnow();
    const bool direct = (f->flags & OH_FRAME_PINNED) != 0 && !(has_db && !bsi && !bs_packed_in);     /* byte grids still have to be packed on the way */
    const size_t own_bytes = align_up(sizeof(DevFrame), 256) + align_up(2 * ((size_t)f->n_pu + 1) * sizeof(uint32_t), 256) + 64 * sizeof(OhPullSeg);
    OhEngine::Stage *sg;
    { HostTimer t(e, OH_HT_UPLOAD_STAGE_WAIT);
    sg = stage_acquire(e, direct ? own_bytes : copy_bytes);   /* a pinned buffer whose previous copy has completed */
    }
    df->sum_host = summary_block_get(e, sum_bytes, &df->sum_pooled);
    df->sum_bytes = sum_bytes;
    hd.summary_host = df->sum_host;                        /* pinned, device-accessible: prep_finish stores the summary there */
    if (!sg || !df->sum_host) {
        free_dev_frame(e, df);
        FAIL(e, OH_E_NOMEM, "hipHostMalloc(%zu) failed", copy_bytes);
    }
    void *stage = sg->p;
    hipStream_t cs = e->copy_stream;
    hipError_t hrc = hipSuccess;
    if (direct) {
        /* the GPU pulls the list: the two segments made here (header, PU block offsets) and the table of segments stand in the staging
         * buffer, every other segment is read from the caller's pinned memory where it lies — ONE kernel launch (prep_pull), no host
         * copy of the lists and no DMA request per array (fifteen of those per picture cost the host as much as the copy they replaced) */
        OhPullSeg *tab;
        int nt = 0;
        size_t pulled = 0;
        { HostTimer t(e, OH_HT_UPLOAD_MEMCPY);
        char *sp = (char *)stage;
        memcpy(sp, seg[s_hdr].src, seg[s_hdr].bytes);
        char *po = sp + align_up(sizeof(DevFrame), 256);
        if (seg[s_puoff].bytes) memcpy(po, seg[s_puoff].src, seg[s_puoff].bytes);
        tab = (OhPullSeg *)(po + align_up(seg[s_puoff].bytes, 256));
        tab[nt++] = OhPullSeg{ sp, (char *)df->arena, seg[s_hdr].bytes };
        if (seg[s_puoff].bytes) tab[nt++] = OhPullSeg{ po, (char *)df->arena + seg[s_puoff].off, seg[s_puoff].bytes };
        for (int i = 0; i < ns; i++)
            if (i != s_hdr && i != s_puoff && seg[i].bytes && seg[i].src && seg[i].off + seg[i].bytes <= copy_bytes) {
                tab[nt++] = OhPullSeg{ seg[i].src, (char *)df->arena + seg[i].off, seg[i].bytes };
                pulled += seg[i].bytes;
            }
        }
        HostTimer t_enq(e, OH_HT_UPLOAD_ENQUEUE);
        ohk_pull(tab, nt, pulled, cs);
        hrc = hipGetLastError();
    } else {
        { HostTimer t(e, OH_HT_UPLOAD_MEMCPY);
        if (!e->copiers) {
            static const char *cenv = getenv("OHEVC_COPY_THREADS");
            e->copiers = new CopyPool();
            e->copiers->start(cenv ? std::max(0, std::min(atoi(cenv), 8)) : 2);
        }
        /* pieces of at most 128 KB, dealt round-robin to the calling thread and the helpers */
        std::vector<CopyJob> &jobs = e->copy_jobs;
        jobs.clear();
        const size_t piece = 128 * 1024;
        for (int i = 0; i < ns; i++)
            if (seg[i].bytes && seg[i].src && seg[i].off + seg[i].bytes <= copy_bytes) {
                if (seg[i].pack_n) {                                      /* four source bytes per byte: pieces of 4 x 128 KB strengths */
                    for (size_t o = 0; o < seg[i].pack_n; o += 4 * piece)
                        jobs.push_back({ (char *)stage + seg[i].off + o / 4, (const char *)seg[i].src + o, std::min(4 * piece, seg[i].pack_n - o), true });
                } else {
                    for (size_t o = 0; o < seg[i].bytes; o += piece)
                        jobs.push_back({ (char *)stage + seg[i].off + o, (const char *)seg[i].src + o, std::min(piece, seg[i].bytes - o), false });
                }
            }
        e->copiers->run(jobs);
        }
        /* asynchronous: the caller's arrays are already copied out; the pinned buffer stays busy until `done` */
        HostTimer t_enq(e, OH_HT_UPLOAD_ENQUEUE);
        hrc = hipMemcpyAsync(df->arena, stage, copy_bytes, hipMemcpyHostToDevice, cs);
    }
    HostTimer t_enq(e, OH_HT_UPLOAD_ENQUEUE);
    if (hrc == hipSuccess)
        hrc = hipEventRecord(sg->done, cs);
    sg->busy = hrc == hipSuccess;
    e->up_bytes += copy_bytes;
    if (hrc == hipSuccess && bsi) {                        /* both grids from the maps: once per work list, the maps never change */
        hrc = hipMemsetAsync(base + seg[s_vbs].off, 0, align_up(bs_packed, 4), cs);          /* bs_kernel ORs the non-zero strengths in; the padded tail is read by the deblock pass */
        if (hrc == hipSuccess) hrc = hipMemsetAsync(base + seg[s_hbs].off, 0, align_up(bs_packed, 4), cs);
        if (hrc == hipSuccess)
            ohk_bs_derive(&p, base + seg[s_mvf].off, base + seg[s_cbf].off, base + seg[s_call].off, base + seg[s_bsf].off, bsi->loop_filter_across_tiles,
                          base + seg[s_vbs].off, base + seg[s_hbs].off, cs);
    }
    df->sum_dev = base + seg[s_sum].off;
    df->cnt = cnt;
    if (hrc != hipSuccess) {
        (void)hipStreamSynchronize(cs);
        free_dev_frame(e, df);
        FAIL(e, OH_E_HIP, "work-list upload failed: %s", hipGetErrorString(hrc));
    }
    if (timing) {
        auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        auto t_end = tnow();
        fprintf(stderr, "oh_frame_upload: host checks + counting %.3f ms, layout %.3f ms, arena %.3f ms, staging copy + enqueue %.3f ms (%zu bytes)\n",
                ms(t_begin, t_valid), ms(t_valid, t_lists), ms(t_lists, t_alloc), ms(t_alloc, t_end), copy_bytes);
    }
    df->d = (DevFrame *)base;
    df->p = p;
    df->n_mc_luma = cnt.n_mc_luma; df->n_mc_chroma = cnt.n_mc_chroma; df->n_tu = f->n_tu; df->n_intra = f->n_intra;
    for (int k = 0; k < 4; k++) df->tu_cnt[k] = tu_cnt[k];
    df->n_cross = n_cross;
    df->has_sao = has_sao;
    df->n_levels = cnt.n_levels;
    df->cur_pic = f->cur_pic;                  /* which half of cur_pic is final changes when the list is EXECUTED, not here */
    df->owner = e;
    *out = df;
    return OH_OK;
}

/* preparation kernels of n freshly copied work lists (one set of launches per 32 of them: the kernels pick the list with a grid
 * dimension, like the passes), their summaries back to pinned memory, their `ready` events */
static int finish_uploads(OhEngine *e, OhDevFrame *const *dfs, int n)
{
    HostTimer t_enq(e, OH_HT_UPLOAD_ENQUEUE);
    hipStream_t cs = e->copy_stream;
    for (int c0 = 0; c0 < n; c0 += OH_MAX_BATCH) {
        const int nb = n - c0 < OH_MAX_BATCH ? n - c0 : OH_MAX_BATCH;
        OhBatch B;
        memset(&B, 0, sizeof(B));
        OhPrepCounts mx;
        memset(&mx, 0, sizeof(mx));
        uint32_t max_cross = 0, max_runs = 0;
        for (int i = 0; i < nb; i++) {
            const OhDevFrame *df = dfs[c0 + i];
            B.f[i] = df->d;
            mx.n_pu = std::max(mx.n_pu, df->cnt.n_pu); mx.n_tu = std::max(mx.n_tu, df->cnt.n_tu);
            mx.n_intra = std::max(mx.n_intra, df->cnt.n_intra); mx.n_sub = std::max(mx.n_sub, df->cnt.n_sub);
            mx.n_ictu = std::max(mx.n_ictu, df->cnt.n_ictu); mx.n_levels = std::max(mx.n_levels, df->cnt.n_levels);
            max_runs = std::max(max_runs, ((df->cnt.n_mc_luma + 63) >> 6) + ((df->cnt.n_mc_chroma + 63) >> 6));
            max_cross = std::max(max_cross, df->n_cross);
        }
        ohk_prepare(&B, nb, &mx, max_runs, max_cross, cs);
        HIPCHK(e, hipGetLastError());
    }
    /* one point in the copy stream makes all of them ready: an event per list, recorded back to back */
    for (int i = 0; i < n; i++) {
        OhDevFrame *df = dfs[i];
        if ((df->ready = sync_event_get(e)) == nullptr)
            FAIL(e, OH_E_NOMEM, "no event for the work list");
        HIPCHK(e, hipEventRecord(df->ready, cs));
    }
    return OH_OK;
}

extern "C" int oh_frames_upload(OhEngine *e, const OhFrame *const *fs, int n, OhDevFrame **out)
{
    if (!e || n < 0 || (n && (!fs || !out)))
        return OH_E_ARG;
    for (int i = 0; i < n; i++) out[i] = nullptr;
    int rc = OH_OK;
    int done = 0;
    for (; done < n && rc == OH_OK; done++)
        rc = fs[done] ? upload_one(e, fs[done], &out[done]) : OH_E_ARG;
    if (rc == OH_OK)
        rc = finish_uploads(e, out, n);
    if (rc != OH_OK) {                                     /* all or nothing */
        (void)hipStreamSynchronize(e->copy_stream);
        for (int i = 0; i < n; i++) { free_dev_frame(e, out[i]); out[i] = nullptr; }
    }
    return rc;
}

extern "C" int oh_frame_upload(OhEngine *e, const OhFrame *f, OhDevFrame **out)
{
    if (!e || !f || !out)
        return OH_E_ARG;
    return oh_frames_upload(e, &f, 1, out);
}

/* the summary the preparation kernels left (prep.hip): waits for the list's `ready` event the first time */
static int read_summary(OhEngine *e, OhDevFrame *df, int index)
{
    if (df->summary_read)
        return df->prep_err ? OH_E_ARG : OH_OK;
    { HostTimer t(e, OH_HT_EXECUTE_WAIT_PREP);
    HIPCHK(e, hipEventSynchronize(df->ready));
    }
    const DevSummary *s = (const DevSummary *)df->sum_host;
    df->summary_read = true;
    df->prep_err = s->err;
    if (s->err) {
        static const char *what[] = { "", "prediction unit", "transform block", "intra block", "intra schedule entry" };
        FAIL(e, OH_E_ARG, "work list %d: %s %u is malformed (rectangle / reference / index out of range); nothing of it was executed",
             index, what[s->err <= 4 ? s->err : 0], s->err_item);
    }
    for (int k = 0; k < 4; k++)
        if (s->tu_cnt[k] != df->tu_cnt[k])
            FAIL(e, OH_E_ARG, "work list %d: transform block counts changed between hand-over and preparation", index);
    df->intra_area64 = s->intra_area64; df->max_passes = s->max_passes;
    const DevLevelStat *ls = (const DevLevelStat *)(s + 1);
    df->levels.resize(df->n_levels);
    for (uint32_t l = 0; l < df->n_levels; l++) {
        OhDevFrame::Level &L = df->levels[l];
        L.n_ctu = ls[l].n_ctu; L.max_items = ls[l].max_items; L.max_sub = ls[l].max_sub; L.max_res = ls[l].max_res;
        L.staged = ls[l].staged != 0; L.sum_items = ls[l].sum_items; L.sum_sub = ls[l].sum_sub;
    }
    return OH_OK;
}

/* Execute n mutually independent pictures: every pass is one launch over all of them (chunks of
 * OH_MAX_BATCH).  Nothing orders the pictures of a batch against each other, so none of them may be a
 * reference of another one. */
/* the CTUs' residual spans staged in LDS, or every block fetching its own from the pool one sub-level ahead (intra.hip: slots_prepare)?
 * Staging costs ~11 KB of LDS per workgroup in a launch that holds one all-intra CTU, i.e. workgroups per CU (OHEVC_INTRA_RES_LDS=0 / 1
 * forces one way: experiments) */
static bool res_in_lds(const OhEngine *e, uint64_t workgroups)
{
    static const char *env = getenv("OHEVC_INTRA_RES_LDS");
    if (env) return atoi(env) != 0;
    /* a launch of at most a workgroup per CU is latency-bound: stage (a 4K I picture alone 6.6 ms against 7.0); any wider one runs at
     * workgroups-per-CU x latency: keep the LDS small (all-intra batches 39.8 against 37.6 Gpix/s, the mixed bench +0.6 %) */
    return workgroups <= (uint64_t)e->n_cu;
}

extern "C" int oh_frames_execute(OhEngine *e, OhDevFrame *const *dfs, int n)
{
    if (!e || n < 0 || (n && !dfs))
        return OH_E_ARG;
    if (n == 0)
        return OH_OK;                                       /* an empty batch is a no-op */
    HostTimer t_all(e, OH_HT_EXECUTE);
    HIPCHK(e, hipSetDevice(e->device));
    for (int i = 0; i < n; i++) {
        if (!dfs[i])
            return OH_E_ARG;
        if (dfs[i]->owner != e)
            FAIL(e, OH_E_ARG, "batch: picture %d was uploaded to another engine", i);
        int src = read_summary(e, dfs[i], i);               /* a malformed list stops the batch before anything is launched */
        if (src)
            return src;
        const OhPicParams &a = dfs[0]->p, &b = dfs[i]->p;
        if (memcmp(&a, &b, sizeof(a)) != 0)
            FAIL(e, OH_E_ARG, "batch: picture %d has other parameters than picture 0", i);
    }
    HIPCHK(e, hipSetDevice(e->device));
    hipStream_t st = e->stream;
    {
        /* Pictures are looked up NOW, not at upload: the work list may have been uploaded before its references were decoded or
         * received (multi-GPU exchange: oh_pic_set_final_half after the upload), and ids may have been freed and reused since.
         * A reference whose finished half differs from the one DevFrame.refs[] points at is patched in HBM, in stream order. */
        std::vector<uint8_t> role(e->pics.size(), 0);             /* 1: written by this batch, 2: read by this batch */
        struct Patch { DevPlanes *dst; DevPlanes v; };
        std::vector<Patch> patches;
        for (int i = 0; i < n; i++) {
            OhDevFrame *df = dfs[i];
            if (!df->waited) {                              /* the list's copy runs on the copy stream */
                HIPCHK(e, hipStreamWaitEvent(st, df->ready, 0));
                df->waited = true;
            }
            Pic *c = get_pic(e, df->cur_pic);
            if (!c || c->gen != df->cur_gen)
                FAIL(e, OH_E_ARG, "batch: picture %d's cur_pic %d was freed after the upload", i, df->cur_pic);
            if (role[df->cur_pic] & 1)
                FAIL(e, OH_E_ARG, "batch: two work lists reconstruct picture %d", df->cur_pic);
            role[df->cur_pic] |= 1;
        }
        for (int i = 0; i < n; i++) {
            OhDevFrame *df = dfs[i];
            for (int s = 0; s < OH_MAX_REFS; s++) {
                if (!(df->ref_used >> s & 1))
                    continue;
                Pic *r = get_pic(e, df->ref_id[s]);
                if (!r || r->gen != df->ref_gen[s])
                    FAIL(e, OH_E_ARG, "batch: picture %d references picture %d, which was freed after the upload", i, df->ref_id[s]);
                if (role[df->ref_id[s]] & 1)
                    FAIL(e, OH_E_ARG, "batch: picture %d is a reference of picture %d of the same batch (nothing orders them)", df->ref_id[s], i);
                const uint8_t half = r->final_b ? 1 : 0;
                if (half != df->ref_half[s]) {
                    Patch pt;
                    pt.dst = &df->d->refs[s];
                    fill_planes(&pt.v, r, r->final_b);
                    patches.push_back(pt);
                    df->ref_half[s] = half;
                }
            }
        }
        if (!patches.empty()) {
            OhEngine::Stage *sg = stage_acquire(e, patches.size() * sizeof(DevPlanes));
            if (!sg)
                FAIL(e, OH_E_NOMEM, "hipHostMalloc for %zu reference patches failed", patches.size());
            DevPlanes *hp = (DevPlanes *)sg->p;
            for (size_t k = 0; k < patches.size(); k++) {
                hp[k] = patches[k].v;
                HIPCHK(e, hipMemcpyAsync(patches[k].dst, &hp[k], sizeof(DevPlanes), hipMemcpyHostToDevice, st));
            }
            HIPCHK(e, hipEventRecord(sg->done, st));
            sg->busy = true;
        }
    }
    /* per-launch events are a sample, not a log: stop bracketing once 100 k launches are pending collection */
    const bool prof = e->profile > 0, prof_launch = e->profile > 1 && e->lev_pending.size() < 200000;
    static const char *wenv = getenv("OHEVC_INTRA_WAVES");            /* experiments: force the waves per CTU */
    for (int c0 = 0; c0 < n; c0 += OH_MAX_BATCH) {
        const int nb = n - c0 < OH_MAX_BATCH ? n - c0 : OH_MAX_BATCH;
        OhDevFrame *const *fr = dfs + c0;
        const OhPicParams *p = &fr[0]->p;
        EventSet es;
        if (prof) {
            if (!e->ev_pool.empty()) {
                es = e->ev_pool.back();
                e->ev_pool.pop_back();
            } else {
                for (auto &ev : es.ev)
                    HIPCHK(e, hipEventCreate(&ev));
            }
            es.n_frames = nb;
            HIPCHK(e, hipEventRecord(es.ev[0], st));
        }
#define MARK(k) do { if (prof) HIPCHK(e, hipEventRecord(es.ev[(k) + 1], st)); } while (0)
        for (int i = 0; i < nb; i++)                       /* which half holds the finished picture once this has run */
            if (Pic *c = get_pic(e, fr[i]->cur_pic))
                c->final_b = fr[i]->has_sao;
        OhBatch all;
        memset(&all, 0, sizeof(all));
        uint32_t max_luma = 0, max_chroma = 0, max_tu[4] = { 0, 0, 0, 0 };
        size_t max_levels = 0;
        for (int i = 0; i < nb; i++) {
            all.f[i] = fr[i]->d;
            max_luma = std::max(max_luma, fr[i]->n_mc_luma); max_chroma = std::max(max_chroma, fr[i]->n_mc_chroma);
            for (int k = 0; k < 4; k++) max_tu[k] = std::max(max_tu[k], fr[i]->tu_cnt[k]);
            max_levels = std::max(max_levels, fr[i]->levels.size());
        }
        ohk_inter(&all, nb, p, max_luma, max_chroma, st);
        MARK(OH_PASS_INTER);
        ohk_residual(&all, nb, p, max_tu, st);
        {
            uint32_t max_cross = 0;
            for (int i = 0; i < nb; i++) max_cross = std::max(max_cross, fr[i]->n_cross);
            ohk_cross(&all, nb, p, max_cross, st);
        }
        MARK(OH_PASS_RESIDUAL);
        const OhCtuAreas areas = oh_ctu_areas(p->log2_ctb_size, p->chroma_format_idc);
        /* How the intra pass of each picture runs (intra.hip):
         *   direct  one launch, a wave per CTU working on the picture in HBM — pictures whose intra blocks cover less than half of
         *           their samples (B / P pictures);
         *   dag     one launch, a workgroup per CTU with the CTU staged in LDS, CTUs waiting for their neighbours' flags — the
         *           others (I pictures);
         *   levels  one launch per wavefront level (rounds 1-2), or CTU rows in one launch for a few deep pictures: kept for
         *           comparison and as the form that needs nothing of the dispatcher (OHEVC_INTRA_MODE=levels).
         * The one-launch forms never deadlock as long as the hardware hands out the workgroups of a grid in id order (per XCD):
         * an entry only waits for lower ids.  HIP does not promise that order; it is what gfx950 does (guide: "blocks are dealt
         * round-robin over the 8 XCDs"), every wait is bounded, and a wait that gives up is reported (kernel_error) instead of
         * producing a silently wrong picture. */
        static const char *menv = getenv("OHEVC_INTRA_MODE");
        const int mode_env = !menv ? 0 : !strcmp(menv, "levels") ? 1 : !strcmp(menv, "dag") ? 2 : !strcmp(menv, "direct") ? 3 : 0;
        bool in_rows[OH_MAX_BATCH] = {};                     /* handled by a one-launch form: not in the level launches below */
        if (mode_env != 1) {
            OhBatch bd, bs;                                  /* direct / staged dag */
            memset(&bd, 0, sizeof(bd)); memset(&bs, 0, sizeof(bs));
            int nd = 0, nsd = 0;
            uint32_t max_ictu_d = 0, max_ictu_s = 0, max_ictu_all = 0, max_items = 1, max_sub = 1, max_res = 0;
            uint64_t sum_items = 0, sum_sub = 0, total_entries = 0;
            bool staged = true;
            const uint64_t pic_samples64 = ((uint64_t)p->width * p->height * (p->chroma_format_idc == 0 ? 2 : p->chroma_format_idc == 1 ? 3 : p->chroma_format_idc == 2 ? 4 : 6) / 2) >> 6;
            for (int i = 0; i < nb; i++) {
                if (fr[i]->levels.empty())
                    continue;
                in_rows[i] = true;
                max_ictu_all = std::max(max_ictu_all, fr[i]->cnt.n_ictu);
                const bool sparse = mode_env == 3 || (mode_env == 0 && (uint64_t)fr[i]->intra_area64 * 2 < pic_samples64);
                if (sparse) {
                    bd.f[nd++] = fr[i]->d;
                    max_ictu_d = std::max(max_ictu_d, fr[i]->cnt.n_ictu);
                    continue;
                }
                bs.f[nsd++] = fr[i]->d;
                max_ictu_s = std::max(max_ictu_s, fr[i]->cnt.n_ictu);
                total_entries += fr[i]->cnt.n_ictu;
                for (const OhDevFrame::Level &L : fr[i]->levels) {
                    max_items = std::max(max_items, L.max_items); max_sub = std::max(max_sub, L.max_sub); max_res = std::max(max_res, L.max_res);
                    sum_items += L.sum_items; sum_sub += L.sum_sub;
                    staged = staged && (L.staged || !L.n_ctu);
                }
            }
            hipEvent_t a = nullptr, b = nullptr;
            if (prof_launch && (nd || nsd)) {                 /* the pass's launches as one bracket: they overlap nothing else on this stream */
                for (hipEvent_t *pe : { &a, &b }) {
                    if (!e->lev_pool.empty()) { *pe = e->lev_pool.back(); e->lev_pool.pop_back(); }
                    else HIPCHK(e, hipEventCreate(pe));
                }
            }
            uint32_t *tk = e->tickets + (size_t)OhEngine::TICKET_WORDS * (e->ticket_seq++ % OhEngine::TICKET_RING);
            if (nd || nsd)
                ohk_intra_dag_reset(&all, nb, max_ictu_all, tk, st);
            if (a) HIPCHK(e, hipEventRecord(a, st));
            if (nd)
                ohk_intra_direct(&bd, nd, p, max_ictu_d, tk, e->spin_limit, st);
            if (nsd) {
                OhIntraLaunch IL;
                /* residual spans in LDS only while the chip holds the whole launch (a picture alone); else the blocks fetch theirs a sub-level ahead */
                IL.staged = staged && res_in_lds(e, total_entries / 8);
                IL.level = 0;
                const double par = sum_sub ? (double)sum_items / (double)sum_sub : 1.0;
                IL.waves = wenv ? (uint32_t)atoi(wenv) : par > (nb < 8 ? 2.5 : 4.5) ? 8 : par > 1.25 ? 4 : 2;
                if (IL.waves != 2 && IL.waves != 4 && IL.waves != 8) IL.waves = 8;
                static const char *penv = getenv("OHEVC_INTRA_PHASES");
                IL.phases = penv ? (uint32_t)atoi(penv) : 2u;
                if (IL.phases < 2 || IL.phases > IL.waves || IL.waves % IL.phases) IL.phases = 2;
                size_t off = align_up((size_t)areas.total * sizeof(uint16_t), 16);
                IL.off_items = (uint32_t)off; off += (size_t)max_items * sizeof(DevIntra);
                IL.off_sub = (uint32_t)off;   off += ((size_t)max_sub + 1) * sizeof(uint32_t);
                IL.off_small = (uint32_t)off; off = align_up(off + (size_t)max_sub * sizeof(uint32_t), 16);
                IL.off_res = (uint32_t)off;   off = align_up(off + (size_t)(IL.staged ? max_res : 0) * sizeof(int16_t), 16);
                IL.off_wave = (uint32_t)off;  off += (size_t)IL.waves * OH_INTRA_WAVE_LDS;
                IL.lds_bytes = (uint32_t)off;
                ohk_intra_dag(&bs, nsd, p, &IL, max_ictu_s, tk + OH_MAX_BATCH * 32, e->spin_limit, st);
            }
            if (b) {
                HIPCHK(e, hipEventRecord(b, st));
                e->lev_pending.push_back(a);
                e->lev_pending.push_back(b);
            }
        }
        /* pictures whose levels are (nearly) the full CTU wavefront — I pictures — may run as CTU rows in ONE launch (intra.hip:
         * intra_rows_kernel): their cost is then the critical path at the average CTU length, not the sum of the levels' slowest
         * CTUs plus a launch per level (4K I picture alone: 6.6 ms against 9.1 ms).  A row's workgroup holds its slot while it
         * waits for the row above, so this only pays while the rows of the batch's I pictures leave the chip room (one
         * workgroup per CU at most); larger batches fill the level launches anyway and keep them, as do the pictures with a
         * handful of levels (measured: 32 I pictures per batch as rows cost the 4-stream bench 22 %). */
        static const char *renv = getenv("OHEVC_INTRA_ROWS");
        const size_t row_threshold = renv && !atoi(renv) ? (size_t)-1 : (size_t)oh_ctb_width(p);
        if (mode_env == 1) {
            OhBatch rows;
            memset(&rows, 0, sizeof(rows));
            int nr = 0;
            uint32_t max_items = 1, max_sub = 1, max_res = 0;
            bool staged = true;
            int deep = 0;
            for (int i = 0; i < nb; i++)
                deep += fr[i]->levels.size() >= row_threshold && !fr[i]->levels.empty();
            if (deep * oh_ctb_height(p) > e->n_cu)
                deep = 0;
            for (int i = 0; i < nb && deep; i++) {
                if (fr[i]->levels.size() < row_threshold || fr[i]->levels.empty())
                    continue;
                in_rows[i] = true;
                rows.f[nr++] = fr[i]->d;
                for (const OhDevFrame::Level &L : fr[i]->levels) {
                    max_items = std::max(max_items, L.max_items); max_sub = std::max(max_sub, L.max_sub); max_res = std::max(max_res, L.max_res);
                    staged = staged && (L.staged || !L.n_ctu);
                }
            }
            if (nr) {
                OhIntraLaunch IL;
                IL.staged = staged && res_in_lds(e, (uint64_t)nr * oh_ctb_height(p)); IL.level = 0; IL.waves = 8; IL.phases = 2;
                size_t off = align_up((size_t)areas.total * sizeof(uint16_t), 16);
                IL.off_items = (uint32_t)off; off += (size_t)max_items * sizeof(DevIntra);
                IL.off_sub = (uint32_t)off;   off += ((size_t)max_sub + 1) * sizeof(uint32_t);
                IL.off_small = (uint32_t)off; off = align_up(off + (size_t)max_sub * sizeof(uint32_t), 16);
                IL.off_res = (uint32_t)off;   off = align_up(off + (size_t)(IL.staged ? max_res : 0) * sizeof(int16_t), 16);
                IL.off_wave = (uint32_t)off;  off += (size_t)IL.waves * OH_INTRA_WAVE_LDS;
                IL.lds_bytes = (uint32_t)off;
                hipEvent_t a = nullptr, b = nullptr;
                if (prof_launch) {                            /* counted like a level launch: one launch of the pass */
                    for (hipEvent_t *pe : { &a, &b }) {
                        if (!e->lev_pool.empty()) { *pe = e->lev_pool.back(); e->lev_pool.pop_back(); }
                        else HIPCHK(e, hipEventCreate(pe));
                    }
                    HIPCHK(e, hipEventRecord(a, st));
                }
                ohk_intra_rows(&rows, nr, p, &IL, e->spin_limit, st);
                if (prof_launch) {
                    HIPCHK(e, hipEventRecord(b, st));
                    e->lev_pending.push_back(a);
                    e->lev_pending.push_back(b);
                }
            }
        }
        for (size_t l = 0; l < max_levels; l++) {
            /* the pictures that have this level, and the LDS carve-up that fits all of them */
            OhBatch sub;
            memset(&sub, 0, sizeof(sub));
            int ns = 0;
            uint32_t max_ctu = 0, max_items = 1, max_sub = 1, max_res = 0;
            bool staged = true;
            uint64_t sum_items = 0, sum_sub = 0;
            for (int i = 0; i < nb; i++) {
                if (l >= fr[i]->levels.size() || in_rows[i])
                    continue;
                const OhDevFrame::Level &L = fr[i]->levels[l];
                sub.f[ns++] = fr[i]->d;
                max_ctu = std::max(max_ctu, L.n_ctu); max_items = std::max(max_items, L.max_items);
                max_sub = std::max(max_sub, L.max_sub); max_res = std::max(max_res, L.max_res);
                sum_items += L.sum_items; sum_sub += L.sum_sub;
                staged = staged && L.staged;
            }
            if (!ns)
                continue;                                  /* every picture that has this level runs as rows */
            OhIntraLaunch IL;
            IL.staged = staged && res_in_lds(e, (uint64_t)max_ctu * ns);
            IL.level = (uint32_t)l;
            /* waves per CTU: as many as blocks run side by side in a sub-level (more only hold LDS and wave slots) */
            const double par = sum_sub ? (double)sum_items / (double)sum_sub : 1.0;
            /* a small batch cannot fill the chip anyway: spend the waves on the single picture's latency (8 as soon as
             * sub-levels hold more than ~2 blocks); a large batch is issue-bound and runs best with 4 */
            IL.waves = wenv ? (uint32_t)atoi(wenv) : par > (nb < 8 ? 2.5 : 4.5) ? 8 : par > 1.25 ? 4 : 2;
            /* a launch of many more workgroups than the chip holds is bound by workgroups per CU x their latency, not by the latency of
             * one: two waves per workgroup (twice the workgroups per CU; measured 87.7 against 85.9 Gpix/s, eight waves 76.5) */
            if (!wenv && (uint64_t)max_ctu * ns > 5ull * (uint64_t)e->n_cu && par <= 4.5) IL.waves = 2;
            if (IL.waves != 2 && IL.waves != 4 && IL.waves != 8) IL.waves = 8;
            /* sub-levels go round-robin to `phases` groups of waves (intra.hip): a group prepares its next sub-level while
             * the others finish theirs */
            static const char *penv = getenv("OHEVC_INTRA_PHASES");
            IL.phases = penv ? (uint32_t)atoi(penv) : 2u;
            if (IL.phases < 2 || IL.phases > IL.waves || IL.waves % IL.phases) IL.phases = 2;   /* a group prepares while another finishes: at least two */
            size_t off = align_up((size_t)areas.total * sizeof(uint16_t), 16);
            IL.off_items = (uint32_t)off; off += (size_t)max_items * sizeof(DevIntra);
            IL.off_sub = (uint32_t)off;   off += ((size_t)max_sub + 1) * sizeof(uint32_t);
            IL.off_small = (uint32_t)off; off = align_up(off + (size_t)max_sub * sizeof(uint32_t), 16);
            IL.off_res = (uint32_t)off;   off = align_up(off + (size_t)(IL.staged ? max_res : 0) * sizeof(int16_t), 16);
            IL.off_wave = (uint32_t)off;  off += (size_t)IL.waves * OH_INTRA_WAVE_LDS;
            IL.lds_bytes = (uint32_t)off;
            hipEvent_t a = nullptr, b = nullptr;
            if (prof_launch) {                            /* bracket every launch: the pass is many dependent launches */
                for (hipEvent_t *pe : { &a, &b }) {
                    if (!e->lev_pool.empty()) { *pe = e->lev_pool.back(); e->lev_pool.pop_back(); }
                    else HIPCHK(e, hipEventCreate(pe));
                }
                HIPCHK(e, hipEventRecord(a, st));
            }
            ohk_intra_level(&sub, ns, p, &IL, max_ctu, st);
            if (prof_launch) {
                HIPCHK(e, hipEventRecord(b, st));
                e->lev_pending.push_back(a);
                e->lev_pending.push_back(b);
            }
        }
        MARK(OH_PASS_INTRA);
        if (p->deblock_enabled)
            ohk_deblock(&all, nb, p, 0, st);
        MARK(OH_PASS_DEBLOCK_V);
        if (p->deblock_enabled)
            ohk_deblock(&all, nb, p, 1, st);
        MARK(OH_PASS_DEBLOCK_H);
        {
            OhBatch sub;
            memset(&sub, 0, sizeof(sub));
            int ns = 0;
            for (int i = 0; i < nb; i++)
                if (fr[i]->has_sao)
                    sub.f[ns++] = fr[i]->d;
            if (ns)
                ohk_sao(&sub, ns, p, st);
        }
        MARK(OH_PASS_SAO);
#undef MARK
        HIPCHK(e, hipGetLastError());
        if (prof)
            e->ev_pending.push_back(es);
        {   /* the batch's pictures are finished behind this point of the stream */
            const uint64_t seq = ++e->batch_seq;
            hipEvent_t &bev = e->batch_ev[seq % OhEngine::BATCH_RING];
            if (!bev) HIPCHK(e, hipEventCreateWithFlags(&bev, hipEventDisableTiming));
            HIPCHK(e, hipEventRecord(bev, st));
            for (int i = 0; i < nb; i++)
                if (Pic *c = get_pic(e, fr[i]->cur_pic))
                    c->done_seq = seq;
        }
    }
    return OH_OK;
}

extern "C" int oh_frame_execute(OhEngine *e, OhDevFrame *df)
{
    return oh_frames_execute(e, &df, 1);
}

extern "C" int oh_frame_free(OhEngine *e, OhDevFrame *df)
{
    if (!e || !df)
        return OH_E_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (df->ready)
        HIPCHK(e, hipEventSynchronize(df->ready));          /* never executed: its copy may still be running */
    free_dev_frame(e, df);
    return OH_OK;
}

/* stream-ordered release: the arena returns to the pool at once; whoever reuses it fills it with a copy enqueued on the engine
 * stream, i.e. behind every pass that still reads it.  No host wait (oh_frame_free waits for the stream first). */
extern "C" int oh_frame_release(OhEngine *e, OhDevFrame *df)
{
    if (!e || !df)
        return OH_E_ARG;
    if (df->owner != e)
        FAIL(e, OH_E_ARG, "oh_frame_release: work list of another engine");
    HostTimer t_all(e, OH_HT_RELEASE);
    HIPCHK(e, hipSetDevice(e->device));
    if (!df->waited && df->ready)
        HIPCHK(e, hipStreamWaitEvent(e->stream, df->ready, 0));      /* never executed: the release still has to stay behind its copy */
    free_dev_frame(e, df, true);
    return OH_OK;
}

extern "C" int oh_frame_submit(OhEngine *e, const OhFrame *f)
{
    OhDevFrame *df = nullptr;
    int rc = oh_frame_upload(e, f, &df);
    if (rc)
        return rc;
    rc = oh_frame_execute(e, df);
    /* stream-ordered release: the arena returns to the pool behind the passes that read it.  (Rounds 1-2 parked the list until
     * oh_engine_sync: a decoder that submits a long stream and syncs once at the end kept one arena per picture in HBM.) */
    const int rr = oh_frame_release(e, df);
    return rc ? rc : rr;
}

/* the two boundary-strength grids of an uploaded work list as the deblock pass reads them (handed over or derived from bs_in) */
extern "C" int oh_frame_download_bs(OhEngine *e, OhDevFrame *df, uint8_t *vbs, uint8_t *hbs, size_t bytes)
{
    if (!e || !df || !vbs || !hbs)
        return OH_E_ARG;
    if (df->owner != e || !df->p.deblock_enabled)
        FAIL(e, OH_E_ARG, "oh_frame_download_bs: work list of another engine / without deblocking");
    HIPCHK(e, hipSetDevice(e->device));
    if (df->ready)
        HIPCHK(e, hipEventSynchronize(df->ready));          /* the list arrives on the copy stream */
    DevFrame hd;
    HIPCHK(e, hipMemcpyAsync(&hd, df->d, sizeof(hd), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    const size_t n = bytes < oh_bs_size(&df->p) ? bytes : oh_bs_size(&df->p), np = (n + 3) / 4;
    std::vector<uint8_t> pk(2 * np);
    HIPCHK(e, hipMemcpyAsync(pk.data(), hd.vbs, np, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(pk.data() + np, hd.hbs, np, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (size_t i = 0; i < n; i++) {                          /* back to one strength per byte, the reference's layout */
        vbs[i] = (pk[i >> 2] >> ((i & 3) * 2)) & 3;
        hbs[i] = (pk[np + (i >> 2)] >> ((i & 3) * 2)) & 3;
    }
    return OH_OK;
}

/* ---------------- profiling ---------------- */
extern "C" int oh_engine_profile(OhEngine *e, int enable)
{
    if (!e)
        return OH_E_ARG;
    e->profile = enable < 0 ? 0 : (enable > 2 ? 2 : enable);
    return OH_OK;
}

extern "C" int oh_engine_pass_times(OhEngine *e, double *ms, uint64_t *executes, int reset)
{
    if (!e)
        return OH_E_ARG;
    HIPCHK(e, hipSetDevice(e->device));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    for (EventSet &s : e->ev_pending) {
        for (int k = 0; k < OH_N_PASSES; k++) {
            float t = 0;
            HIPCHK(e, hipEventElapsedTime(&t, s.ev[k], s.ev[k + 1]));
            e->pass_ms[k] += t;
        }
        e->executes += (uint64_t)s.n_frames;
        e->ev_pool.push_back(s);
    }
    e->ev_pending.clear();
    for (size_t i = 0; i + 1 < e->lev_pending.size(); i += 2) {
        float t = 0;
        HIPCHK(e, hipEventElapsedTime(&t, e->lev_pending[i], e->lev_pending[i + 1]));
        e->intra_launch_ms += t;
        e->intra_launches++;
        e->lev_pool.push_back(e->lev_pending[i]);
        e->lev_pool.push_back(e->lev_pending[i + 1]);
    }
    e->lev_pending.clear();
    if (ms)
        for (int k = 0; k < OH_N_PASSES; k++) ms[k] = e->pass_ms[k];
    if (executes)
        *executes = e->executes;
    if (reset) {
        for (double &v : e->pass_ms) v = 0;
        e->executes = 0;
    }
    return OH_OK;
}

/* per-LAUNCH device time of the intra pass (sum over launches, count); call after oh_engine_pass_times */
extern "C" int oh_engine_intra_launch_times(OhEngine *e, double *ms, uint64_t *launches, int reset)
{
    if (!e)
        return OH_E_ARG;
    if (ms) *ms = e->intra_launch_ms;
    if (launches) *launches = e->intra_launches;
    if (reset) { e->intra_launch_ms = 0; e->intra_launches = 0; }
    return OH_OK;
}
