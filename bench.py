#!/usr/bin/env python3
"""bench.py — decoded Mpixels/s of the MI355X block-reconstruction engine on a synthetic stream.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

The unit of work is a closed GOP (openhevc_amd/parallel.py): 1 I picture, 3 reference B pictures and 12
non-reference B pictures of the workload's geometry.  The headline (--mode decode) hands every picture's
work list over INSIDE the timed region (host work lists -> oh_frames_upload: validation, one copy, H2D,
list preparation on the GPU -> the passes -> stream-ordered release): its inputs are NOT resident in HBM;
the same passes over resident lists follow as `kernel_only` in the same JSON line.  GOPs are independent, so
--chains of them are kept in flight per GPU: the GPU form of the reference's frame threads
(pthread_frame.c).  One STEP advances every chain by one GOP (chains x 16 pictures per GPU).  The chains
are split over --streams HIP streams; the chains of one stream advance in LOCKSTEP, i.e. picture k of
all of them is one batch of independent pictures and every pass is one launch over the batch
(oh_frames_execute), which fills the GPU where a single picture's dependency chain cannot.  With N GPUs every rank decodes its own 16 pictures per step and the four
reference pictures of every rank are replicated with one RCCL all-gather per wave (weak scaling).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # BASELINE.json `metric` is quoted on 4K Main10 (configs[3] geometry); it fits one GPU
    "2160p_main10": dict(width=3840, height=2160, bit_depth=10, chroma_format_idc=1),
    # configs[1] / configs[2] geometries (parity-test and profiling cases)
    "1080p_main8": dict(width=1920, height=1080, bit_depth=8, chroma_format_idc=1),
    "2160p_main8": dict(width=3840, height=2160, bit_depth=8, chroma_format_idc=1),
    "480p_main8": dict(width=832, height=480, bit_depth=8, chroma_format_idc=1),
    # north-star target geometry (8K60 = 1990 Mpixels/s) and configs[4] (range extension 4:4:4); use --chains 16: a picture is 100 / 200 MB
    "4320p_main10": dict(width=7680, height=4320, bit_depth=10, chroma_format_idc=1),
    "4320p_444_main10": dict(width=7680, height=4320, bit_depth=10, chroma_format_idc=3),
}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
# kernel that implements each pass (name as rocprofv3 reports it, without template arguments)
PASS_KERNEL = dict(inter="mc_kernel", residual="residual_kernel", intra="intra_direct_kernel+intra_dag_kernel", deblock_v="deblock_luma_kernel<0>+deblock_chroma_kernel<0>",
                   deblock_h="deblock_luma_kernel<1>+deblock_chroma_kernel<1>", sao="sao_kernel")


def kernels_sha():
    """identifies the kernels a profile was taken with: sha256 over the engine's device sources"""
    import glob
    import hashlib
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "openhevc_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "openhevc_amd", "csrc", "*.h"))):
        with open(path, "rb") as fh:
            h.update(os.path.basename(path).encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def committed_profile(workload, kind, ext, want_round=None):
    """(path, note): the newest profiles/rNN_<kind>_<workload>.<ext> (or round `want_round`) — cited only when its
    rNN_profile_meta_<workload>.json says it was taken with the kernels of this tree; otherwise (None, why)."""
    import glob
    import re
    cands = []
    for path in glob.glob(os.path.join(ROOT, "profiles", f"r*_{kind}_{workload}.{ext}")):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        if m and (want_round is None or int(m.group(1)) == want_round):
            cands.append((int(m.group(1)), path))
    if not cands:
        return None, "no committed profile of this workload"
    rnd, path = max(cands)
    meta = os.path.join(ROOT, "profiles", f"r{rnd:02d}_profile_meta_{workload}.json")
    if not os.path.exists(meta):
        return None, f"{os.path.relpath(path, ROOT)} has no profile_meta beside it (taken before the kernels were identified): not cited"
    with open(meta) as fh:
        sha = json.load(fh).get("kernels_sha")
    if sha != kernels_sha():
        return None, f"{os.path.relpath(path, ROOT)} was taken with other kernels (sha {sha}, this tree {kernels_sha()}): not cited"
    return path, None


def cpu_baseline(params, plan_kwargs, budget_s=12.0):
    """One host thread on the SAME step plan: the reference's own C kernels when oracle/_ref/libohevc_ref.so is there
    (kind "reference": built in the container from the reference sources, shipped to the GPU box as a binary),
    else the CPU checker oracle/ (kind "port")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from openhevc_amd import parallel as P
    from oracle_backend import OracleBackend, RefBackend
    plan = P.make_step_plan(1, 0, **plan_kwargs)
    have_ref = os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libohevc_ref.so"))
    be = RefBackend(params, plan) if have_ref else OracleBackend(params, plan)
    t0 = time.perf_counter()
    steps = 0
    while True:
        P.run_step(plan, be, None)
        steps += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or steps >= 50:
            break
    pics = steps * P.pictures_per_step(plan)
    what = ("the reference's C kernels (gcc -O2 -fno-tree-vectorize on hevcdsp.c, hevcpred.c, hevc_filter.c, videodsp.c; oracle/ref_harness.c drives them)"
            if have_ref else "gcc -O2 oracle/oracle.c")
    return dict(value=round(pics * params.width * params.height / dt / 1e6, 2), unit="Mpixels/s", cores=1, kind="reference" if have_ref else "port",
                sample=f"{steps} GOP(s) = {pics} pictures of the same synthetic stream in {dt:.1f} s, single thread, {what}")


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def decoder_baseline(params, n_pictures=16):
    """The reference's WHOLE decoder (oracle/_ref/libopenhevc_ref.so = its libavutil + libavcodec HEVC files + libOpenHevc* wrapper, built in
    the container from the sources where they lie; _sse: the same with the tree's SSE4 intrinsics in the tables, deblocking in C) on a
    synthetic stream of the workload's geometry written by openhevc_amd/synth/stream.c: 1 thread and all host threads
    (slice threads over the stream's wavefront entry points, pthread_slice.c).  BASELINE.json north_star: "openHEVC's own CPU path (C and SSE4)
    on synthetic streams".  Returns None when the libraries did not travel."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refdec
    import streamgen
    if not (os.path.exists(refdec.LIB) or os.path.isdir(refdec.REF_TREE)):
        return None
    if params.chroma_format_idc != 1:
        return None
    avail = len(os.sched_getaffinity(0))                     # the host cores this process may use (the GPU box hands a share of the node to one GPU)
    cores = min(avail, 16)                                   # pthread_internal.h:26 MAX_AUTO_THREADS 16: the reference's slice-thread tables end there
    t0 = time.perf_counter()
    data, aus = streamgen.write_stream(params.width, params.height, 5, n_pictures=n_pictures, gop=2, bit_depth=params.bit_depth, wpp=1)
    t_write = time.perf_counter() - t0
    mpix = n_pictures * params.width * params.height / 1e6
    out = dict(cpu_model=cpu_model(), host_threads=cores, host_cores_available=avail,
               stream=f"{n_pictures} pictures {params.width}x{params.height} {params.bit_depth} bit, IDR + low-delay B (2 references), 64x64 CTBs, wavefront entry points, "
                      f"SAO + deblocking on, {len(data) / 1e6:.1f} MB (written in {t_write:.1f} s)")

    def run(L, threads, kind, reps):
        best = None
        for _ in range(reps):
            t = time.perf_counter()
            pics = refdec.decode(data, threads=threads, thread_type=kind, L=L, keep=False)
            dt = time.perf_counter() - t
            assert len(pics) == n_pictures, (len(pics), n_pictures)
            best = dt if best is None else min(best, dt)
        return dict(Mpixels_per_s=round(mpix / best, 1), fps=round(n_pictures / best, 2), threads=threads)

    devnull = os.open(os.devnull, os.O_WRONLY)               # the wrapper prints its thread count on every open
    saved = os.dup(2)
    os.dup2(devnull, 2)
    try:
        out["c_1t"] = run(refdec.lib(), 1, 1, 1)
        out["c_Nt"] = run(refdec.lib(), cores, 2, 2)
        if os.path.exists(refdec.SSE_LIB) or os.path.isdir(refdec.REF_TREE):
            out["sse_1t"] = run(refdec.sse_lib(), 1, 1, 1)
            out["sse_Nt"] = run(refdec.sse_lib(), cores, 2, 2)
    finally:
        os.dup2(saved, 2)
        os.close(saved)
        os.close(devnull)
    # every host core: the reference scales one decoder to 16 slice threads at most, so the node's cores are used the way its frame-level
    # users would — k independent decoder instances (child processes; this process holds the GPU), each on its own 16 threads
    k = avail // 16
    if k >= 2 and (os.path.exists(refdec.SSE_LIB) or os.path.exists(refdec.LIB)):
        import subprocess
        import tempfile
        with tempfile.NamedTemporaryFile(suffix=".265", delete=False) as fh:
            fh.write(data)
            spath = fh.name
        try:
            for key, sse in (("c_allcores", 0), ("sse_allcores", 1)):
                if sse and not os.path.exists(refdec.SSE_LIB):
                    continue
                t = time.perf_counter()
                kids = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "refdec.py"), spath, "16", str(sse)],
                                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) for _ in range(k)]
                ok = all(kid.wait() == 0 for kid in kids)
                dt = time.perf_counter() - t
                if ok:
                    out[key] = dict(Mpixels_per_s=round(k * mpix / dt, 1), fps=round(k * n_pictures / dt, 2), threads=16 * k, instances=k,
                                    note="wall time of k concurrent decoder processes incl. their start-up")
        finally:
            os.unlink(spath)
    return out


def end_to_end(n_pictures=60):
    """Real STREAMS through the public API, end to end: written 4K Main 10 and 8K Main 10 streams (openhevc_amd/synth/stream.c: IDR +
    low-delay B on two references, 64x64 CTBs, wavefront entry points, SAO + deblocking) decoded by openhevc_amd/ohevc_dec — the
    reference's `hevc` harness loop (main_hm/main.c:149-306: Init, Decode per access unit, flush; prints frame= N fps= F) as a client
    of openHevcWrapper.h only — once through the drop-in library (the reference's decoder with this repository's recording table slots
    in its CTU loop and the MI355X engine behind them) and once through the reference's own library (C + its SSE4 intrinsics), both on
    the same file with 16 slice threads over the wavefront rows (the reference's slice-thread limit).  Child processes; rank 0, N = 1."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import streamgen
    harness = os.path.join(ROOT, "openhevc_amd", "ohevc_dec")
    libs = dict(dropin=os.path.join(ROOT, "oracle", "_ref", "libopenhevc_hip.so"), reference_sse=os.path.join(ROOT, "oracle", "_ref", "libopenhevc_ref_sse.so"),
                reference_c=os.path.join(ROOT, "oracle", "_ref", "libopenhevc_ref.so"))
    if not os.path.exists(harness) or not os.path.exists(libs["dropin"]):
        return None
    threads = min(len(os.sched_getaffinity(0)), 16)
    out = dict(what="ohevc_dec (the reference's harness loop, public libOpenHevc* API only) on written streams: the drop-in library (engine inside) "
                    "next to the reference's own library on the same file, the same loop and the same clock",
               front_end_threads=threads, cpu_model=cpu_model(), streams={})

    def run(lib, path, ttype, extra=(), passes=3):
        """one child process; passes > 1: the file is decoded that many times in a row through the same decoder (ohevc_dec -l): the first pass
        is a cold start (the decoder's frame-buffer pools are empty: every picture's host buffers are new memory, page-faulted in on first
        touch — for the drop-in library and the reference alike), the later ones are the steady state of a long stream"""
        r = subprocess.run([harness, "-i", path, "-F", lib, "-c", "-n", "-p", str(threads), "-f", str(ttype), "-l", str(passes), *extra],
                           capture_output=True, text=True, timeout=900)
        lines = r.stdout.strip().splitlines()
        last = lines[-1] if lines else ""
        if r.returncode != 0 or not last.startswith("frame= "):
            return dict(error=(r.stdout[-300:] + r.stderr[-300:]).strip())
        f = last.split()
        n, t = int(f[1]), float(f[5])
        rec = dict(frames=n, seconds=t, fps=round(n / t, 2) if t > 0 else None, passes=passes)
        per = [ln.split() for ln in lines if ln.startswith("pass ")]          # "pass k: N pictures released in T s = F fps"
        if len(per) >= 2:
            rec["fps_first_pass_cold"] = float(per[0][-2])
            steady_n = sum(int(q[2]) for q in per[1:]); steady_t = sum(float(q[6]) for q in per[1:])
            rec["fps_steady"] = round(steady_n / steady_t, 2) if steady_t > 0 else None
        return rec

    for name, (w, h) in (("2160p_main10", (3840, 2160)), ("4320p_main10", (7680, 4320))):
        t0 = time.perf_counter()
        data, _ = streamgen.write_stream(w, h, 5, n_pictures=n_pictures, gop=2, bit_depth=10, wpp=1)
        rec = dict(pictures=n_pictures, stream_MB=round(len(data) / 1e6, 1), written_in_s=round(time.perf_counter() - t0, 1))
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as fh:
            fh.write(data)
            path = fh.name
        del data
        try:
            run(libs["dropin"], path, 1, passes=1)                                    # warm-up: library load, file cache
            # decode only: pictures stay in HBM (as `hevc -n` never looks at them); _with_output: every released picture fetched into
            # host planes (libOpenHevcGetOutput).  Front end on the reference's FRAME threads (every worker records its own picture;
            # recording needs no reference samples, so its motion-compensation waits fall away) and on its slice / wavefront threads
            rec["dropin_frame_threads"] = run(libs["dropin"], path, 1)
            rec["dropin_frame_threads_with_output"] = run(libs["dropin"], path, 1, ("-g",))
            rec["dropin_slice_threads"] = run(libs["dropin"], path, 2)
            rec["dropin_slice_threads_with_output"] = run(libs["dropin"], path, 2, ("-g",))
            for k in ("reference_sse", "reference_c"):
                if os.path.exists(libs[k]):
                    rec[k + "_slice_threads"] = run(libs[k], path, 2)
                    if name == "2160p_main10" and k == "reference_sse":
                        # the reference's own frame threads are an order of magnitude behind its slice threads on these streams (random far
                        # motion vectors: every picture waits for its whole reference): one pass, the 4K stream and the faster build only
                        rec[k + "_frame_threads"] = run(libs[k], path, 1, passes=1)
        finally:
            os.unlink(path)
        for k, v in rec.items():
            if isinstance(v, dict) and v.get("fps"):
                v["Mpixels_per_s"] = round((v.get("fps_steady") or v["fps"]) * w * h / 1e6, 1)
        out["streams"][name] = rec
    # SHVC: a two-layer stream, 1080p base layer + 2160p enhancement layer (x2 spatial scalability, 8 bit 4:2:0 — what
    # the reference's up-sampler is written for); both layers' pictures go through the engine, the inter-layer reference picture is resampled in
    # HBM (oh_pic_upsample).  fps counts the released (enhancement-layer) pictures, every one standing for a base-layer picture as well
    try:
        t0 = time.perf_counter()
        data, _ = streamgen.write_stream(1920, 1080 + 8, 6, n_pictures=n_pictures, gop=2, wpp=1, shvc_el_width=3840, shvc_el_height=2160 + 16)
        rec = dict(pictures=n_pictures, layers="1920x1088 -> 3840x2176, x2", stream_MB=round(len(data) / 1e6, 1), written_in_s=round(time.perf_counter() - t0, 1))
        with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as fh:
            fh.write(data)
            path = fh.name
        del data
        try:
            rec["dropin_frame_threads"] = run(libs["dropin"], path, 1)
            rec["dropin_frame_threads_with_output"] = run(libs["dropin"], path, 1, ("-g",))
            rec["dropin_slice_threads"] = run(libs["dropin"], path, 2)
            for k in ("reference_sse", "reference_c"):
                if os.path.exists(libs[k]):
                    rec[k + "_slice_threads"] = run(libs[k], path, 2)
        finally:
            os.unlink(path)
        for k, v in rec.items():
            if isinstance(v, dict) and v.get("fps"):
                v["Mpixels_per_s_both_layers"] = round((v.get("fps_steady") or v["fps"]) * (1920 * 1088 + 3840 * 2176) / 1e6, 1)
        out["streams"]["shvc_1080p_to_2160p"] = rec
    except Exception as ex:                                    # the figure is an extra: never the reason a bench run fails
        out["streams"]["shvc_1080p_to_2160p"] = dict(error=repr(ex)[:300])
    r8 = out["streams"].get("4320p_main10", {})
    d8, o8 = r8.get("dropin_frame_threads") or {}, r8.get("dropin_frame_threads_with_output") or {}
    out["north_star_8K60"] = dict(target_fps=60,
                                  steady_fps_decode=d8.get("fps_steady"), steady_fps_with_every_picture_fetched_to_host=o8.get("fps_steady"),
                                  cold_start_fps_decode=d8.get("fps_first_pass_cold"), cold_start_fps_with_every_picture_fetched_to_host=o8.get("fps_first_pass_cold"),
                                  met_decode=bool(d8.get("fps_steady") and d8["fps_steady"] >= 60), met_with_output=bool(o8.get("fps_steady") and o8["fps_steady"] >= 60),
                                  met_decode_cold_start=bool(d8.get("fps_first_pass_cold") and d8["fps_first_pass_cold"] >= 60),
                                  met_with_output_cold_start=bool(o8.get("fps_first_pass_cold") and o8["fps_first_pass_cold"] >= 60),
                                  note="steady = passes 2-3 of the 60-picture stream decoded three times in a row through one decoder; cold start = the first pass, in which every "
                                       "picture's host frame buffers (100 MB at 8K Main 10) are new memory, every worker's decoder context and recorder are built, and the 16-deep "
                                       "frame-thread pipeline fills (a picture takes ~190 ms on its worker: 16 workers x 60 pictures leave little steady state inside one pass). End to end on ONE stream the host front end (the reference's CABAC / syntax / "
                                       "motion derivation on its frame threads) sets the pace: the GPU passes of a picture take a fraction of its parse time (compare the headline: "
                                       "thousands of pictures per second over work lists)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="2160p_main10", choices=sorted(WORKLOADS),
                    help="default: the configuration BASELINE.json's metric is quoted on (4K Main10)")
    ap.add_argument("--chains", type=int, default=None, help="closed GOPs (steps) in flight per GPU (default: 128 = batches of 32 pictures on each of the 4 streams; 64 / 32 for the 8K 4:2:0 / 4:4:4 workloads, whose pictures are 100 / 200 MB)")
    ap.add_argument("--streams", type=int, default=4, help="HIP streams per GPU; the chains of one stream run in lockstep batches")
    ap.add_argument("--waves", type=int, default=4, help="reference pictures per rank and step (first is an I picture)")
    ap.add_argument("--tail", type=int, default=12, help="non-reference B pictures per rank and step")
    ap.add_argument("--mode", default="decode", choices=["decode", "kernel_only"],
                    help="decode (the headline): every picture's work list is handed over inside the timed region — oh_frame_upload (validation, "
                         "list preparation, H2D copy, boundary-strength pass when used), the passes, stream-ordered release; a second, shorter timed "
                         "region then replays resident work lists and is reported beside it as `kernel_only`.  kernel_only: only that second figure")
    ap.add_argument("--host-gops", type=int, default=4, help="distinct GOPs kept as host work lists (the chains in flight decode them round-robin, "
                    "each into its own pictures): bounds host memory, every hand-over still copies its own bytes")
    ap.add_argument("--sparse-pct", type=int, default=100, help="%% of transform blocks handed over as quantised levels and de-quantised on the GPU "
                    "(SURVEY 8f rank 1: what residual_coding parses) instead of dense de-quantised coefficients (2 B per coded sample over PCIe)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --chains GOP chains per GPU (work grows with N); strong: --chains chains over ALL GPUs — a wave of a chain is N "
                         "pictures, picture p decoded on GPU p mod N, so the pictures of the step are fixed and split N ways")
    ap.add_argument("--gop", default="ra", choices=["ra", "ldp", "intra"],
                    help="GOP shape of the chains (openhevc_amd/parallel.py make_step_plan): random access (I + reference B + non-reference B), "
                         "low delay P (every picture on the neighbour GPU's previous picture), all intra")
    ap.add_argument("--exchange", default="readers", choices=["readers", "allgather"],
                    help="N>1: send finished reference pictures to the ranks that reference them (point-to-point batches), or replicate them everywhere (one all-gather per chain and wave)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: REHEARSAL of the N>1 control flow with several ranks on one GPU (transfers staged through host memory)")
    ap.add_argument("--bs-from-motion", action="store_true", help="work lists carry the motion field instead of finished boundary-strength grids; "
                    "the engine derives the grids at upload (bs_kernel: inside the timed region in decode mode)")
    ap.add_argument("--pinned-lists", action="store_true", help="experiment: keep the host work lists in page-locked blocks from oh_host_alloc (OH_FRAME_PINNED, "
                    "boundary strengths packed): the GPU pulls them over PCIe from where they lie (prep_pull, one launch per picture) instead of the host "
                    "staging them.  Measured SLOWER: the host's share falls to 0.10 ms per picture, but beside the passes of the other batches the pull moves "
                    "~30 GB/s where the copy engines move 41 (58-62 against 83 Gpixels/s; one DMA request per array: 47): the default is ordinary host "
                    "memory, staged into one pinned block by the calling thread and the engine's copy helpers, ONE DMA per picture")
    ap.add_argument("--host-threads", type=int, default=0, help="1: a single host thread enqueues every stream (default: one thread per stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the end-to-end stream decode through the drop-in library (ohevc_dec child processes, ~1-2 min)")
    ap.add_argument("--no-profile", action="store_true", help="no HIP events between passes in the timed region")
    ap.add_argument("--no-check", action="store_true", help="skip the picture check after the timed region")
    ap.add_argument("--profile-round", type=int, default=None, help="cite profiles/rNN_* of this round (default: the newest round that was taken with this tree's kernels)")
    args = ap.parse_args()

    # One HIP stream per chain only overlaps if the runtime maps them to distinct hardware queues
    # (ROCm default: 4).  Must be set before the HIP runtime starts.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(max(8, min(2 * args.streams + 4, 24))))   # engine streams + the collectives' own

    import threading

    import torch
    import torch.distributed as dist

    from openhevc_amd import frame as F
    from openhevc_amd import parallel as P
    from openhevc_amd.engine import Engine
    from openhevc_amd.engine import lib as _engine_lib
    engine_lib = _engine_lib()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch N>1 with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the engine has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()            # rehearsal: the ranks share the GPUs that are there
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    params = F.pic_params(**WORKLOADS[args.workload])
    if args.chains is None:
        # about 100 GB of picture buffers on one GPU: 128 4K Main 10 chains (batches of 32 on each of the 4 streams), 64 / 32 at 8K 4:2:0 / 4:4:4
        args.chains = 128 if params.width * params.height <= 3840 * 2160 else (64 if params.chroma_format_idc != 3 else 32)
    n_chains = max(1, args.chains)
    if args.scaling == "strong" and world > 1:
        n_chains = max(1, n_chains // world)               # the same pictures per step as one GPU decodes alone, split over the ranks
    # the chains' picture buffers (every rank keeps the reference pictures of all ranks, parallel.GroupStore) must fit the HBM
    # that is free now; a default that does not fit is cut (and said so) rather than left to die in the allocator
    half_bytes = F.half_layout(params)[0]
    per_chain = (args.waves * (world + 1) + max(args.tail, 1) * 2) * half_bytes       # parallel.GroupStore: other ranks' pictures keep one half
    per_chain += 2 * int(2.5 * half_bytes)                 # + the work lists in flight (this batch and the next): lists, prepared lists, residual pool
    # the second timed region (resident work lists, `kernel_only`) keeps EVERY list of every chain in HBM: one GPU only — with N > 1 the
    # picture buffers of all ranks need that room, and the figure says nothing about the exchange
    with_kernel_only = world == 1 or args.mode == "kernel_only"
    if with_kernel_only:
        per_chain += (args.waves + args.tail) * int(1.5 * half_bytes)
    free_b, _total_b = torch.cuda.mem_get_info()
    chains_asked = n_chains
    fits = max(1, int(0.85 * free_b // per_chain))
    if world > 1:                                          # the same number on every rank: every rank takes part, whether it is short of memory or not
        t = torch.tensor([fits], device=torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu"))
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        fits = int(t.item())
    if n_chains > fits:
        n_chains = fits
        if rank == 0:
            print(f"bench.py: {chains_asked} chains x {per_chain / 1e9:.2f} GB do not fit {free_b / 1e9:.0f} GB of free HBM: {n_chains} chains in flight", file=sys.stderr)
    n_streams = max(1, min(args.streams, n_chains))
    knobs = dict(sparse_pct=args.sparse_pct, bs_from_motion=int(args.bs_from_motion))
    # host work lists: --host-gops distinct GOPs (seeds), generated once — the stand-in for what the reference's CTU loop records
    n_host = max(1, min(args.host_gops, n_chains))
    host = []
    for m in range(n_host):
        plan_m = P.make_step_plan(world, rank, n_waves=args.waves, n_tail=args.tail, seed=0x48455643 + m, gop=args.gop)
        host.append((plan_m, P.host_work_lists(params, plan_m, knobs, pinned_by=engine_lib if args.pinned_lists else None)))
    host_bytes = sum(fc.bytes for _, (lists, _) in host for fc in lists.values())
    # N > 1: every stream keeps its own host thread (one thread hands over 3.2 k 4K pictures/s, four 8.5 k); their exchanges are
    # issued through a turnstile in one order on every rank (parallel.Turnstile)
    turnstile = P.Turnstile(n_streams) if world > 1 and not args.host_threads == 1 else None
    groups = [[] for _ in range(n_streams)]                # per stream: [(stream, engine, process group), (plan, backend, group)...]
    chains = []
    for k in range(n_chains):
        plan_k, lists_k = host[k % n_host]
        g = groups[k % n_streams]
        if not g:
            stream = torch.cuda.Stream()
            # the collectives of one stream are issued in the same order on every rank, so its chains share a
            # communicator; different streams must not (their collectives interleave differently per rank).
            # The chains of the stream advance in lockstep: their pictures live in ONE GroupStore, laid out so that a wave is one
            # message per peer (or one all-gather), and the exchange runs on the group's own stream (P.Comm).
            n_here = len(range(k, n_chains, n_streams))
            gstore = P.GroupStore(torch, torch.device("cuda", local_rank), params, world, n_here, args.waves, args.tail)
            g.append((stream, Engine(local_rank, stream=stream.cuda_stream), dist.new_group() if world > 1 else None, gstore,
                      P.Comm(torch, torch.device("cuda", local_rank), turnstile, k % n_streams) if world > 1 else None))
        stream, engine, group, gstore, comm = g[0]
        with torch.cuda.stream(stream):
            be_k = P.EngineBackend(torch, local_rank, params, plan_k, engine=engine, host_lists=lists_k, resident=False,
                                   group=(gstore, len(g) - 1))
        g.append((plan_k, be_k, group))
        chains.append((plan_k, be_k, stream, group))
    engines = [g[0][1] for g in groups]
    plan = chains[0][0]
    plan_kwargs = dict(n_waves=args.waves, n_tail=args.tail, seed=0x48455643, gop=args.gop)
    pics_per_step = P.pictures_per_step(plan) * n_chains      # a step advances every chain in flight by one GOP

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    exchange = P.exchange_map(world, rank, args.waves, args.tail, gop=args.gop) if world > 1 and args.exchange == "readers" else None
    comms = [g[0][4] for g in groups if g and g[0][4] is not None]
    # one host thread per engine / stream (an engine is a single-submitter object).  With N > 1 the streams' collectives must be
    # issued in the same relative order on every rank: the turnstile does that (--host-threads 1: one thread enqueues everything)
    host_threads = 1 if args.host_threads == 1 else n_streams

    def run_group(g, n_steps):
        with torch.cuda.stream(g[0][0]):
            for _ in range(n_steps):
                P.run_steps_batched(g[1:], dist if world > 1 else None, exchange, g[0][4])

    def run(n_steps):
        """one step = every chain in flight advances by one GOP: each stream runs its chains as lockstep batches"""
        if host_threads == 1:
            for _ in range(n_steps):
                for g in groups:
                    run_group(g, 1)
            return
        errs = []

        def work(g):
            try:
                torch.cuda.set_device(local_rank)
                run_group(g, n_steps)
            except BaseException as exc:        # noqa: BLE001 - re-raised on the main thread
                errs.append(exc)
                if turnstile is not None:
                    turnstile.fail(exc)
        ths = [threading.Thread(target=work, args=(g,)) for g in groups]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        if errs:
            raise errs[0]

    def timed(n_warm, n_steps):
        """(seconds of n_steps steps, host seconds until everything was enqueued); per-pass events of the region stay in the engines"""
        run(max(n_warm, 1))
        barrier()
        for eng in engines:
            eng.pass_times(reset=True)
            eng.intra_launch_times(reset=True)
            eng.host_times(reset=True)
            eng.upload_bytes(reset=True)
            eng.n_batches = 0
            eng.profile(0 if args.no_profile else prof_level[0])      # 1: events between the passes of a batch (7 per batch); 2: also around every launch of the intra pass
        for _, be_k, _, _ in chains:
            be_k.upload_s, be_k.uploads, be_k.execute_s, be_k.release_s = 0.0, 0, 0.0, 0.0
        for cm in comms:
            cm.reset()
        barrier()
        t0 = time.perf_counter()
        run(n_steps)
        dt_enqueue = time.perf_counter() - t0              # host time to hand everything over / enqueue everything
        barrier()
        dt = time.perf_counter() - t0
        for eng in engines:
            eng.profile(0)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            # the exchange of the region, per rank: time on the exchange streams (events), bytes, messages
            mine = dict(ms_per_step=round(sum(cm.ms() for cm in comms) / n_steps, 3),
                        MB_sent_per_step=round(sum(cm.bytes_sent for cm in comms) / n_steps / 1e6, 2),
                        MB_received_per_step=round(sum(cm.bytes_recv for cm in comms) / n_steps / 1e6, 2),
                        messages_per_step=sum(cm.messages for cm in comms) / n_steps, collectives_per_step=sum(cm.collectives for cm in comms) / n_steps)
            every = [None] * world
            dist.all_gather_object(every, mine)
            exchange_stats.clear()
            exchange_stats.extend(every)
        return dt, dt_enqueue

    # who takes part: what the communication library itself reports (world size, backend) and every rank's device (index, name, UUID):
    # the first run on real xGMI should say what it ran on
    def device_info():
        pr = torch.cuda.get_device_properties(local_rank)
        return dict(rank=rank, local_rank=local_rank, device=pr.name, uuid=str(getattr(pr, "uuid", "")), gcn_arch=getattr(pr, "gcnArchName", ""),
                    hbm_GB=round(pr.total_memory / 1e9, 1), compute_units=pr.multi_processor_count)
    ranks_info = None
    if world > 1:
        every = [None] * world
        dist.all_gather_object(every, device_info())
        ranks_info = dict(world_size_reported_by_backend=dist.get_world_size(), backend=dist.get_backend(), devices=every)
    else:
        ranks_info = dict(world_size_reported_by_backend=1, backend=None, devices=[device_info()])
    exchange_stats = []
    prof_level = [1]            # the timed regions carry the per-pass events only; per-launch events get a short region of their own

    def launch_durations():
        """one extra step outside the timed regions with an event pair around every launch of the intra pass: (ms, launches)"""
        if args.no_profile:
            return 0.0, 0
        prof_level[0] = 2
        timed(0, 1)
        prof_level[0] = 1
        ms = n = 0
        for eng in engines:
            eng.pass_times()                               # collects the pending events (the per-launch pairs with them)
            a, c = eng.intra_launch_times(reset=True)
            ms += a
            n += c
            eng.pass_times(reset=True)
        intra_steps[0] = 1
        return ms, n

    intra_steps = [1]

    luma_px = params.width * params.height
    b = 2 if params.bit_depth > 8 else 1

    def collect(dt, n_steps):
        """per-pass figures of the region just timed: HIP events on the engines' streams (this run, nothing read from profiles/)"""
        pass_ms, n_exec, per_stream = None, 0, []
        for eng in engines:                                # sum over the streams
            ms_k, n_k = eng.pass_times()
            per_stream.append({k: round(v / n_steps, 3) for k, v in ms_k.items()})
            pass_ms = ms_k if pass_ms is None else {k: pass_ms[k] + ms_k[k] for k in ms_k}
            n_exec += n_k
        if not n_exec:
            return None
        abytes = {k: 0.0 for k in pass_ms}                 # algorithmic bytes of this rank's step, per pass (parallel.py)
        for plan_k, be_k, _, _ in chains:
            for pic in plan_k.pictures():
                for k, v in P.algorithmic_bytes(be_k.stats[pic.name], b).items():
                    abytes[k] += v
        steps_timed = n_exec / float(pics_per_step)
        n_batches = max(sum(e.n_batches for e in engines), 1)
        # the per-launch events of the intra pass come from a short region of their own (one more step, not timed)
        intra_ms, intra_n = launch_durations() if world == 1 else (0.0, 0)     # (a region of its own runs collectives: one GPU only)
        # dominant pass = the largest share of the streams' time in THIS region.  With several streams sharing the chip a pass's
        # stream-elapsed time also counts the time its launches waited for the other streams' kernels: `avg_launch_us` is that
        # event-to-event time per launch; the kernel's OWN duration comes from the committed rocprofv3 kernel trace of this command
        # (cited only when it was taken with this tree's kernels) and is what `frac` uses when it is there.
        dom = max(pass_ms, key=lambda k: pass_ms[k])
        per_batch = dict(inter=2, residual=4, deblock_v=2, deblock_h=2, sao=1, intra=1)[dom]
        if dom == "intra" and intra_n:
            n_launch_total = intra_n / max(intra_steps[0], 1e-9) * steps_timed
            avg_launch_us = intra_ms * 1e3 / intra_n
            launch_source = "HIP events around every launch of the pass, one extra step after the timed region, all streams"
        else:
            n_launch_total = n_batches * per_batch
            avg_launch_us = pass_ms[dom] * 1e3 / n_launch_total
            launch_source = "pass time between HIP events / launches of the pass, timed region, all streams"
        bytes_per_launch = abytes[dom] * steps_timed / n_launch_total
        achieved_ev = bytes_per_launch / (avg_launch_us * 1e-6) / 1e9
        knames = [k.split("<")[0] for k in PASS_KERNEL[dom].split("+")]      # every instantiation of the pass's kernels
        traffic, traffic_source = None, None
        tpath, why_t = committed_profile(args.workload, "pmc_traffic", "json", args.profile_round)
        if tpath:                                          # rocprofv3 --pmc passes of this command, recorded under profiles/ (bench.py cannot profile itself)
            with open(tpath) as fh:                        # every instantiation of the kernel, weighted by its launches
                recs = [rec for kn, rec in json.load(fh).items() if any(kn.startswith(x) for x in knames)]
            n_l = sum(rec["launches"] for rec in recs)
            if n_l:
                traffic = round(sum(rec["hbm_bytes_per_launch_corrected"] * rec["launches"] for rec in recs) / n_l)
                traffic_source = os.path.relpath(tpath, ROOT)
        else:
            traffic_source = why_t
        trace_us, trace_source = None, None
        spath, why_s = committed_profile(args.workload, "kernel_stats", "csv", args.profile_round)
        if spath:
            import csv
            with open(spath) as fh:
                rows = [r for r in csv.DictReader(fh) if any(r["Name"].startswith(x) for x in knames)]
            calls = sum(int(r["Calls"]) for r in rows)
            if calls:
                trace_us, trace_source = round(sum(float(r["TotalDurationNs"]) for r in rows) / calls / 1e3, 3), os.path.relpath(spath, ROOT)
        else:
            trace_source = why_s
        achieved_tr = bytes_per_launch / (trace_us * 1e-6) / 1e9 if trace_us else None
        achieved = achieved_tr if achieved_tr is not None else achieved_ev
        return dict(bound="hbm", kernel=PASS_KERNEL[dom], achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 6),
                    frac_from=("the kernel's own average duration in the committed rocprofv3 --kernel-trace --stats summary of this command"
                               if achieved_tr is not None else "event-to-event time per launch of this run (no committed kernel trace of these kernels)"),
                    achieved_by_events=round(achieved_ev, 3), frac_by_events=round(achieved_ev / HBM_PEAK_GBS, 6),
                    achieved_by_trace=None if achieved_tr is None else round(achieved_tr, 3),
                    frac_by_trace=None if achieved_tr is None else round(achieved_tr / HBM_PEAK_GBS, 6),
                    traffic=traffic, traffic_source=traffic_source,
                    kernel_trace_avg_launch_us=trace_us, kernel_trace_source=trace_source,
                    dominant_pass=dom, dominant_by="largest sum of stream-elapsed pass time between HIP events, this run",
                    launches_per_step=round(n_launch_total / steps_timed, 3), avg_launch_us=round(avg_launch_us, 3), avg_launch_source=launch_source,
                    algorithmic_bytes_per_launch=round(bytes_per_launch, 1),
                    intra_pass_launches=dict(per_step=round(intra_n / max(intra_steps[0], 1e-9), 1), avg_us=round(intra_ms * 1e3 / intra_n, 3) if intra_n else None,
                                             what="one launch per picture batch and form (wave per CTU on the picture / workgroup per CTU staged in LDS), dependency flags between CTUs"),
                    pass_ms_per_step={k: round(v / steps_timed, 4) for k, v in pass_ms.items()},
                    pass_ms_per_step_per_stream=per_stream,
                    pass_algorithmic_GBps={k: round(abytes[k] / max(pass_ms[k] / steps_timed, 1e-9) / 1e6, 2) for k in pass_ms},
                    pass_share={k: round(v / max(sum(pass_ms.values()), 1e-9), 4) for k, v in pass_ms.items()})

    total_pics = world * pics_per_step * args.steps
    decode = None
    decode_exchange = None
    if args.mode == "decode":
        dt, dt_enq = timed(args.warmup, args.steps)
        decode_exchange = list(exchange_stats)
        up_s = sum(be_k.upload_s for _, be_k, _, _ in chains)
        n_up = sum(be_k.uploads for _, be_k, _, _ in chains)
        ex_s = sum(getattr(be_k, "execute_s", 0.0) for _, be_k, _, _ in chains)
        rl_s = sum(getattr(be_k, "release_s", 0.0) for _, be_k, _, _ in chains)
        ht = {}
        for eng in engines:
            for k, (ms, calls) in eng.host_times().items():
                ht[k] = ht.get(k, 0.0) + ms
        host_profile = {k: round(v / max(n_up, 1), 4) for k, v in ht.items()}       # ms per picture, summed over the host threads
        up_bytes = sum(eng.upload_bytes() for eng in engines)
        host_profile["MB_over_pcie_per_picture"] = round(up_bytes / max(n_up, 1) / 1e6, 3)
        host_profile["pcie_GBps_in_region"] = round(up_bytes / dt / 1e9, 2)
        decode = dict(dt=dt, dt_enqueue=dt_enq, roofline=collect(dt, args.steps) if rank == 0 else None, host_profile=host_profile,
                      upload_ms_per_picture=round(up_s * 1e3 / max(n_up, 1), 4),
                      host_ms_per_picture=dict(upload=round(up_s * 1e3 / max(n_up, 1), 4), execute_enqueue_incl_wait_for_preparation=round(ex_s * 1e3 / max(n_up, 1), 4),
                                               release=round(rl_s * 1e3 / max(n_up, 1), 4)))
    # resident work lists: the same passes without the hand-over
    k_steps = args.steps if args.mode == "kernel_only" else max(2, args.steps // 2)
    k_warm = args.warmup if args.mode == "kernel_only" else 1
    kernel_only = None
    if with_kernel_only:
        for g in groups:
            with torch.cuda.stream(g[0][0]):
                for _, be_k, _ in g[1:]:
                    be_k.make_resident()
        kdt, kdt_enq = timed(k_warm, k_steps)
        if args.mode == "kernel_only":
            decode_exchange = list(exchange_stats)
        kernel_only = dict(dt=kdt, dt_enqueue=kdt_enq, roofline=collect(kdt, k_steps) if rank == 0 else None)

    # the pictures of the timed regions are real pictures: one chain per (stream, host GOP) — together every distinct work list of the
    # run — against the checker (bit-exact) before anything is printed
    check = None
    if rank == 0 and not args.no_check:
        seen, picked = set(), []
        for k, ch in enumerate(chains):
            key = (k % n_streams, k % n_host)
            if key not in seen:
                seen.add(key)
                picked.append(k)
        check = dict(ok=True, pictures=0, chains=picked, mismatching=[], checker=None,
                     what=f"{len(picked)} chains = one per (stream, host GOP): every distinct work list of the run once")
        for k in picked:
            c = check_pictures(params, chains[k], world)
            check["checker"] = c.get("checker")
            check["pictures"] += c.get("pictures", 0)
            check["mismatching"] += [f"chain {k}: {m}" for m in c.get("mismatching", [])]
            if not c["ok"]:
                check["ok"] = False
                check["error"] = c.get("error")
        if not check["ok"]:
            print(f"bench.py: PICTURE MISMATCH against the checker: {check}", file=sys.stderr)
            sys.exit(4)

    out = None
    if rank == 0:
        head = decode if decode is not None else kernel_only
        steps_h = args.steps if decode is not None else k_steps
        dt_h = head["dt"]
        n_pics_h = world * pics_per_step * steps_h
        kv = world * pics_per_step * k_steps
        gen = dict(P.default_synth_knobs(), **knobs)
        out = {
            "metric": "decoded Mpixels/s (luma), synthetic stream" + ("" if check is None else ", pictures of the run checked bit-exact against the reference-pinned oracle"),
            "value": round(n_pics_h * luma_px / dt_h / 1e6, 2), "unit": "Mpixels/s", "fps": round(n_pics_h / dt_h, 2),
            "n_gpus": world, "steps": steps_h, "warmup": args.warmup if decode is not None else k_warm, "ms_per_step": round(dt_h / steps_h * 1e3, 4),
            "host_enqueue_ms_per_step": round(head["dt_enqueue"] / steps_h * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None,
            "dtype": "u8" if params.bit_depth == 8 else "u16", "data": "synthetic",
            "mode": ("decode: every picture's work list handed over inside the timed region (oh_frame_upload = validation + list preparation + H2D, "
                     "passes, stream-ordered release)" if decode is not None else "kernel_only: resident work lists replayed"),
            "host_threads": host_threads,
            "upload_ms_per_picture": decode["upload_ms_per_picture"] if decode is not None else None,
            "host_ms_per_picture": decode["host_ms_per_picture"] if decode is not None else None,
            "host_profile_ms_per_picture": decode["host_profile"] if decode is not None else None,
            "kernel_only": None if kernel_only is None else {
                "value": round(kv * luma_px / kernel_only["dt"] / 1e6, 2), "fps": round(kv / kernel_only["dt"], 2), "steps": k_steps,
                "ms_per_step": round(kernel_only["dt"] / k_steps * 1e3, 4),
                "host_enqueue_ms_per_step": round(kernel_only["dt_enqueue"] / k_steps * 1e3, 4),
                "what": "the same passes over work lists already resident in HBM (no hand-over); one GPU only",
                "roofline": kernel_only["roofline"] if decode is not None else None},
            "config": {"workload": args.workload, "width": params.width, "height": params.height, "bit_depth": params.bit_depth,
                       "chroma_format_idc": params.chroma_format_idc, "pictures_per_step_per_gpu": pics_per_step,
                       "step": f"every chain in flight advances by one closed GOP (1 I + {args.waves - 1} reference B + {args.tail} "
                               f"non-reference B pictures): {n_chains} GOPs per GPU and step",
                       "chains_in_flight_per_gpu": n_chains, "chains_asked_per_gpu": chains_asked, "streams_per_gpu": n_streams,
                       "host_work_lists": f"{n_host} distinct GOPs, {round(host_bytes / n_host / P.pictures_per_step(plan) / 1e6, 2)} MB per picture on average, held in "
                                          + ("page-locked blocks lent by the engine (oh_host_alloc), boundary strengths packed four to the byte: pulled by the GPU from where they lie (prep_pull)"
                                             if args.pinned_lists else "ordinary host memory (staged into one pinned block by the calling thread + the engine's 2 copy helpers, one DMA per picture)"),
                       "batching": f"picture k of the {n_chains // n_streams} chains of a stream is one batch: one launch per pass",
                       "gop": args.gop,
                       "exchange": ("none (1 GPU)" if world == 1 else
                                    "finished reference pictures sent point-to-point (RCCL) to the ranks that reference them: ONE message per peer, wave and "
                                    "stream (the pictures of all chains of the stream are contiguous per rank), on the stream's own exchange stream between HIP events"
                                    if exchange is not None else "one RCCL all-gather of the finished reference pictures per wave and stream, on the stream's own exchange stream"),
                       "exchange_per_rank": decode_exchange if world > 1 else None,
                       "ranks": ranks_info,
                       "shvc": "8-bit 4:2:0 only, as reference (its up-sampler, hevc_filter.c:34 / hevcdsp_template.c:1834-2438); not part of this workload: "
                               "a two-layer stream is decoded in end_to_end.streams.shvc_1080p_to_2160p, parity in tests/test_wrapper_dropin.py",
                       "generator": dict(gen, seed=hex(plan_kwargs["seed"]), ctb=64, min_cb=8, tu="4-32")},
            "roofline": head["roofline"],
            "check": check,
        }
    for _, be_k, _, _ in chains:
        be_k.close()
    for eng in engines:
        eng.close()
    if rank == 0:
        out["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, plan_kwargs)
            dec = decoder_baseline(params)
            if dec is not None:
                out["cpu_baseline"]["whole_decoder"] = dec
                for k in ("c_1t", "c_Nt", "sse_1t", "sse_Nt"):
                    if k in dec:
                        out["cpu_baseline"][k] = dec[k]["Mpixels_per_s"]
                out["cpu_baseline"]["cpu_model"] = dec["cpu_model"]
        out["end_to_end"] = None
        if world == 1 and not args.no_end_to_end and not args.no_cpu_baseline:
            try:
                out["end_to_end"] = end_to_end()
            except Exception as exc:      # noqa: BLE001 - the bench line must come out
                out["end_to_end"] = dict(error=repr(exc))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def check_pictures(params, chain, world):
    """Downloads pictures of chain 0 as the timed regions left them — the last reference picture and the last non-reference picture of
    its GOP — and decodes the same GOP with the CPU checker (oracle/, test infrastructure: never part of what was timed).  With N > 1
    the references decoded by other ranks are taken from this rank's copies of them (which the exchange delivered)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C

    from oracle_lib import host_pic_array, oracle
    plan, be, stream, _ = chain
    be.engine.sync()
    names = [p.name for p in plan.pictures()]
    host_id = {n: i for i, n in enumerate(be.store.names())}
    pics = {}
    for n in be.store.names():
        pics[host_id[n]] = be.engine.pic_download(be.ids[n], params)
    want = {k: v.copy() for k, v in pics.items()}
    bad = []
    for pic in plan.pictures():                              # decode order: waves, then the tail
        f = be.host_lists[pic.name].with_ids(host_id[pic.name], [host_id[r] for r in pic.refs])
        if oracle().oh_or_frame(C.byref(f), host_pic_array(want)) != 0:
            return dict(ok=False, error="checker failed")
        if not want[host_id[pic.name]].equal(pics[host_id[pic.name]]):
            bad.append(str(pic.name))
        if world > 1:                                        # keep going from the engine's pictures: remote references are inputs
            want[host_id[pic.name]] = pics[host_id[pic.name]].copy()
    return dict(ok=not bad, pictures=len(names), mismatching=bad, checker="oracle/liboracle.so (pinned against the reference, tests/)")


if __name__ == "__main__":
    main()
