#!/usr/bin/env python3
"""Random sweep of the stream writer's parameter space through the reference decoder (container only: needs oracle/_ref):
every stream is decoded by the unmodified reference and by the hooked reference + CPU checker; any difference is printed with
the parameters that reproduce it.  usage: sweep_streams.py [count] [seed]   |   sweep_streams.py --harness [count] [seed]  (GPU box: engine vs reference)
   |   sweep_streams.py --sparse | --sparse-engine [count] [seed]   (the sparse hand-over through the checker / the engine)
   |   sweep_streams.py --harness-frame-threads | --harness-shvc | --harness-shvc-threads | --harness-shvc-frame-threads [count] [seed]   (the drop-in library under the
       reference's frame threads; two-layer streams)"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import refdec  # noqa: E402
import streamgen  # noqa: E402
import test_streams as T  # noqa: E402


BIG = False
TINY = False


def draw(rng):
    lc = rng.choice([4, 5, 6])
    cf = rng.choice([1, 1, 2, 3])
    kw = dict(n_pictures=rng.choice([2, 3, 4]), gop=rng.choice([0, 1, 2, 2, 3]), chroma_format_idc=cf, bit_depth=rng.choice([8, 8, 10, 10, 12, 9]),
              log2_ctb_size=lc, log2_max_tb_size=min(lc, rng.choice([3, 4, 5])), n_refs=rng.choice([1, 2, 3, 4]),
              qp=rng.randint(12, 44), sao_pct=rng.choice([0, 50, 90]), split_pct=rng.choice([20, 50, 80]), intra_pct=rng.choice([5, 20, 60]),
              cbf_pct=rng.choice([20, 55, 90]), coeff_density=rng.choice([10, 60, 100]))
    for flag, prob in (("amp", .5), ("tmvp", .4), ("weighted_pred", .3), ("constrained_intra_pred", .25), ("cu_qp_delta", .3),
                       ("transform_skip", .4), ("transquant_bypass", .3), ("sign_data_hiding", .4), ("cabac_init_present", .3),
                       ("deblocking_override", .3), ("strong_intra_smoothing", .5)):
        kw[flag] = int(rng.random() < prob)
    kw["sao"] = int(rng.random() < 0.8)
    kw["scaling_list"] = rng.choice([0, 0, 1, 2])
    if rng.random() < 0.3:
        kw["pcm"] = 1
    if kw.get("pcm"):
        kw["pcm_loop_filter"] = rng.choice([0, 1])
    if kw["bit_depth"] == 12 and kw["sao"] and rng.random() < 0.6:
        kw.update(sao_offset_scale_luma=rng.randint(0, 2), sao_offset_scale_chroma=rng.randint(0, 2))
    if rng.random() < 0.4:
        kw.update(chroma_qp_offsets=1, cb_qp_offset=rng.randint(-12, 12), cr_qp_offset=rng.randint(-12, 12))
    mcb = min(lc, rng.choice([3, 3, 3, 4, 5]))                 # smallest coding block: the picture is a whole number of them
    if mcb > 3:
        kw["log2_min_cb_size"] = mcb
        if cf != 2 and rng.random() < 0.5:                      # a smallest transform block of 8x8 / 16x16 (not with 4:2:2: the writer says why)
            kw["log2_min_tb_size"] = rng.randint(3, mcb - 1)
            kw["log2_max_tb_size"] = max(kw["log2_max_tb_size"], kw["log2_min_tb_size"])
            kw["max_th_depth_intra"] = kw["max_th_depth_inter"] = min(2, lc - kw["log2_min_tb_size"])
    if kw.get("pcm") and mcb > min(lc, 5):
        kw["pcm"] = 0
    if rng.random() < 0.35:
        kw["n_slices"] = rng.choice([2, 3, 5])
        kw["lf_across_slices"] = rng.choice([0, 1])
        kw["dependent_slices"] = int(rng.random() < 0.4)
    r = rng.random()
    if r < 0.25:
        kw.update(tile_cols=rng.choice([1, 2, 3]), tile_rows=rng.choice([1, 2, 3]), lf_across_tiles=rng.choice([0, 1]))
    elif r < 0.4:
        kw["wpp"] = 1
    if rng.random() < 0.3:
        kw.update(tskip_rotation=rng.choice([0, 1]), tskip_context=rng.choice([0, 1]), implicit_rdpcm=rng.choice([0, 1]), explicit_rdpcm=rng.choice([0, 1]),
                  intra_smoothing_disabled=rng.choice([0, 1]), log2_max_tskip_size=rng.choice([0, 3, 5]))
        if not kw.get("wpp"):
            kw["persistent_rice"] = rng.choice([0, 1])
    if cf == 3 and rng.random() < 0.5:
        kw["cross_component_pred"] = 1
    if kw["gop"] == 3:
        kw["n_refs"] = max(kw["n_refs"], 2)
        kw["n_pictures"] = rng.choice([5, 6, 9])
    if rng.random() < 0.2:
        kw["idr_period"] = rng.choice([2, 3, 5])
    if cf == 1 and rng.random() < 0.15:
        kw.update(conf_win_left=2 * rng.randint(0, 4), conf_win_right=2 * rng.randint(0, 4), conf_win_top=2 * rng.randint(0, 4), conf_win_bottom=2 * rng.randint(0, 4))
    if rng.random() < 0.15:
        kw["mvd_range"] = rng.choice([8, 600, 4000])
    unit = 1 << mcb

    def fit_window(w, h):                                       # a window that leaves at least 8 x 8 samples
        if kw.get("conf_win_left", 0) + kw.get("conf_win_right", 0) > w - 8 or kw.get("conf_win_top", 0) + kw.get("conf_win_bottom", 0) > h - 8:
            for k in ("conf_win_left", "conf_win_right", "conf_win_top", "conf_win_bottom"):
                kw.pop(k, None)
        return w, h

    def dim(lo, hi):
        return max(unit, 8 * rng.randint(lo, hi) // unit * unit)
    if TINY:                                                    # --tiny: pictures of 8 .. 64 samples (one or a few CTBs, partial ones)
        kw = {k: v for k, v in kw.items() if not k.startswith("conf_win")}
        return unit * rng.randint(1, max(1, 64 // unit)), unit * rng.randint(1, max(1, 64 // unit)), rng.randint(1, 10 ** 6), kw
    if BIG:                                                     # --big: pictures up to 1920 x 1088 (many workgroups per pass, the CTU-row intra kernel)
        kw["n_pictures"] = min(kw["n_pictures"], 3) if kw["gop"] != 3 else 5
        return fit_window(dim(40, 240), dim(30, 136)) + (rng.randint(1, 10 ** 6), kw)
    return fit_window(dim(2, 40), dim(2, 30)) + (rng.randint(1, 10 ** 6), kw)


def two_layers(rng, w, h, kw):
    """turns a drawn parameter set into a two-layer (SHVC) one: what the writer's enhancement layer allows (8 bit 4:2:0, no window, not
    hierarchical B) and an enhancement-layer size at a ratio of 1 (SNR), 1.5, 2 or anything between 1 and 2.5 per axis"""
    kw = dict(kw, bit_depth=8, chroma_format_idc=1)
    for k in ("conf_win_left", "conf_win_right", "conf_win_top", "conf_win_bottom", "sao_offset_scale_luma", "sao_offset_scale_chroma",
              "tskip_rotation", "tskip_context", "implicit_rdpcm", "explicit_rdpcm", "intra_smoothing_disabled", "log2_max_tskip_size",
              "persistent_rice", "cross_component_pred"):
        kw.pop(k, None)                                             # (the range extensions: the reference reads a layer-1 PPS extension as something else)
    if kw.get("gop") == 3:
        kw["gop"] = 2
    m = 1 << kw.get("log2_min_cb_size", 3)
    # the ratios the reference has its own slot variants for: x1 (SNR), x1.5, x2.  For other ratios its CTB path and its whole-picture slot
    # can differ in the last chroma row of a CTB row (tests/test_upsample_vs_ref.py records it): there is no single reference output
    r = rng.choice([1, 1.5, 2, 2])
    if r == 1.5 and ((w * 3) % (2 * m) or (h * 3) % (2 * m)):
        r = 2
    ew, eh = int(w * r), int(h * r)                             # (the writer refuses what is too small: one CTB row / column)
    kw.update(shvc_el_width=ew, shvc_el_height=eh)
    return kw


def harness_sweep(count, seed, extra=(), min_ctb=4, shvc=False):
    """the same sweep with the ENGINE on the other side (GPU box; oracle/_ref travels there): ohevc_dec -o against the reference's output"""
    import subprocess
    import tempfile
    rng = random.Random(seed)
    harness, hooked = os.path.join(ROOT, "openhevc_amd", "ohevc_dec"), os.path.join(ROOT, "oracle", "_ref", "libopenhevc_hip.so")
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(count):
            w, h, s, kw = draw(rng)
            if kw["log2_ctb_size"] < min_ctb:
                kw["log2_ctb_size"] = min_ctb
            if shvc:
                kw = two_layers(rng, w, h, kw)
            if "-p" in extra and shvc:                             # (tiles under slice threads: the reference's own output depends on the thread count)
                kw.update(tile_cols=1, tile_rows=1)
            if "-p" in extra and extra[extra.index("-f") + 1] == "1":
                kw.update(pcm=0, transquant_bypass=0)              # (its PCM / bypass flags are per frame-thread context and never cleared: hevc.c:147, 1440)
            try:
                data, _ = streamgen.write_stream(w, h, s, **kw)
            except ValueError:
                continue                                            # (a combination the two-layer writer refuses)
            # the reference's own output depends on its threading in places (tiles: its filters then run at the end of the picture in
            # raster order, hevc.c:2967-3003; PCM / bypass flags per frame-thread context): the unmodified decoder runs with the same
            # threads as the front end
            with refdec.captured_stderr():
                # FRAME threads: against the reference's SINGLE-threaded output — under its own frame threads the reference differs from itself on
                # ~2 % of these streams (and not always the same way twice: profiles/r03_stream_sweep.txt), the drop-in library does not
                slice_threads = "-p" in extra and extra[extra.index("-f") + 1] == "2"
                pics = refdec.decode(data, threads=int(extra[extra.index("-p") + 1]), thread_type=2) if slice_threads else refdec.decode(data)
            open(os.path.join(tmp, "s.bin"), "wb").write(data)
            for old in os.listdir(tmp):
                if old.startswith("o_"):
                    os.unlink(os.path.join(tmp, old))
            r = subprocess.run([harness, "-i", os.path.join(tmp, "s.bin"), "-F", hooked, "-c", "-o", os.path.join(tmp, "o.yuv")] + list(extra),
                               capture_output=True, text=True, timeout=600)
            ow, oh = pics[0][0].shape[1], pics[0][0].shape[0]
            name = os.path.join(tmp, f"o_{ow}x{oh}.yuv")
            ok = r.returncode == 0 and os.path.exists(name) and open(name, "rb").read() == b"".join(np.ascontiguousarray(pl).tobytes() for p in pics for pl in p)
            if not ok:
                bad += 1
                print("FAIL", w, h, s, kw, r.returncode, (r.stdout[-300:] + r.stderr[-300:]).replace("\n", " | "), flush=True)
            if i % 50 == 49:
                print(f"{i + 1} streams, {bad} differ", flush=True)
    print(f"{count} streams through the harness, {bad} differ")


def sparse_sweep(count, seed, engine):
    """the SPARSE hand-over (levels de-quantised by the checker / the engine, tests/test_sparse_pin.py) over random streams"""
    import ctypes as C
    import test_sparse_pin as S
    from openhevc_amd import frame as F
    from test_streams import host_pic_array, oracle
    rng = random.Random(seed)
    eng = None
    if engine:
        from openhevc_amd.engine import Engine, remap_frame
        eng = Engine(0)
    bad = 0
    for i in range(count):
        w, h, s, kw = draw(rng)
        kw["cu_qp_delta"] = 0                                   # the writer's level log carries the slice QP
        if kw["gop"] == 3:
            kw.update(gop=2, n_pictures=3)
        kw = {k: v for k, v in kw.items() if not k.startswith("conf_win")}
        try:
            lists, want = S.sparse_work_lists(("sweep", w, h, s, kw))
            pics = {}
            for k, (a, cur, _) in enumerate(lists):
                ff = F.FrameFromArrays(a)
                f = ff.frame
                for q in [cur] + [f.ref_pics[r] for r in range(F.OH_MAX_REFS) if f.ref_pics[r] >= 0]:
                    if q not in pics:
                        pics[q] = eng.pic_alloc(f.p) if eng else F.HostPic(f.p)
                if eng:
                    eng.frame_submit(remap_frame(f, pics))
                    got = eng.pic_download(pics[cur], f.p)
                else:
                    assert oracle().oh_or_frame(C.byref(f), host_pic_array(pics)) == 0
                    got = pics[cur]
                for c in range(3):
                    assert np.array_equal(got.visible(c), want[k][c]), ("picture", k, "plane", c)
            if eng:
                for q in pics.values():
                    eng.pic_free(q)
        except AssertionError as exc:
            bad += 1
            print("FAIL", w, h, s, kw, str(exc)[:160], flush=True)
        if i % 50 == 49:
            print(f"{i + 1} streams, {bad} differ", flush=True)
    print(f"{count} streams in sparse form through the {'engine' if engine else 'checker'}, {bad} differ")


def crop(got, kw):
    """the checker's pictures are the coded ones: crop like the decoder's output"""
    if not any(k.startswith("conf_win") for k in kw):
        return got
    l, r, t, b = (kw.get("conf_win_" + k, 0) for k in ("left", "right", "top", "bottom"))
    hs, vs = (1, 1) if kw["chroma_format_idc"] == 1 else (0, 0)
    return [[pl[(t >> (vs if c else 0)):pl.shape[0] - (b >> (vs if c else 0)), (l >> (hs if c else 0)):pl.shape[1] - (r >> (hs if c else 0))]
             for c, pl in enumerate(p)] for p in got]


def threads_sweep(count, seed):
    """hooked front end on 4 slice / wavefront threads + checker against the unmodified reference on the same threads (32x32 CTBs and larger)"""
    rng = random.Random(seed)
    bad = dep = 0
    for i in range(count):
        w, h, s, kw = draw(rng)
        kw["log2_ctb_size"] = max(kw["log2_ctb_size"], 5)
        data, _ = streamgen.write_stream(w, h, s, **kw)
        one, ref = refdec.decode(data), refdec.decode(data, threads=4, thread_type=2)
        dep += not all(np.array_equal(one[k][c], ref[k][c]) for k in range(len(ref)) for c in range(3))
        got = crop(T.decode_through_hooks(data, 4, 2), kw)
        if not (len(got) == len(ref) and all(np.array_equal(ref[k][c], got[k][c]) for k in range(len(ref)) for c in range(3))):
            bad += 1
            print("FAIL", w, h, s, kw, flush=True)
    print(f"{count} streams on 4 front-end threads: {bad} differ from the reference on 4 threads; the reference's own 1-thread output differs from its 4-thread output on {dep}")


def main():
    global BIG, TINY
    if "--big" in sys.argv:
        BIG = True
        sys.argv.remove("--big")
    if "--tiny" in sys.argv:
        TINY = True
        sys.argv.remove("--tiny")
    if len(sys.argv) > 1 and sys.argv[1] == "--threads":
        return threads_sweep(int(sys.argv[2]) if len(sys.argv) > 2 else 100, int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    if len(sys.argv) > 1 and sys.argv[1] in ("--sparse", "--sparse-engine"):
        return sparse_sweep(int(sys.argv[2]) if len(sys.argv) > 2 else 100, int(sys.argv[3]) if len(sys.argv) > 3 else 1, sys.argv[1] == "--sparse-engine")
    if len(sys.argv) > 1 and sys.argv[1] == "--harness-shvc-frame-threads":
        return harness_sweep(int(sys.argv[2]) if len(sys.argv) > 2 else 100, int(sys.argv[3]) if len(sys.argv) > 3 else 1, ("-p", "3", "-f", "1"), 4, shvc=True)
    if len(sys.argv) > 1 and sys.argv[1] in ("--harness-frame-threads", "--harness-shvc", "--harness-shvc-threads"):
        # the drop-in library under the reference's FRAME threads; two-layer streams (single thread / 4 slice threads)
        mode = sys.argv[1]
        return harness_sweep(int(sys.argv[2]) if len(sys.argv) > 2 else 100, int(sys.argv[3]) if len(sys.argv) > 3 else 1,
                             ("-p", "3", "-f", "1") if mode == "--harness-frame-threads" else ("-p", "4", "-f", "2") if mode == "--harness-shvc-threads" else (),
                             5 if mode == "--harness-shvc-threads" else 4, shvc=mode != "--harness-frame-threads")
    if len(sys.argv) > 1 and sys.argv[1] in ("--harness", "--harness-bs", "--harness-threads"):
        # -bs: boundary strengths derived on the GPU (ohevc_dec -b); -threads: the front end on 4 slice / wavefront threads (32x32 CTBs and
        # larger: with 16x16 CTBs the reference's own output depends on its thread count)
        mode = sys.argv[1]
        return harness_sweep(int(sys.argv[2]) if len(sys.argv) > 2 else 100, int(sys.argv[3]) if len(sys.argv) > 3 else 1,
                             ("-b",) if mode == "--harness-bs" else ("-p", "4", "-f", "2") if mode == "--harness-threads" else (),
                             5 if mode == "--harness-threads" else 4)
    count = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = refused = 0
    for i in range(count):
        w, h, seed, kw = draw(rng)
        try:
            data, _ = streamgen.write_stream(w, h, seed, **kw)
        except ValueError:
            refused += 1
            continue
        try:
            want = refdec.decode(data)
            got = T.decode_through_hooks(data)
            assert len(want) == len(got) == kw["n_pictures"], (len(want), len(got))
            got = crop(got, kw)
            for k in range(len(want)):
                for c in range(3):
                    assert np.array_equal(want[k][c], got[k][c]), ("picture", k, "plane", c)
        except (AssertionError, RuntimeError) as exc:
            bad += 1
            print("FAIL", w, h, seed, kw, str(exc)[:120], flush=True)
    print(f"{count} streams, {refused} refused by the writer, {bad} differ")


if __name__ == "__main__":
    main()
