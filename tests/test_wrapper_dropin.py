"""The drop-in library oracle/_ref/libopenhevc_hip.so: the reference's own decoder and libOpenHevc* wrapper (compiled from its sources
where they lie) with this repository's recording table slots in its CTU loop and the MI355X engine behind them.  A caller that knows
only openHevcWrapper.h — Init, StartDecoder, SetCheckMD5, Decode per access unit, GetPictureInfoCpy + GetOutputCpy per released
picture, flush, Close: the loop of main_hm/main.c:149-306 — must get from it what the reference's library gives, plane for plane.
The checker is the unmodified reference decoder (oracle/_ref/libopenhevc_ref.so)."""
import os
import subprocess

import numpy as np
import pytest

import refdec
import streamgen

need_lib = pytest.mark.skipif(not (os.path.exists(refdec.HIP_LIB) or os.path.isdir(refdec.REF_TREE)), reason="the drop-in library is built where the reference tree is")


@need_lib
def test_exports_exactly_the_wrapper_api():
    """openHevcWrapper.h:79-98 declares 18 functions (bin/ffmpeg_w64/libLibOpenHevcWrapper.def lists the same 18): the drop-in library
    defines every one of them.  libOpenHevcInit needs no GPU; libOpenHevcStartDecoder brings the engine up with the decoder and FAILS
    (-1, the API's error value) on a box without one — there is no CPU fallback behind this library"""
    L = refdec.hip_lib()
    out = subprocess.run(["nm", "-D", "--defined-only", refdec.HIP_LIB], capture_output=True, text=True, check=True).stdout
    defined = {ln.split()[-1] for ln in out.splitlines() if " T " in ln}
    assert set(refdec.WRAPPER_API) <= defined, sorted(set(refdec.WRAPPER_API) - defined)
    assert {n for n in defined if n.startswith("libOpenHevc")} == set(refdec.WRAPPER_API)
    import ctypes as C
    h = C.c_void_p(L.libOpenHevcInit(1, 1))
    assert h
    L.libOpenHevcVersion.restype = C.c_char_p
    L.libOpenHevcVersion.argtypes = [C.c_void_p]
    assert L.libOpenHevcVersion(h).startswith(b"OpenHEVC")
    with refdec.captured_stderr():
        started = L.libOpenHevcStartDecoder(h)
    assert started == (1 if os.path.exists("/dev/kfd") else -1)       # the ROCm compute device node: a GPU box has it, the build container does not
    L.libOpenHevcClose(h)


STREAMS = [
    ("main8_lowdelay", 416, 240, 31, dict(n_pictures=8, gop=2), 1, 1),
    ("main10_tools", 416, 240, 32, dict(n_pictures=6, gop=2, bit_depth=10, amp=1, pcm=1, transform_skip=1, transquant_bypass=1, weighted_pred=1, scaling_list=1), 1, 1),
    ("hier_b_reordered", 416, 240, 33, dict(n_pictures=11, gop=3, tmvp=1, n_refs=3, idr_period=7), 1, 1),
    ("main10_window", 416, 240, 34, dict(n_pictures=5, gop=2, bit_depth=10, conf_win_left=6, conf_win_right=10, conf_win_top=4, conf_win_bottom=12), 1, 1),
    ("rext444_ccp", 416, 240, 35, dict(n_pictures=4, gop=2, chroma_format_idc=3, cross_component_pred=1, transform_skip=1), 1, 1),
    ("slices_tiles", 416, 240, 36, dict(n_pictures=5, gop=1, n_slices=3, tile_cols=2, tile_rows=2, lf_across_tiles=0), 1, 1),
    ("wavefront_slice_threads", 832, 480, 37, dict(n_pictures=6, gop=2, wpp=1), 4, 2),
    ("hd_main10_wavefront", 1920, 1080, 38, dict(n_pictures=4, gop=2, bit_depth=10, wpp=1), 8, 2),
    # the reference's FRAME threads (pthread_frame.c): several pictures between frame start and output at once, each recorded by its
    # own worker, handed to the engine in decode order; the next picture's TMVP waits for the parsed rows of its collocated picture
    ("frame_threads_lowdelay_tmvp", 832, 480, 39, dict(n_pictures=14, gop=2, tmvp=1, n_refs=2), 4, 1),
    ("frame_threads_hier_b", 416, 240, 40, dict(n_pictures=13, gop=3, tmvp=1, n_refs=3, idr_period=9), 6, 1),
    ("frame_threads_intra_and_tools", 416, 240, 42, dict(n_pictures=9, gop=1, bit_depth=10, pcm=1, transform_skip=1, weighted_pred=1, scaling_list=1), 3, 1),
]


@pytest.mark.gpu
@need_lib
@pytest.mark.parametrize("name,w,h,seed,kw,threads,ttype", STREAMS, ids=[s[0] for s in STREAMS])
def test_wrapper_loop_matches_the_reference_library(name, w, h, seed, kw, threads, ttype):
    """the harness loop through the drop-in library vs through the reference's: same number of output pictures, in the same (output)
    order, every plane of every picture equal (cropped size, bit depth and chroma format as libOpenHevcGetPictureInfoCpy reports them)"""
    data, _ = streamgen.write_stream(w, h, seed, **kw)
    # frame threads: the reference's library in the SAME thread configuration — its per-context side arrays make its output depend on it
    # where a stream uses PCM / bypass with the loop filter off (s->is_pcm is never cleared per picture, hevc.c:147,1440: every frame
    # thread's context accumulates the flags of ITS pictures only)
    want = refdec.decode(data, threads=threads, thread_type=ttype) if ttype == 1 else refdec.decode(data)
    got = refdec.decode(data, threads=threads, thread_type=ttype, L=refdec.hip_lib())
    assert len(got) == len(want) and len(want) > 0, (len(got), len(want))
    for k, (a, b) in enumerate(zip(want, got)):
        for c in range(3):
            assert a[c].shape == b[c].shape and a[c].dtype == b[c].dtype, (name, k, c, a[c].shape, b[c].shape)
            assert np.array_equal(a[c], b[c]), f"{name}: output picture {k} plane {c} differs from the reference library's"


# SHVC: two-layer streams as the reference parses them (SHM 4.1 draft syntax).  The wrapper runs one decoder per
# layer on every access unit (openHevcWrapper.c:112-133); the enhancement layer's pictures predict from the base layer's picture of the
# same access unit resampled to their size — ff_upsample_block and the upsample_filter_block_* slots in the reference
# (hevc_filter.c:1370-1426, hevcdsp_template.c:1834-2162), oh_pic_upsample on the engine's pictures here.
SHVC = [
    ("x2", 416, 240, 832, 480, 51, dict(n_pictures=5, gop=2), 1, 1),
    ("x1_5", 416, 240, 624, 360, 52, dict(n_pictures=5, gop=2, amp=1, transform_skip=1), 1, 1),
    ("snr_x1", 416, 240, 416, 240, 53, dict(n_pictures=4, gop=1), 1, 1),
    ("any_ratio", 416, 240, 560, 400, 54, dict(n_pictures=4, gop=2), 1, 1),                   # the slots' generic ("DEFAULT") filter path, different ratios per axis
    ("x2_idr_period_hier_bl", 192, 128, 384, 256, 55, dict(n_pictures=9, gop=2, idr_period=4, n_refs=2, tmvp=1), 1, 1),
    ("x2_wavefront_slice_threads", 416, 240, 832, 480, 56, dict(n_pictures=4, gop=2, wpp=1), 4, 2),
    # full size: a 4K base layer under an 8K enhancement layer (SURVEY.md 8 configs[4]'s SHVC half, 8 bit 4:2:0 as the reference's up-sampler)
    ("uhd_to_8k_x2", 3840, 2160, 7680, 4320, 59, dict(n_pictures=3, gop=2, wpp=1), 8, 2),
    ("x1_5_tiny_ctb16", 96, 64, 144, 96, 57, dict(n_pictures=4, gop=2, log2_ctb_size=4, log2_max_tb_size=4), 1, 1),
]


@pytest.mark.gpu
@need_lib
@pytest.mark.parametrize("name,w,h,ew,eh,seed,kw,threads,ttype", SHVC, ids=[s[0] for s in SHVC])
def test_two_layer_streams_match_the_reference_library(name, w, h, ew, eh, seed, kw, threads, ttype):
    """the same harness loop over a two-layer stream: the pictures the wrapper releases (it exposes the highest layer that has one,
    openHevcWrapper.c:139-152) are the reference library's, plane for plane — base-layer AND enhancement-layer pictures"""
    data, _ = streamgen.write_stream(w, h, seed, shvc_el_width=ew, shvc_el_height=eh, **kw)
    with refdec.captured_stderr():
        want = refdec.decode(data, threads=threads, thread_type=ttype) if ew > 4000 else refdec.decode(data)      # (slice threads do not change its output: test_streams.py)
    got = refdec.decode(data, threads=threads, thread_type=ttype, L=refdec.hip_lib())
    assert len(got) == len(want) and len(want) > 0, (len(got), len(want))
    assert sum(p[0].shape == (eh, ew) for p in want) >= kw["n_pictures"] - 1, [p[0].shape for p in want]
    for k, (a, b) in enumerate(zip(want, got)):
        for c in range(3):
            assert a[c].shape == b[c].shape and a[c].dtype == b[c].dtype, (name, k, c, a[c].shape, b[c].shape)
            assert np.array_equal(a[c], b[c]), f"{name}: output picture {k} plane {c} ({a[c].shape}) differs from the reference library's"


SHVC_FRAME_THREADS = [
    ("x2", 416, 240, 832, 480, 61, dict(n_pictures=9, gop=2), 4),
    ("x1_5_idr_period", 416, 240, 624, 360, 62, dict(n_pictures=10, gop=2, idr_period=4, amp=1), 3),
    ("snr", 416, 240, 416, 240, 63, dict(n_pictures=7, gop=1), 5),
    ("hd_to_uhd_wavefront_syntax", 1920, 1080 + 8, 3840, 2160 + 16, 64, dict(n_pictures=8, gop=2, wpp=1), 8),
]


@pytest.mark.gpu
@need_lib
@pytest.mark.parametrize("name,w,h,ew,eh,seed,kw,threads", SHVC_FRAME_THREADS, ids=[s[0] for s in SHVC_FRAME_THREADS])
def test_two_layer_streams_under_frame_threads(name, w, h, ew, eh, seed, kw, threads):
    """both layers' decoders on the reference's FRAME threads (pthread_frame.c: the enhancement layer's workers wait for the base layer's picture
    of their access unit through ff_thread_await_il_progress, and for the base-layer rows whose motion field they scale): every worker records
    its own picture per layer, pictures reach the engine in frame-start order (a base-layer picture before the enhancement-layer picture that
    up-samples it).  Against the reference library on ONE thread, three runs"""
    data, _ = streamgen.write_stream(w, h, seed, shvc_el_width=ew, shvc_el_height=eh, **kw)
    with refdec.captured_stderr():
        want = refdec.decode(data)
    assert [p[0].shape for p in want] == [(eh, ew)] * kw["n_pictures"]
    for run in range(3):
        with refdec.captured_stderr():
            got = refdec.decode(data, threads=threads, thread_type=1, L=refdec.hip_lib())
        assert len(got) == len(want), (run, len(got), len(want))
        for k, (a, b) in enumerate(zip(want, got)):
            for c in range(3):
                assert a[c].shape == b[c].shape and np.array_equal(a[c], b[c]), f"{name}: run {run}, output picture {k} plane {c} differs from the reference library's"


@pytest.mark.gpu
@need_lib
def test_base_layer_alone_of_a_two_layer_stream():
    """libOpenHevcSetActiveDecoders(h, 0): only the base layer is decoded (openHevcWrapper.c:405-414) — under frame threads too"""
    kw = dict(n_pictures=6, gop=2)
    data, _ = streamgen.write_stream(192, 128, 58, shvc_el_width=384, shvc_el_height=256, **kw)
    base, _ = streamgen.write_stream(192, 128, 58, **kw)
    want = refdec.decode(base)
    got = refdec.decode(data, threads=3, thread_type=1, L=refdec.hip_lib(), active_decoders=0)
    assert len(got) == len(want) == 6
    for a, b in zip(want, got):
        for c in range(3):
            assert np.array_equal(a[c], b[c])


@pytest.mark.gpu
@need_lib
def test_the_decoders_own_md5_check_judges_the_engines_pictures():
    """libOpenHevcSetCheckMD5(1): the decoder's "Correct MD5 (poc, plane)" lines (hevc.c:4146-4169) come from digests computed on the
    GPU over the engine's pictures against the stream's picture-hash SEI; a corrupted SEI digest is reported as Incorrect"""
    data, aus = streamgen.write_stream(416, 240, 41, n_pictures=6, gop=2, bit_depth=10)
    pics = refdec.decode(data)                                # low delay: output order = decode order
    digests = [refdec.md5_of(p) for p in pics]
    with_sei, _ = streamgen.add_md5(data, aus, digests)
    refdec.hip_lib().libOpenHevcSetDebugMode.argtypes = refdec.lib().libOpenHevcSetDebugMode.argtypes
    with refdec.captured_stderr() as cap:
        got = refdec.decode(with_sei, check_md5=True, L=refdec.hip_lib(), keep=False)
    assert len(got) == len(pics)
    assert cap.text.count("Correct MD5") == 3 * len(pics) and "Incorrect MD5" not in cap.text, cap.text[-2000:]
    wrong = [list(d) for d in digests]
    wrong[2][1] = bytes(16)
    bad, _ = streamgen.add_md5(data, aus, wrong)
    with refdec.captured_stderr() as cap:
        refdec.decode(bad, check_md5=True, L=refdec.hip_lib(), keep=False)
    assert cap.text.count("Incorrect MD5") == 1 and cap.text.count("Correct MD5") == 3 * len(pics) - 1, cap.text[-2000:]
