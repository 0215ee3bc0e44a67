"""-m gpu: the byte-stream side on the MI355X — picture MD5s computed on the GPU against hashlib, and the command-line harness
(openhevc_amd/ohevc_dec) decoding written streams end to end: Annex-B file -> access units -> front end (the reference's host
decoder with this repository's recording table slots linked in: oracle/_ref/libopenhevc_hooked.so, built in the container)
-> work lists -> GPU passes -> GPU MD5 against the stream's picture-hash SEI, whose digests come from the reference decoder."""
import hashlib
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from openhevc_amd import frame as F

pytestmark = pytest.mark.gpu


def params(w, h, bd, cf):
    return F.pic_params(w, h, bit_depth=bd, chroma_format_idc=cf)


@pytest.mark.parametrize("w,h,bd,cf", [(64, 64, 8, 1), (416, 240, 8, 1), (416, 240, 10, 1), (200, 136, 10, 3), (264, 200, 8, 2), (72, 40, 8, 0),
                                        (1920, 1080, 10, 1), (8, 8, 8, 1), (24, 8, 8, 1)])
def test_picture_md5_on_the_gpu_is_hashlib_md5(w, h, bd, cf):
    """oh_pics_md5 == MD5 of the packed rows of each plane (calc_md5, hevc.c:4623-4638), several pictures per launch; the sizes cover
    messages that end exactly on a block boundary, one-block messages (8x8 chroma = 16 bytes) and rows that are not a multiple of 64 bytes"""
    from openhevc_amd.engine import Engine
    rng = np.random.default_rng(w * 131 + h * 7 + bd + cf)
    eng = Engine(0)
    p = params(w, h, bd, cf)
    pids, want = [], []
    for k in range(3):
        hp = F.HostPic(p)
        for c in range(F.n_planes(p)):
            v = hp.visible(c)
            v[...] = rng.integers(0, 1 << bd, v.shape, dtype=v.dtype)
        pid = eng.pic_alloc(p)
        eng.pic_upload(pid, hp)
        pids.append(pid)
        want.append([hashlib.md5(np.ascontiguousarray(hp.visible(c)).tobytes()).digest() if c < F.n_planes(p) else bytes(16) for c in range(3)])
    got = eng.pics_md5(pids)
    eng.close()
    assert got == want


def test_picture_md5_follows_the_finished_half():
    """after a work list with SAO the finished picture lives in the other half of the allocation: the hash is of what oh_pic_download returns"""
    from openhevc_amd.engine import Engine, remap_frame
    eng = Engine(0)
    rec = F.Recorder(params(416, 240, 8, 1))
    f = rec.synth(F.synth_params(0, 3, sao_pct=90), 0)
    pid = eng.pic_alloc(f.p)
    eng.frame_submit(remap_frame(f, {0: pid}))
    assert eng.pic_final_half(pid) == 1
    hp = eng.pic_download(pid, f.p)
    want = [hashlib.md5(np.ascontiguousarray(hp.visible(c)).tobytes()).digest() for c in range(3)]
    assert eng.pics_md5([pid]) == [want]
    eng.close()
    rec.close()


HOOKED = os.path.join(ROOT, "oracle", "_ref", "libopenhevc_hip.so")          # the drop-in library: the reference's decoder + wrapper, engine inside
REFDEC = os.path.join(ROOT, "oracle", "_ref", "libopenhevc_ref.so")
HARNESS = os.path.join(ROOT, "openhevc_amd", "ohevc_dec")


@pytest.fixture
def front_end():
    """the hooked reference decoder (front end of the harness), the plain one (checker) and the harness binary.  These tests are
    -m gpu: on a GPU box the three files must have travelled with the snapshot (oracle/_ref/ is git-ignored, not gpurun-ignored) — a
    run without them would be 16 tests short and look just as green, so their absence FAILS here instead of skipping."""
    missing = [f for f in (HOOKED, REFDEC, HARNESS) if not os.path.exists(f)]
    assert not missing, f"did not travel to the GPU box (build them in the container: python -c 'import __graft_entry__ as g; g.build()'): {missing}"


need_front_end = pytest.mark.usefixtures("front_end")

STREAMS = [
    ("main8_lowdelay", 416, 240, 11, dict(n_pictures=8, gop=2)),
    ("main10_tools", 416, 240, 12, dict(n_pictures=6, gop=2, bit_depth=10, amp=1, pcm=1, transform_skip=1, transquant_bypass=1, weighted_pred=1, scaling_list=1)),
    ("slices_tiles", 416, 240, 13, dict(n_pictures=5, gop=1, n_slices=3, tile_cols=2, tile_rows=2, lf_across_tiles=0)),
    ("intra_ctb16", 264, 200, 14, dict(n_pictures=3, gop=0, log2_ctb_size=4, log2_max_tb_size=4)),
    ("hd_main10", 1920, 1080, 15, dict(n_pictures=4, gop=2, bit_depth=10, wpp=1)),
    ("rext444_ccp", 416, 240, 16, dict(n_pictures=4, gop=2, chroma_format_idc=3, cross_component_pred=1, transform_skip=1)),
    ("hier_b_reordered", 416, 240, 18, dict(n_pictures=11, gop=3, tmvp=1, n_refs=3, idr_period=7)),
    ("main10_window", 416, 240, 19, dict(n_pictures=4, gop=2, bit_depth=10, conf_win_left=6, conf_win_right=10, conf_win_top=4, conf_win_bottom=12)),
    ("rext422_tiles_ctb16", 264, 200, 20, dict(n_pictures=4, gop=2, chroma_format_idc=2, bit_depth=10, log2_ctb_size=4, log2_max_tb_size=4, tile_cols=2, tile_rows=2,
                                               n_slices=2, sao_pct=90)),
    ("rext_tools", 416, 240, 17, dict(n_pictures=5, gop=2, bit_depth=10, transform_skip=1, transquant_bypass=1, tskip_rotation=1, tskip_context=1, implicit_rdpcm=1,
                                      explicit_rdpcm=1, persistent_rice=1, intra_smoothing_disabled=1, log2_max_tskip_size=5, tskip_pct=45, bypass_pct=20)),
]


@need_front_end
@pytest.mark.parametrize("case", STREAMS, ids=[c[0] for c in STREAMS])
def test_harness_decodes_streams_and_the_gpu_md5_matches_the_sei(case, tmp_path):
    import refdec
    import streamgen
    name, w, h, seed, kw = case
    data, aus = streamgen.write_stream(w, h, seed, **kw)
    pics = refdec.decode(data)                                # output order
    n = kw["n_pictures"]
    rank = streamgen.output_rank(n, kw.get("gop", 2), kw.get("idr_period", 0))
    coded = pics
    if any(k.startswith("conf_win") for k in kw):            # the SEI hashes the CODED picture: the same stream written without the window
        coded = refdec.decode(streamgen.write_stream(w, h, seed, **{k: v for k, v in kw.items() if not k.startswith("conf_win")})[0])
        w, h = pics[0][0].shape[1], pics[0][0].shape[0]       # what the decoder hands out (and the harness reports) is the window
    digests = [refdec.md5_of(coded[rank[k]]) for k in range(n)]   # the SEI of an access unit describes ITS picture: decode order
    with_sei, _ = streamgen.add_md5(data, aus, digests)
    path = tmp_path / (name + ".bin")
    path.write_bytes(with_sei)
    out = tmp_path / "out.yuv"
    r = subprocess.run([HARNESS, "-i", str(path), "-F", HOOKED, "-n", "-o", str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("Correct MD5") == 3 * n and "Incorrect MD5" not in r.stdout
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith(f"frame= {n} fps= ") and last.endswith(f"video_size= {w}x{h}"), last
    # -o: the pictures themselves, in the order and inside the window the front end releases them for output
    raw = (tmp_path / f"out_{w}x{h}.yuv").read_bytes()
    assert raw == b"".join(np.ascontiguousarray(pl).tobytes() for p in pics for pl in p)
    # a wrong digest in the stream is reported per plane and fails the run (exit code 3)
    wrong = [list(d) for d in digests]
    wrong[1][2] = bytes(16)
    bad, _ = streamgen.add_md5(data, aus, wrong)
    path.write_bytes(bad)
    r = subprocess.run([HARNESS, "-i", str(path), "-F", HOOKED], capture_output=True, text=True, timeout=600)
    assert r.returncode == 3 and r.stdout.count("Incorrect MD5") == 1 and r.stdout.count("Correct MD5") == 3 * n - 1
    if kw.get("wpp"):                                        # the front end on 4 slice threads (wavefront entry points): same verdict
        r = subprocess.run([HARNESS, "-i", str(path), "-F", HOOKED, "-p", "4", "-f", "2"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 3 and r.stdout.count("Incorrect MD5") == 1 and r.stdout.count("Correct MD5") == 3 * n - 1
        r = subprocess.run([HARNESS, "-i", str(path), "-F", HOOKED, "-p", "4", "-f", "4"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 2                             # frame AND slice threads together are not what the recording slots support
    # the front end on 4 FRAME threads (pthread_frame.c: every worker records its own picture, hand-over in decode order): same verdict
    # (not for streams with PCM / bypass and the loop filter off: there the REFERENCE's own pictures depend on the thread configuration —
    # s->is_pcm is never cleared per picture and every frame thread's context keeps its own, hevc.c:147,1440 — so digests taken from its
    # single-threaded run do not describe its frame-threaded one either)
    if not (kw.get("pcm") or kw.get("transquant_bypass")):
      r = subprocess.run([HARNESS, "-i", str(path), "-F", HOOKED, "-p", "4", "-f", "1"], capture_output=True, text=True, timeout=600)
      assert r.returncode == 3 and r.stdout.count("Incorrect MD5") == 1 and r.stdout.count("Correct MD5") == 3 * n - 1, r.stdout[-1500:] + r.stderr[-1500:]
    # -b: the front end hands over its motion field, the engine derives the boundary strengths (bs_kernel): same verdict
    r = subprocess.run([HARNESS, "-i", str(path), "-F", HOOKED, "-b"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 3 and r.stdout.count("Incorrect MD5") == 1 and r.stdout.count("Correct MD5") == 3 * n - 1
    # -c: no check, -s: stop early
    r = subprocess.run([HARNESS, "-i", str(path), "-F", HOOKED, "-c", "-s", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "MD5" not in r.stdout and r.stdout.strip().splitlines()[-1].startswith("frame= 2 ")


@need_front_end
@pytest.mark.parametrize("bit_depth,win", [(8, (6, 10, 4, 12)), (10, (0, 2, 8, 0)), (8, (14, 0, 0, 2))], ids=["8bit", "10bit", "8bit_left_bottom"])
def test_output_window_equals_the_reference_decoder_output(bit_depth, win):
    """SURVEY §8 f4 pinned against the reference: a stream whose SPS carries a conformance window — the reference decoder outputs
    the cropped pictures (ff_hevc_output_frame hevc_refs.c:248-254 + libOpenHevcGetOutputCpy); the engine reconstructs the coded
    pictures from the recorded work lists and oh_pic_download_window must hand back exactly those planes"""
    import refdec
    import streamgen
    from openhevc_amd.engine import Engine, remap_frame
    left, right, top, bottom = win
    data, _ = streamgen.write_stream(416, 240, 90 + bit_depth, n_pictures=3, gop=2, bit_depth=bit_depth, conf_win_left=left, conf_win_right=right,
                                     conf_win_top=top, conf_win_bottom=bottom)
    want = refdec.decode(data)
    assert want[0][0].shape == (240 - top - bottom, 416 - left - right)
    eng = Engine(0)
    ids, got = {}, []

    def on_picture(f, cur, poc):
        for i in [cur] + [f.ref_pics[k] for k in range(F.OH_MAX_REFS) if f.ref_pics[k] >= 0]:
            if i not in ids:
                ids[i] = eng.pic_alloc(f.p)
        eng.frame_submit(remap_frame(f, ids))
        got.append(eng.pic_download_window(ids[cur], f.p, left=left, right=right, top=top, bottom=bottom, pad=64 if bit_depth == 8 else 0))
    assert refdec.record_work_lists(data, on_picture) == 3
    eng.close()
    for k in range(3):
        for c in range(3):
            assert np.array_equal(got[k][c], want[k][c]), (k, c)


@need_front_end
@pytest.mark.timeout(1100)
@pytest.mark.parametrize("name,w,h,kw", [
    ("4k_main10", 3840, 2160, dict(n_pictures=3, gop=2, bit_depth=10, wpp=1)),
    ("8k_444_10bit_ccp", 7680, 4320, dict(n_pictures=2, gop=1, bit_depth=10, chroma_format_idc=3, cross_component_pred=1)),
], ids=["4k_main10", "8k_444_10bit_ccp"])
def test_harness_at_baseline_geometries(name, w, h, kw, tmp_path):
    """BASELINE.json configs[3] (3840x2160 Main 10) and the range-extension half of configs[4] (7680x4320 4:4:4 10 bit) as real
    streams: written here, decoded by the reference decoder for the SEI digests, then file -> access units -> hooked reference decoder
    as front end -> work lists -> MI355X passes -> MD5 on the GPU: every plane of every picture "Correct MD5" """
    import refdec
    import streamgen
    data, aus = streamgen.write_stream(w, h, 77, **kw)
    pics = refdec.decode(data, threads=8, thread_type=2)
    assert len(pics) == kw["n_pictures"]
    threads = ["-p", "8", "-f", "2"] if kw.get("wpp") else []
    with_sei, _ = streamgen.add_md5(data, aus, [refdec.md5_of(p) for p in pics])
    del pics
    path = tmp_path / (name + ".bin")
    path.write_bytes(with_sei)
    r = subprocess.run([HARNESS, "-i", str(path), "-F", HOOKED, "-n"] + threads, capture_output=True, text=True, timeout=1000)
    n = kw["n_pictures"]
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("Correct MD5") == 3 * n and "Incorrect MD5" not in r.stdout
    assert r.stdout.strip().splitlines()[-1].startswith(f"frame= {n} fps= ")


@need_front_end
def test_random_streams_through_the_harness(tmp_path):
    """a seeded sweep over the stream writer's parameter space (tools/sweep_streams.py: sizes, chroma formats, CTB sizes, slices / tiles /
    wavefronts, every GOP shape, range-extension tools, windows ...): file -> ohevc_dec (hooked reference front end -> engine) must write
    exactly the pictures the unmodified reference decoder outputs, in its order and window"""
    import random
    import refdec
    import streamgen
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import sweep_streams
    rng = random.Random(77)
    done = 0
    for _ in range(40):
        w, h, seed, kw = sweep_streams.draw(rng)
        data, _ = streamgen.write_stream(w, h, seed, **kw)
        pics = refdec.decode(data)
        path = tmp_path / "s.bin"
        path.write_bytes(data)
        out = tmp_path / "o.yuv"
        for old in tmp_path.glob("o_*.yuv"):
            old.unlink()
        r = subprocess.run([HARNESS, "-i", str(path), "-F", HOOKED, "-c", "-o", str(out)], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (w, h, seed, kw, r.stdout[-1500:] + r.stderr[-1500:])
        ow, oh = pics[0][0].shape[1], pics[0][0].shape[0]
        raw = (tmp_path / f"o_{ow}x{oh}.yuv").read_bytes()
        assert raw == b"".join(np.ascontiguousarray(pl).tobytes() for p in pics for pl in p), (w, h, seed, kw)
        done += 1
    assert done == 40
