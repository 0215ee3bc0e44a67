"""Whole streams through the reference's own decoder (container only: oracle/_ref is built from /root/reference).

1. What the stream writer (openhevc_amd/synth/stream.c) wrote is what the reference PARSES, syntax element by syntax element
   (oracle/ref_trace_unit.c logs the reference's CABAC calls).
2. INTEGRATION.md for real: the reference's CTU loop with this repository's recording hooks in its DSP tables
   (oracle/ref_hooked_unit.c) yields work lists; the checker's pictures from those lists equal the pictures the UNMODIFIED
   reference decoder outputs — bit for bit, for every tool the writer covers.  This pins what earlier rounds could only restate:
   the MC drivers (hevc.c:1641-1949), de-quantisation (hevc_cabac.c:1478-1494, 1818-1841), neighbour availability
   (hevc.c:2592-2642), boundary strengths, SAO parsing, PCM / bypass, slices, tiles, wavefronts.
3. The picture-hash SEI the writer appends is accepted by the reference's decode-checksum option."""
import ctypes as C

import numpy as np
import pytest

import refdec
import streamgen
from openhevc_amd import frame as F
from oracle_lib import host_pic_array, oracle

pytestmark = pytest.mark.skipif(not refdec.have_refdec(), reason="reference tree / oracle/_ref not present")

CASES = [
    # name, w, h, seed, writer parameters
    ("i_64", 64, 64, 1, dict(n_pictures=1, gop=0, sao=0, amp=0)),
    ("intra", 416, 240, 7, dict(n_pictures=2, gop=0)),
    ("low_delay_p", 416, 240, 7, dict(n_pictures=3, gop=1)),
    ("low_delay_b", 416, 240, 7, dict(n_pictures=4, gop=2)),
    ("pcm_bypass_tskip_qpdelta", 416, 240, 9, dict(n_pictures=3, gop=2, pcm=1, transquant_bypass=1, transform_skip=1, cu_qp_delta=1)),
    ("main10_ctb32", 416, 240, 9, dict(n_pictures=3, gop=2, bit_depth=10, log2_ctb_size=5, pcm=1, transform_skip=1)),
    ("ctb16", 416, 240, 9, dict(n_pictures=3, gop=2, log2_ctb_size=4, log2_max_tb_size=4, cu_qp_delta=1)),
    ("slices", 416, 240, 9, dict(n_pictures=3, gop=2, n_slices=4)),
    ("tiles", 416, 240, 9, dict(n_pictures=3, gop=2, tile_cols=3, tile_rows=2, log2_ctb_size=5)),
    ("wpp", 416, 240, 9, dict(n_pictures=3, gop=2, wpp=1, log2_ctb_size=5)),
    ("wpp_slices_ctb16", 416, 240, 9, dict(n_pictures=3, gop=2, wpp=1, n_slices=3, log2_ctb_size=4, log2_max_tb_size=4)),
    ("tiles_slices_no_lf_across", 416, 240, 9, dict(n_pictures=3, gop=2, tile_cols=2, tile_rows=2, n_slices=3, log2_ctb_size=4, log2_max_tb_size=4,
                                                  lf_across_tiles=0, lf_across_slices=0)),
    ("weighted_cip_lists", 416, 240, 9, dict(n_pictures=3, gop=1, weighted_pred=1, scaling_list=1, constrained_intra_pred=1)),
    ("tmvp_3refs", 416, 240, 9, dict(n_pictures=4, gop=2, tmvp=1, cabac_init_present=1, deblocking_override=1, n_refs=3)),
    ("partial_ctbs", 200, 136, 5, dict(n_pictures=3, gop=2, n_slices=2, pcm=1)),
    ("far_mvd", 264, 200, 6, dict(n_pictures=3, gop=2, mvd_range=4000, skip_pct=10, merge_pct=10)),
    # format range extensions, 4:4:4: chroma blocks of luma size (4x4 included), a chroma mode per partition, cross-component prediction
    ("rext444", 416, 240, 7, dict(n_pictures=3, gop=2, chroma_format_idc=3, transquant_bypass=1)),
    ("rext444_ccp", 416, 240, 8, dict(n_pictures=3, gop=2, chroma_format_idc=3, cross_component_pred=1, transform_skip=1, scaling_list=1)),
    ("rext444_ccp_10_intra_ctb16", 264, 200, 9, dict(n_pictures=2, gop=0, bit_depth=10, chroma_format_idc=3, cross_component_pred=1, log2_ctb_size=4, log2_max_tb_size=4)),
    ("rext444_ccp_slices_p", 200, 136, 10, dict(n_pictures=3, gop=1, chroma_format_idc=3, cross_component_pred=1, n_slices=3, weighted_pred=1, log2_ctb_size=5)),
    # range-extension coding tools (sps / pps_range_extension), on 4:2:0 and 4:4:4: transform skip up to 32x32 with the rotation of 4x4
    # intra blocks and its own significance context, implicit / explicit residual DPCM on bypass and skip blocks, no intra smoothing,
    # persistent Rice adaptation
    ("rext_tskip_rdpcm", 416, 240, 11, dict(n_pictures=3, gop=2, transform_skip=1, transquant_bypass=1, tskip_rotation=1, tskip_context=1,
                                            implicit_rdpcm=1, explicit_rdpcm=1, log2_max_tskip_size=5, tskip_pct=50, bypass_pct=25)),
    ("rext_rice_nosmooth_10", 264, 200, 12, dict(n_pictures=3, gop=1, bit_depth=10, persistent_rice=1, intra_smoothing_disabled=1, transform_skip=1,
                                                 transquant_bypass=1, explicit_rdpcm=1, log2_max_tskip_size=3, qp=20, coeff_density=60)),
    ("rext444_tools_intra", 200, 136, 13, dict(n_pictures=2, gop=0, chroma_format_idc=3, cross_component_pred=1, transform_skip=1, transquant_bypass=1,
                                               tskip_rotation=1, tskip_context=1, implicit_rdpcm=1, persistent_rice=1, log2_max_tskip_size=4,
                                               tskip_pct=40, bypass_pct=30)),
    ("rext_tools_tiles_slices", 416, 240, 14, dict(n_pictures=3, gop=2, tile_cols=2, tile_rows=2, n_slices=3, log2_ctb_size=5, transform_skip=1,
                                                   transquant_bypass=1, explicit_rdpcm=1, implicit_rdpcm=1, tskip_rotation=1, persistent_rice=1,
                                                   log2_max_tskip_size=4, tskip_pct=40)),
    # sign data hiding (the sign of a sub-block's first coefficient is the parity of its magnitudes), also next to the tools that switch it off per block
    ("sdh", 416, 240, 17, dict(n_pictures=3, gop=2, sign_data_hiding=1, coeff_density=100, cbf_pct=80)),
    ("sdh_10_tools", 264, 200, 18, dict(n_pictures=3, gop=1, bit_depth=10, sign_data_hiding=1, transform_skip=1, transquant_bypass=1, implicit_rdpcm=1,
                                        explicit_rdpcm=1, tskip_rotation=1, scaling_list=2, cu_qp_delta=1, log2_max_tskip_size=4)),
    # 4:2:2 (format range extensions): two chroma blocks per transform unit one above the other, two chroma cbf flags, the 4:2:2 chroma mode mapping
    ("rext422_intra", 264, 200, 31, dict(n_pictures=2, gop=0, chroma_format_idc=2)),
    ("rext422_b_tools", 264, 200, 31, dict(n_pictures=3, gop=2, chroma_format_idc=2, transform_skip=1, transquant_bypass=1, cu_qp_delta=1, tmvp=1)),
    # ... with 16x16 CTBs (8 chroma samples wide): the SAO of a CTB in the last two CTB rows whose neighbour's neighbour is the last CTB column
    ("rext422_p_ctb16_slices_weighted_10", 264, 200, 31, dict(n_pictures=3, gop=1, chroma_format_idc=2, bit_depth=10, log2_ctb_size=4, log2_max_tb_size=4,
                                                              n_slices=2, weighted_pred=1)),
    # tiles with 16x16 CTBs and subsampled chroma: the order of the reference's filter calls follows the tile scan (OhFrame.sao_pending)
    ("tiles_ctb16_420", 296, 168, 521581, dict(n_pictures=3, gop=0, log2_ctb_size=4, log2_max_tb_size=4, n_slices=2, sao_pct=90, tile_cols=2, tile_rows=2, lf_across_tiles=0)),
    ("tiles_ctb16_422_10", 128, 160, 347975, dict(n_pictures=2, gop=1, chroma_format_idc=2, bit_depth=10, log2_ctb_size=4, log2_max_tb_size=4, n_slices=2, sao_pct=90,
                                                  tile_cols=2, tile_rows=2, lf_across_tiles=0, deblocking_override=1)),
    ("tiles3x3_ctb16_420_hier", 152, 88, 806835, dict(n_pictures=5, gop=3, log2_ctb_size=4, log2_max_tb_size=4, n_slices=3, sao_pct=90, tile_cols=3, tile_rows=3,
                                                       lf_across_tiles=1, deblocking_override=1)),
    # dependent slice segments (a header that carries the address only; CABAC states carried over from the segment before, reloaded at a
    # CTB-row start with wavefronts, fresh at a tile start), with slices / wavefronts / tiles
    ("dependent_slices", 416, 240, 41, dict(n_pictures=3, gop=2, n_slices=5, dependent_slices=1)),
    ("dependent_slices_wpp", 416, 240, 41, dict(n_pictures=3, gop=2, n_slices=4, dependent_slices=1, wpp=1, log2_ctb_size=5)),
    ("dependent_slices_tiles_ctb16", 416, 240, 41, dict(n_pictures=3, gop=1, n_slices=4, dependent_slices=1, tile_cols=2, tile_rows=2, log2_ctb_size=4,
                                                        log2_max_tb_size=4, lf_across_slices=0)),
    # smallest coding block 16x16 / 32x32 (inter NxN partitions, coarser min-PU / QP / PCM maps), PCM with 4:2:2 and 4:4:4 chroma
    ("min_cb16_b", 416, 256, 43, dict(n_pictures=3, gop=2, log2_min_cb_size=4, tmvp=1, cu_qp_delta=1)),
    ("min_cb16_ctb16_pcm_slices", 416, 256, 43, dict(n_pictures=3, gop=2, log2_min_cb_size=4, log2_ctb_size=4, log2_max_tb_size=4, cu_qp_delta=1, pcm=1, n_slices=2)),
    ("min_cb32_ctb64_pcm_bypass_10", 416, 256, 43, dict(n_pictures=2, gop=2, log2_min_cb_size=5, log2_ctb_size=6, pcm=1, transquant_bypass=1, bit_depth=10)),
    ("min_tb8_min_cb16", 416, 256, 45, dict(n_pictures=3, gop=2, log2_min_cb_size=4, log2_min_tb_size=3, scaling_list=2, sign_data_hiding=1)),
    ("min_tb16_min_cb32_444_ccp", 416, 256, 45, dict(n_pictures=2, gop=1, log2_min_cb_size=5, log2_min_tb_size=4, log2_ctb_size=6, log2_max_tb_size=5, chroma_format_idc=3,
                                                     cross_component_pred=1)),
    # 12 and 9 bit
    ("main12_tools", 264, 200, 47, dict(n_pictures=3, gop=2, bit_depth=12, pcm=1, transform_skip=1, transquant_bypass=1, weighted_pred=1)),
    ("rext444_12_ccp_ctb16_intra", 264, 200, 47, dict(n_pictures=2, gop=0, bit_depth=12, chroma_format_idc=3, cross_component_pred=1, log2_ctb_size=4, log2_max_tb_size=4)),
    ("nine_bit_lists_qpdelta", 264, 200, 47, dict(n_pictures=3, gop=2, bit_depth=9, scaling_list=2, cu_qp_delta=1)),
    ("pcm_422", 264, 200, 44, dict(n_pictures=3, gop=2, chroma_format_idc=2, pcm=1, pcm_pct=30)),
    ("pcm_444_ccp_min_cb16", 416, 256, 43, dict(n_pictures=3, gop=2, chroma_format_idc=3, pcm=1, pcm_pct=30, log2_min_cb_size=4, cross_component_pred=1)),
    # hierarchical B (decode order != output order, two pictures of reordering, sub-layer non-reference pictures, references from the future)
    ("hier_b", 416, 240, 15, dict(n_pictures=9, gop=3)),
    ("hier_b_tmvp_weighted_10_idr", 264, 200, 16, dict(n_pictures=11, gop=3, bit_depth=10, tmvp=1, weighted_pred=1, n_refs=3, idr_period=6, n_slices=2)),
]
IDS = [c[0] for c in CASES]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_writer_syntax_is_what_the_reference_parses(case):
    _, w, h, seed, kw = case
    data, aus = streamgen.write_stream(w, h, seed, trace=1, **kw)
    wrote = streamgen.written_trace()
    assert [tuple(a) for a in refdec.split_access_units(data)] == [tuple(a) for a in aus]
    parsed = refdec.parsed_trace(data)
    for flag in (7, 13):                                      # end_of_slice / pcm_flag: the reference returns a non-zero position for "1"
        parsed[parsed[:, 0] == flag, 1] = parsed[parsed[:, 0] == flag, 1] != 0
    assert len(wrote) == len(parsed), (len(wrote), len(parsed))
    bad = np.nonzero((wrote != parsed).any(axis=1))[0]
    assert len(bad) == 0, f"element {bad[0]}: wrote {streamgen.SE_NAMES[wrote[bad[0], 0]]}={wrote[bad[0], 1]}, parsed {streamgen.SE_NAMES[parsed[bad[0], 0]]}={parsed[bad[0], 1]}"
    assert len(wrote) > 20


def decode_through_hooks(data, threads=1, thread_type=1, bs_from_motion=False):
    """pictures reconstructed by the CHECKER from the work lists the hooked reference decoder records, in OUTPUT order
    (ascending POC inside an IDR period — the order the plain decoder hands its pictures out in)"""
    pics, got, keys = {}, [], []

    def on_picture(f, cur, poc):
        for i in [cur] + [f.ref_pics[k] for k in range(F.OH_MAX_REFS) if f.ref_pics[k] >= 0]:
            if i not in pics:
                pics[i] = F.HostPic(f.p)
        assert oracle().oh_or_frame(C.byref(f), host_pic_array(pics)) == 0
        got.append([pics[cur].visible(c).copy() for c in range(3)])
        keys.append((sum(1 for _, q in keys if q == 0) + (poc == 0), poc))       # POC 0 only at IDR pictures in the written streams
    refdec.record_work_lists(data, on_picture, threads, thread_type, bs_from_motion)
    return [got[k] for k in sorted(range(len(got)), key=lambda k: keys[k])]


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_hooked_ctu_loop_reproduces_the_reference_decoder(case):
    _, w, h, seed, kw = case
    data, _ = streamgen.write_stream(w, h, seed, **kw)
    want = refdec.decode(data)
    got = decode_through_hooks(data)
    assert len(got) == len(want) == kw["n_pictures"]
    for k in range(len(want)):
        for c in range(3):
            assert np.array_equal(want[k][c], got[k][c]), (case[0], "picture", k, "plane", c)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_boundary_strengths_from_the_decoders_motion_field(case):
    """SURVEY §8 f2 on real streams: the hooked decoder hands over what ff_hevc_deblocking_boundary_strengths READS — its motion
    field, cbf_luma and the (position, block size) of every call (hevc.c:1578, 1607, 2400, 2484) — instead of the grids it writes
    (left zero); the checker derives the boundary strengths and must still reproduce the plain decoder's pictures"""
    _, w, h, seed, kw = case
    data, _ = streamgen.write_stream(w, h, seed, **kw)
    want = refdec.decode(data)
    got = decode_through_hooks(data, bs_from_motion=True)
    assert len(got) == len(want)
    for k in range(len(want)):
        for c in range(3):
            assert np.array_equal(want[k][c], got[k][c]), (case[0], "picture", k, "plane", c)


def test_picture_hash_sei_is_accepted_by_the_reference():
    data, aus = streamgen.write_stream(264, 200, 21, n_pictures=3, gop=2, bit_depth=10)
    pics = refdec.decode(data)
    with_sei, aus2 = streamgen.add_md5(data, aus, [refdec.md5_of(p) for p in pics])
    assert len(with_sei) == len(data) + 3 * (4 + 2 + 51 + 1)                      # start code, NAL header, payload, trailing byte
    assert [tuple(a) for a in refdec.split_access_units(with_sei)] == [tuple(a) for a in aus2]      # a suffix SEI stays in its picture's AU
    with refdec.captured_stderr() as log:                                           # hevc.c:4144-4162: the verdict is logged per plane
        again = refdec.decode(with_sei, check_md5=True)
    assert all(np.array_equal(a[c], b[c]) for a, b in zip(pics, again) for c in range(3))
    assert log.text.count("Correct MD5") == 9 and "Incorrect MD5" not in log.text
    wrong = [refdec.md5_of(p) for p in pics]
    wrong[1] = [bytes(16)] * 3
    bad, _ = streamgen.add_md5(data, aus, wrong)
    with refdec.captured_stderr() as log:
        refdec.decode(bad, check_md5=True)
    assert log.text.count("Incorrect MD5") == 3 and log.text.count("Correct MD5") == 6


def test_reference_threading_modes_agree():
    """the reference's frame / slice threads (pthread_frame.c, pthread_slice.c incl. its wavefront entry points) output what
    its single-threaded decode outputs: the streams exercise them too"""
    data, _ = streamgen.write_stream(416, 240, 31, n_pictures=4, gop=2, wpp=1, log2_ctb_size=4, log2_max_tb_size=4)
    one = refdec.decode(data)
    for threads, kind in ((4, 1), (4, 2)):
        other = refdec.decode(data, threads=threads, thread_type=kind)
        assert len(other) == len(one)
        assert all(np.array_equal(a[c], b[c]) for a, b in zip(one, other) for c in range(3)), (threads, kind)


@pytest.mark.parametrize("bit_depth", [8, 10])
def test_reference_sse_build_agrees_with_its_c_build(bit_depth):
    """oracle/_ref/libopenhevc_ref_sse.so (the tree's SSE4 intrinsics in the tables, bench.py's cpu_baseline.sse_*) decodes what the C
    build decodes, bit for bit: the baseline timed is a correct decoder"""
    for seed in (41, 42):
        data, _ = streamgen.write_stream(416, 240, seed, n_pictures=5, gop=2, bit_depth=bit_depth, amp=1, pcm=1, transform_skip=1,
                                         weighted_pred=int(seed == 42), wpp=1)
        want = refdec.decode(data)
        got = refdec.decode(data, L=refdec.sse_lib(), threads=3, thread_type=2)
        assert len(got) == len(want) == 5
        assert all(np.array_equal(a[c], b[c]) for a, b in zip(want, got) for c in range(3))


def test_committed_fixtures_are_current():
    """tests/golden/streams/*.npz = what make_stream_golden.py produces today (writer, hooks and reference unchanged)"""
    import hashlib
    import os
    import sys
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    import make_stream_golden as M
    for name, w, h, seed, kw in M.STREAM_CASES:
        have = np.load(os.path.join(gold, "streams", name + ".npz"))
        data, _ = streamgen.write_stream(w, h, seed, **kw)
        assert hashlib.md5(data).digest() == have["stream_md5"].tobytes(), name
        want = refdec.decode(data)
        rank = streamgen.output_rank(len(want), kw.get("gop", 2), kw.get("idr_period", 0))       # the fixture keeps decode order
        assert b"".join(b"".join(refdec.md5_of(want[rank[i]])) for i in range(len(want))) == have["md5"].tobytes(), name


def test_conformance_window_is_cropped_by_the_reference_output():
    """the writer's conformance window reaches the reference's output path: its pictures are the coded pictures (here: reconstructed by
    the checker from the recorded work lists) cropped by the window — the data the engine's oh_pic_download_window is pinned on (-m gpu)"""
    left, right, top, bottom = 6, 10, 4, 12
    data, _ = streamgen.write_stream(416, 240, 97, n_pictures=3, gop=2, conf_win_left=left, conf_win_right=right, conf_win_top=top, conf_win_bottom=bottom)
    want = refdec.decode(data)
    full = decode_through_hooks(data)
    for k in range(3):
        for c in range(3):
            s = 1 if c else 0
            crop = full[k][c][top >> s:(240 >> s) - (bottom >> s), left >> s:(416 >> s) - (right >> s)]
            assert np.array_equal(crop, want[k][c]), (k, c)


@pytest.mark.parametrize("kw", [dict(wpp=1, log2_ctb_size=4, log2_max_tb_size=4), dict(wpp=1, log2_ctb_size=5, n_slices=2, weighted_pred=1, pcm=1),
                                dict(wpp=1, log2_ctb_size=4, log2_max_tb_size=4, chroma_format_idc=3, cross_component_pred=1)],
                         ids=["wpp_ctb16", "wpp_slices_weighted_pcm", "wpp_444_ccp"])
def test_recording_slots_under_the_reference_wavefront_threads(kw):
    """the reference's slice threads (pthread_slice.c: one CTU row per job, hls_decode_entry_wpp on 4 worker threads) call the recording
    slots concurrently: workers adopt the picture's binding, the recorder takes its lock, every PU is complete when its last slot call
    returns — and the recorded work lists still reconstruct the reference decoder's pictures (the items arrive in another order)"""
    data, _ = streamgen.write_stream(416, 240, 33, n_pictures=4, gop=2, **kw)
    want = refdec.decode(data)
    for rep in range(3):
        got = decode_through_hooks(data, threads=4, thread_type=2)
        assert len(got) == len(want) == 4
        for k in range(4):
            for c in range(3):
                assert np.array_equal(want[k][c], got[k][c]), (rep, "picture", k, "plane", c)


def test_filter_call_order_simulation_equals_the_closed_form_in_raster_order():
    """16x16 CTBs, subsampled chroma: which horizontal chroma edges the reference's CTB driver had NOT filtered yet in the right
    neighbour's first column when it ran a CTB's SAO.  The engine and the checker use a closed form for pictures decoded in raster
    order and, for tiled pictures, the bits oh_sao_pending_driver derives by replaying ff_hevc_hls_filters over the decoding order —
    with one tile the replay must give the closed form, for every picture size in CTBs"""
    H = F.host()
    H.oh_sao_pending_driver.argtypes = [C.POINTER(C.c_int32), C.c_int, C.c_int, C.POINTER(C.c_uint8)]
    for W in range(1, 10):
        for Hh in range(1, 9):
            tid, out = (C.c_int32 * (W * Hh))(), (C.c_uint8 * (W * Hh))()
            H.oh_sao_pending_driver(tid, W, Hh, out)
            for cy in range(Hh):
                for cx in range(W):
                    want = 0
                    if cx + 2 < W:
                        r_lag = Hh - 1 if cx + 2 == W - 1 else (Hh - 2 if Hh >= 2 else 0)
                        rp = min(cy + 1, r_lag)
                        want = (1 if cy >= rp else 0) | (2 if cy + 1 >= rp and cy + 1 < Hh else 0)
                    assert out[cy * W + cx] == want, (W, Hh, cx, cy)


def test_random_small_ctb_streams():
    """a seeded sweep over 16x16-CTB streams (every chroma format, slices, tiles up to 3 x 3, all GOP shapes): the configuration in
    which the reference's filter-call order shows in the output"""
    import random
    rng = random.Random(2024)
    for _ in range(16):
        w, h = 8 * rng.randint(4, 36), 8 * rng.randint(3, 26)
        kw = dict(n_pictures=rng.choice([2, 3]), gop=rng.choice([0, 1, 2, 3]), chroma_format_idc=rng.choice([1, 2, 2, 3]), bit_depth=rng.choice([8, 10]),
                  log2_ctb_size=4, log2_max_tb_size=4, n_slices=rng.choice([1, 2, 3]), sao_pct=90)
        if rng.random() < 0.6:
            kw.update(tile_cols=rng.choice([2, 3]), tile_rows=rng.choice([1, 2, 3]), lf_across_tiles=rng.choice([0, 1]))
        if rng.random() < 0.3:
            kw.update(lf_across_slices=0)
        if kw["gop"] == 3:
            kw.update(n_pictures=5)
        test_hooked_ctu_loop_reproduces_the_reference_decoder(("sweep", w, h, rng.randint(1, 10 ** 6), kw))


# ---- SHVC: two-layer streams (SURVEY.md 8 row a30 on real streams) ----
SHVC_CASES = [
    ("x2", 96, 64, 192, 128, 61, dict(n_pictures=4, gop=2)),
    ("x1_5", 96, 64, 144, 96, 62, dict(n_pictures=4, gop=2, amp=1, transform_skip=1)),
    ("snr", 160, 96, 160, 96, 63, dict(n_pictures=3, gop=1, log2_ctb_size=5)),
    ("ratios_1_33_by_1_75", 96, 64, 128, 112, 64, dict(n_pictures=3, gop=2)),
    ("x2_idr_period", 64, 64, 128, 128, 65, dict(n_pictures=7, gop=2, idr_period=3, n_refs=2, tmvp=1)),
    ("x1_5_ctb16", 96, 64, 144, 96, 66, dict(n_pictures=3, gop=2, log2_ctb_size=4, log2_max_tb_size=4)),
]


def test_two_layer_writer_refuses_what_the_up_sampler_cannot_do():
    for kw in (dict(bit_depth=10), dict(chroma_format_idc=3), dict(gop=3), dict(conf_win_left=2), dict(transform_skip=1, log2_max_tskip_size=4),
               dict(implicit_rdpcm=1)):
        with pytest.raises(ValueError):
            streamgen.write_stream(64, 64, 1, n_pictures=2, shvc_el_width=128, shvc_el_height=128, **kw)
    with pytest.raises(ValueError):                           # an enhancement layer smaller than the base layer
        streamgen.write_stream(64, 64, 1, n_pictures=2, shvc_el_width=32, shvc_el_height=64)
    with pytest.raises(ValueError):                           # more than twice the base layer: the reference's two up-sampling paths disagree
        streamgen.write_stream(64, 64, 1, n_pictures=2, shvc_el_width=144, shvc_el_height=128)
    with pytest.raises(ValueError):                           # a single CTB row: the reference's block up-sampler emulates one picture edge per block
        streamgen.write_stream(64, 32, 1, n_pictures=2, shvc_el_width=128, shvc_el_height=64)
    with pytest.raises(ValueError):                           # not a multiple of the smallest coding block
        streamgen.write_stream(64, 64, 1, n_pictures=2, shvc_el_width=100, shvc_el_height=128)


@pytest.mark.parametrize("case", SHVC_CASES, ids=[c[0] for c in SHVC_CASES])
def test_two_layer_stream_is_what_the_reference_decodes(case):
    """every access unit holds a base-layer and an enhancement-layer picture (nuh_layer_id 0 / 1: the access-unit splitter keeps them
    together); the reference decodes both layers without complaint and releases the enhancement layer's pictures"""
    _, w, h, ew, eh, seed, kw = case
    data, aus = streamgen.write_stream(w, h, seed, shvc_el_width=ew, shvc_el_height=eh, **kw)
    assert [tuple(a) for a in refdec.split_access_units(data)] == [tuple(a) for a in aus]
    layers = []
    for a, b in aus:
        au, ls, i = data[a:b], [], 0
        while (i := au.find(b"\x00\x00\x01", i)) >= 0:
            if (au[i + 3] >> 1) & 63 < 32:                    # a slice segment: its nuh_layer_id
                ls.append(((au[i + 3] & 1) << 5) | (au[i + 4] >> 3))
            i += 3
        layers.append(ls)
    assert all(ls == [0, 1] for ls in layers), layers
    with refdec.captured_stderr() as cap:
        pics = refdec.decode(data)
    assert "rror" not in cap.text.replace("Could not find ref with POC", ""), cap.text[-1500:]
    # the wrapper exposes the highest layer that released a picture (openHevcWrapper.c:139-152): the enhancement layer's, every time
    assert [p[0].shape for p in pics] == [(eh, ew)] * kw["n_pictures"]


def decode_layers_through_hooks(data):
    """both layers' pictures reconstructed by the CHECKER from the work lists of the hooked reference decoder, in decode order:
    [(layer, poc, planes)].  The inter-layer reference picture is made with the checker's whole-picture up-sampling
    (oracle.c oh_or_upsample_frame) from the base layer's reconstructed picture before the enhancement layer's work list runs."""
    from test_upsample_vs_ref import host_pic
    pics, out = ({}, {}), []

    def on_picture(layer, f, cur, poc, il):
        mine = pics[layer]
        for i in [cur] + [f.ref_pics[k] for k in range(F.OH_MAX_REFS) if f.ref_pics[k] >= 0]:
            if i not in mine or (mine[i].params.width, mine[i].params.height) != (f.p.width, f.p.height):
                mine[i] = F.HostPic(f.p)
        assert (il is not None) == (layer == 1)
        if il is not None:
            slot, bl_id, up = il
            assert f.n_pu > 0 and all(f.pu[k].ref[0] == slot and f.pu[k].mv[0][0] == 0 and f.pu[k].mv[0][1] == 0 for k in range(min(f.n_pu, 50)))     # inter-layer prediction is what its PUs do
            bl, ilr = pics[0][bl_id], mine[f.ref_pics[slot]]
            hb, he = host_pic(bl.params, bl.planes), host_pic(ilr.params, ilr.planes)
            assert oracle().oh_or_upsample_frame(C.byref(hb), C.byref(he), C.byref(up)) == 0
        assert oracle().oh_or_frame(C.byref(f), host_pic_array(mine)) == 0
        out.append((layer, poc, [mine[cur].visible(c).copy() for c in range(3)]))
    n = refdec.record_layer_work_lists(data, on_picture)
    return n, out


@pytest.mark.parametrize("case", SHVC_CASES, ids=[c[0] for c in SHVC_CASES])
def test_hooked_ctu_loops_of_both_layers_reproduce_the_reference_decoder(case):
    """the reference's two decoders with the RECORDING slots: the base layer's and the enhancement layer's work lists, the inter-layer
    reference picture made by the checker's up-sampling, give the pictures the plain reference library outputs for both layers"""
    _, w, h, ew, eh, seed, kw = case
    data, _ = streamgen.write_stream(w, h, seed, shvc_el_width=ew, shvc_el_height=eh, **kw)
    with refdec.captured_stderr():
        want = refdec.decode(data)
        n, got = decode_layers_through_hooks(data)
    assert n == [kw["n_pictures"], kw["n_pictures"]]
    base, _ = streamgen.write_stream(w, h, seed, **kw)       # the same base layer as a one-layer stream: what the base-layer decoder reconstructs
    for layer, theirs in ((0, refdec.decode(base)), (1, want)):
        mine = [p for l, _, p in got if l == layer]
        assert len(mine) == len(theirs) == kw["n_pictures"]
        for k, (t, m) in enumerate(zip(theirs, mine)):       # low delay: output order = decode order
            for c in range(3):
                assert t[c].shape == m[c].shape and np.array_equal(t[c], m[c]), (case[0], "layer", layer, "picture", k, "plane", c)
