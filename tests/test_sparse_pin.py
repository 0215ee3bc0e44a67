"""SURVEY §8 f1 pinned against the reference: the SPARSE hand-over — quantised levels + QP + scaling matrix per block, de-quantised by
the engine (hevc_cabac.c:1478-1494, 1818-1841 moved to the GPU) — must reproduce the reference decoder's pictures.

The stream writer knows the levels it coded (OhStreamParams.levels); the reference decoder with the recording hooks gives the rest of
each picture's work list (with DENSE coefficients, i.e. after ITS de-quantisation).  Here every IDCT / DST / transform-skip block of
the recorded list is switched to the sparse form built from the writer's levels (its dense coefficients are zeroed, so nothing can
fall back on them), and the list is reconstructed: by the CPU checker here, by the engine in the -m gpu test.  Expected output = the
plain reference decoder's.  Scaling lists: flat, the default lists, and random lists coded in the SPS (as the reference's parser
holds them: ref_hooked_scaling_list)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import refdec
import streamgen
from openhevc_amd import frame as F

CASES = [
    ("flat8", 416, 240, 51, dict(n_pictures=4, gop=2, transform_skip=1, transquant_bypass=1, pcm=1)),
    ("flat10_intra", 264, 200, 52, dict(n_pictures=2, gop=0, bit_depth=10, transform_skip=1, qp=22)),
    ("default_lists10", 416, 240, 53, dict(n_pictures=4, gop=2, bit_depth=10, scaling_list=1, transform_skip=1)),
    ("coded_lists8", 416, 240, 54, dict(n_pictures=4, gop=2, scaling_list=2, transform_skip=1, transquant_bypass=1, qp=35)),
    ("coded_lists10_ctb16", 264, 200, 55, dict(n_pictures=3, gop=1, bit_depth=10, scaling_list=2, log2_ctb_size=4, log2_max_tb_size=4, qp=18)),
    # 4:4:4 with cross-component prediction: the chroma blocks' levels are de-quantised and transformed on the GPU and the scaled luma
    # residual is added there (cross_kernel) — the links come from the hooked decoder's shim
    ("ccp444_8", 264, 200, 56, dict(n_pictures=3, gop=2, chroma_format_idc=3, cross_component_pred=1, transform_skip=1, transquant_bypass=1)),
    ("sdh8_lists", 264, 200, 58, dict(n_pictures=3, gop=2, sign_data_hiding=1, scaling_list=2, transform_skip=1, cbf_pct=80)),
    # range-extension tools with scaling lists: skip blocks up to 32x32 stay flat (hevc_cabac.c:1485), the 4x4 intra rotation moves to the GPU with the levels
    ("rext_tools_lists10", 216, 184, 59, dict(n_pictures=3, gop=2, bit_depth=10, scaling_list=2, transform_skip=1, tskip_rotation=1, tskip_context=1,
                                              implicit_rdpcm=1, explicit_rdpcm=1, log2_max_tskip_size=5, tskip_pct=50, sign_data_hiding=1, intra_pct=40)),
    # 12 bit at the top of the QP range: QP 74 / 75 (luma QP 50 / 51 + 24) de-quantise like QP 0 in the reference (its rem6[] / div6[] tables end two
    # entries early, hevc_cabac.c:1428-1440) — the chroma QP offset pushes Cr there
    ("high_qp_12bit_444", 48, 176, 406270, dict(n_pictures=2, gop=0, chroma_format_idc=3, bit_depth=12, qp=44, scaling_list=1, chroma_qp_offsets=1, cb_qp_offset=-1,
                                                cr_qp_offset=11, log2_max_tb_size=4, cbf_pct=60, coeff_density=60)),
    ("ccp444_10_lists_intra", 200, 136, 57, dict(n_pictures=2, gop=0, bit_depth=10, chroma_format_idc=3, cross_component_pred=1, scaling_list=2, qp=24)),
]
OH_TUF_SPARSE, OH_NO_COEFF, OH_FLAT_MATRIX = 16, 0xFFFFFFFF, 0xFF
TU_DT = np.dtype([("x", "<u2"), ("y", "<u2"), ("c_idx", "u1"), ("log2", "u1"), ("kind", "u1"), ("flags", "u1"), ("coeff_off", "<u4")])


def sparse_work_lists(case):
    """[(arrays of the work list in sparse form, cur_id)] per picture + the reference decoder's pictures"""
    _, w, h, seed, kw = case
    data, _ = streamgen.write_stream(w, h, seed, levels=1, **kw)
    lev = streamgen.written_levels()
    want = refdec.decode(data)
    out, pos = [], [0]
    L = refdec.hooked_lib()

    def on_picture(f, cur, poc):
        a = F.frame_to_arrays(f)
        sl = (C.c_uint8 * (4 * 6 * 64 + 2 * 6))()
        have_lists = L.ref_hooked_scaling_list(sl)
        tu = a["tu"].view(TU_DT).copy()
        coeffs = a["coeffs"].copy()
        tu_sparse = np.full(len(tu), OH_NO_COEFF, np.uint32)
        words = []
        p = pos[0]
        n_sparse_blocks = 0
        for i in range(len(tu)):
            t = tu[i]
            if t["kind"] == 4:                                   # PCM: not a residual block in the writer's log
                continue
            head, n = int(lev[p]), int(lev[p + 1]) & 0xffffff
            log2, c_idx, tskip, bypass, intra, qp = head & 15, (head >> 4) & 3, (head >> 8) & 1, (head >> 9) & 1, (head >> 10) & 1, head >> 16
            cross, no_residual = (head >> 11) & 1, (head >> 12) & 1
            assert (log2, c_idx) == (int(t["log2"]), int(t["c_idx"])), (i, head, t)
            scale = ((int(lev[p + 1]) >> 24) ^ 0x80) - 0x80
            if "tu_cross" in a and cross and scale:                  # the link the shim recorded = what the writer coded
                link = int(a["tu_cross"][i])
                assert link != OH_NO_COEFF and ((link >> 24) ^ 0x80) - 0x80 == scale and (t["flags"] & 32), (i, hex(link), scale)
            if no_residual:                                           # cross-component block with cbf = 0: a block of zeros, no levels
                assert t["kind"] == 3 and n == 0
                p += 2
                continue
            assert bypass == (t["kind"] == 3) and (tskip == (t["kind"] == 2) or bypass)
            pairs = lev[p + 2:p + 2 + n]
            p += 2 + n
            if bypass:
                continue
            matrix = 3 * (1 - intra) + c_idx if have_lists and not (tskip and log2 > 2) else OH_FLAT_MATRIX     # hevc_cabac.c:1485: skip blocks > 4x4 stay flat
            tu_sparse[i] = len(words)
            words.append(n | qp << 16 | matrix << 24)
            words.extend(int(v) for v in pairs)
            tu["flags"][i] |= OH_TUF_SPARSE
            if kw.get("tskip_rotation") and tskip and log2 == 2 and intra:
                tu["flags"][i] |= 8                              # OH_TUF_ROTATE: the host's swap of a 4x4 intra skip block (hevc_cabac.c:1877-1884) moves to the GPU with the levels
            nn = 1 << (2 * log2)
            coeffs[int(t["coeff_off"]):int(t["coeff_off"]) + nn] = 0
            n_sparse_blocks += 1
        pos[0] = p
        a["tu"] = tu.view(np.uint8)
        a["coeffs"] = coeffs
        a["sparse"] = np.array(words, np.uint32)
        a["tu_sparse"] = tu_sparse
        if have_lists:
            a["scaling"] = np.frombuffer(bytes(sl), np.uint8).copy()
        out.append((a, cur, n_sparse_blocks))
    refdec.record_work_lists(data, on_picture)
    assert pos[0] == len(lev), "the writer's residual blocks and the recorded transform blocks must pair up one to one"
    return out, want


@pytest.mark.skipif(not refdec.have_refdec(), reason="reference tree / oracle/_ref not present")
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_checker_dequantisation_reproduces_the_reference_decoder(case):
    from test_streams import host_pic_array, oracle
    lists, want = sparse_work_lists(case)
    pics = {}
    assert sum(n for _, _, n in lists) > 50
    for k, (a, cur, _) in enumerate(lists):
        ff = F.FrameFromArrays(a)
        f = ff.frame
        for i in [cur] + [f.ref_pics[r] for r in range(F.OH_MAX_REFS) if f.ref_pics[r] >= 0]:
            if i not in pics:
                pics[i] = F.HostPic(f.p)
        assert oracle().oh_or_frame(C.byref(f), host_pic_array(pics)) == 0
        for c in range(3):
            assert np.array_equal(pics[cur].visible(c), want[k][c]), (case[0], "picture", k, "plane", c)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(refdec.HOOKED_LIB) or not os.path.exists(refdec.LIB), reason="oracle/_ref did not travel")
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_engine_dequantisation_reproduces_the_reference_decoder(case):
    from openhevc_amd.engine import Engine, remap_frame
    lists, want = sparse_work_lists(case)
    eng = Engine(0)
    ids = {}
    for k, (a, cur, _) in enumerate(lists):
        ff = F.FrameFromArrays(a)
        f = ff.frame
        for i in [cur] + [f.ref_pics[r] for r in range(F.OH_MAX_REFS) if f.ref_pics[r] >= 0]:
            if i not in ids:
                ids[i] = eng.pic_alloc(f.p)
        eng.frame_submit(remap_frame(f, ids))
        got = eng.pic_download(ids[cur], f.p)
        for c in range(3):
            d = np.argwhere(got.visible(c) != want[k][c])
            assert len(d) == 0, (case[0], "picture", k, "plane", c, "samples", len(d), "first", d[:4].tolist(), "box", d.min(0).tolist(), d.max(0).tolist())
    eng.close()
