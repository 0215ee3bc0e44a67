"""Picture-level pinning of the oracle against the reference's own drivers (container only):

* intra: s->hpc.intra_pred[] (hevcpred_template.c:30-344) with the reference's availability
  derivation, block after block in decode order, vs the oracle's level-ordered pass 3 working
  from the recorder's resolved flags;
* in-loop filters: ff_hevc_hls_filters / ff_hevc_hls_filter (hevc_filter.c:1027-1064) driving
  deblocking_filter_CTB and sao_filter_CTB CTB by CTB, vs the oracle's whole-picture V / H / SAO
  passes.
"""
import ctypes as C

import numpy as np
import pytest

from openhevc_amd import frame as F
from oracle_lib import have_ref, host_pic_array, oracle, plane_ptrs, ref

pytestmark = pytest.mark.skipif(not have_ref(), reason="reference tree / oracle/_ref not present")

CASES = [
    # w, h, bd, chroma, log2_ctb
    (128, 72, 8, 1, 6),
    (200, 136, 8, 1, 5),
    (832, 480, 8, 1, 6),
    (264, 200, 10, 1, 6),
    (136, 88, 8, 3, 6),
    (136, 88, 10, 3, 4),
    (96, 64, 12, 1, 5),
    (64, 64, 8, 0, 6),
    (136, 88, 8, 2, 6),          # 4:2:2
    (200, 136, 10, 2, 5),
    # 16x16 CTBs with subsampled chroma: the reference's SAO sees its right neighbour's first chroma column before the
    # horizontal-edge deblocking reached it (driver order, see oracle.c: g_pre_h); one and two CTB rows are special cases
    (200, 136, 8, 1, 4),
    (136, 88, 10, 2, 4),
    (96, 16, 8, 1, 4),
    (96, 32, 10, 1, 4),
    (96, 40, 8, 2, 4),
]


def residuals_of(frame):
    """inverse-transformed residual pool (what pass 2 leaves behind), via the oracle"""
    n = int(frame.n_coeff)
    res = np.ctypeslib.as_array(frame.coeffs, shape=(max(n, 1),)).copy()
    dummy = {frame.cur_pic: F.HostPic(frame.p)}
    arr = host_pic_array(dummy)
    assert oracle().oh_or_pass_residual(C.byref(frame), arr, res.ctypes.data_as(C.POINTER(C.c_int16))) == 0
    return res


@pytest.mark.parametrize("w,h,bd,chroma,lc", CASES)
@pytest.mark.parametrize("strong", [0, 1])
def test_intra_picture(w, h, bd, chroma, lc, strong):
    p = F.pic_params(w, h, bit_depth=bd, chroma_format_idc=chroma, log2_ctb_size=lc, strong_intra_smoothing=strong)
    rec = F.Recorder(p)
    for seed in range(3):
        sp = F.synth_params(0, 1000 + seed, split_pct=40 + 15 * seed, cbf_pct=50)
        f = rec.synth(sp, 0)
        assert f.n_intra > 0 and f.n_levels > 0
        res = residuals_of(f)
        rng = np.random.default_rng(seed)
        start = F.HostPic(p, rng=rng)             # garbage start: every sample must be overwritten or unread
        a, b = start.copy(), start.copy()
        assert oracle().oh_or_pass_intra(C.byref(f), host_pic_array({0: a}), res.ctypes.data_as(C.POINTER(C.c_int16))) == 0
        # the reference needs the blocks in DECODE order: rebuild that order from the recorder by
        # re-sorting level-sorted items is impossible, so the harness gets a second frame whose
        # intra list is in recording order (levels of one item each => same list)
        d, s = plane_ptrs(b)
        assert ref().ref_intra_picture(C.byref(decode_order(rec, f)), d, s, res.ctypes.data_as(C.POINTER(C.c_int16))) == 0
        for c in range(F.n_planes(p)):
            assert np.array_equal(a.visible(c), b.visible(c)), (seed, c)
        # the smooth-flat case: strong smoothing only triggers on near-flat neighbourhoods
    rec.close()


@pytest.mark.parametrize("w,h,bd,chroma,lc", CASES)
def test_intra_picture_constrained_intra_pred(w, h, bd, chroma, lc):
    """constrained_intra_pred_flag = 1 (hevcpred_template.c:116-163, 185-249): intra blocks of P/B pictures must
    not use samples of inter CUs; the reference reads tab_mvf[].pred_flag, the oracle OhFrame.is_intra.
    Mixed pictures with about half the CUs intra (and PCM CUs, which count as intra) exercise every sweep."""
    if chroma == 0:
        pytest.skip("covered by the 4:2:0 cases (the path does not depend on chroma)")
    p = F.pic_params(w, h, bit_depth=bd, chroma_format_idc=chroma, log2_ctb_size=lc, constrained_intra_pred=1,
                     pcm_loop_filter_disable=1)
    rec = F.Recorder(p)
    n_diff_plain = 0
    for seed in range(4):
        sp = F.synth_params(2, 2000 + seed, intra_pct=(30, 50, 70, 50)[seed], split_pct=35 + 15 * seed, cbf_pct=50,
                            pcm_pct=(0, 0, 10, 10)[seed])
        f = rec.synth(sp, 0, [1, 2])
        assert f.n_intra > 0 and bool(f.is_intra)
        res = residuals_of(f)
        rng = np.random.default_rng(seed)
        start = F.HostPic(p, rng=rng)             # "inter" samples: whatever passes 1-2 left there
        a, b, plain = start.copy(), start.copy(), start.copy()
        assert oracle().oh_or_pass_intra(C.byref(f), host_pic_array({0: a}), res.ctypes.data_as(C.POINTER(C.c_int16))) == 0
        d, s = plane_ptrs(b)
        assert ref().ref_intra_picture(C.byref(decode_order(rec, f)), d, s, res.ctypes.data_as(C.POINTER(C.c_int16))) == 0
        for c in range(F.n_planes(p)):
            assert np.array_equal(a.visible(c), b.visible(c)), (seed, c)
        # the flag matters: the same list without it predicts from the inter samples and differs
        g = F.OhFrame()
        C.memmove(C.byref(g), C.byref(f), C.sizeof(F.OhFrame))
        g.p.constrained_intra_pred = 0
        assert oracle().oh_or_pass_intra(C.byref(g), host_pic_array({0: plain}), res.ctypes.data_as(C.POINTER(C.c_int16))) == 0
        n_diff_plain += int(not plain.equal(a))
    assert n_diff_plain > 0
    rec.close()


def decode_order(rec, f):
    """Copy of OhFrame f whose intra[] is in decode (z-scan) order.  Items of equal level keep
    recording order (stable counting sort), and z-scan order is recoverable as: CTB raster
    order, then Morton order of the block's luma position inside the CTB, then plane."""
    p = f.p
    items = [f.intra[i] for i in range(f.n_intra)]

    def key(it):
        hs, vs = F.hshift(p, it.c_idx), F.vshift(p, it.c_idx)
        x, y = it.x << hs, it.y << vs
        ctb = 1 << p.log2_ctb_size
        cx, cy = x // ctb, y // ctb
        xi, yi = x % ctb, y % ctb
        # 4:2:0 chroma of four 4x4 luma blocks is coded after the 4th luma block (hevc.c:1395)
        if it.c_idx and p.chroma_format_idc == 1 and it.log2_size == 2:
            xi += 4
            yi += 4
        m = 0
        for b in range(6):
            m |= ((xi >> b) & 1) << (2 * b) | ((yi >> b) & 1) << (2 * b + 1)
        return (cy, cx, m, it.c_idx)

    items.sort(key=key)
    arr = (F.OhIntra * len(items))(*items)
    g = F.OhFrame()
    C.memmove(C.byref(g), C.byref(f), C.sizeof(F.OhFrame))
    g.intra = C.cast(arr, C.POINTER(F.OhIntra))
    g._keep = arr
    return g


def smooth_picture(p, rng):
    """low-activity content so that deblocking decisions and SAO categories all occur"""
    hp = F.HostPic(p)
    for c, pl in enumerate(hp.planes):
        h, w = pl.shape
        yy, xx = np.mgrid[0:h, 0:w]
        base = (np.sin(xx / 37.0) + np.cos(yy / 23.0)) * (40 << (p.bit_depth - 8)) + (128 << (p.bit_depth - 8))
        blocks = rng.integers(-6, 7, size=(h // 8 + 1, w // 8 + 1)) * (1 << (p.bit_depth - 8))
        noise = rng.integers(-2, 3, size=(h, w)) * (1 << (p.bit_depth - 8))
        v = base + np.kron(blocks, np.ones((8, 8)))[:h, :w] + noise
        pl[:] = np.clip(v, 0, (1 << p.bit_depth) - 1).astype(pl.dtype)
    return hp


@pytest.mark.parametrize("w,h,bd,chroma,lc", CASES)
@pytest.mark.parametrize("flavour", ["plain", "offsets", "pcm_bypass"])
def test_loop_filters_picture(w, h, bd, chroma, lc, flavour):
    pcm = flavour == "pcm_bypass"
    p = F.pic_params(w, h, bit_depth=bd, chroma_format_idc=chroma, log2_ctb_size=lc,
                     pcm_loop_filter_disable=int(pcm), transquant_bypass_enable=int(pcm),
                     cb_qp_offset=3 if flavour == "offsets" else 0, cr_qp_offset=-4 if flavour == "offsets" else 0)
    rec = F.Recorder(p)
    touched = [False, False]
    for seed in range(3):
        sp = F.synth_params(2 if seed else 0, 2000 + seed, sao_pct=70, qp_base=30 + 4 * seed, qp_var=8,
                            vary_deblock_offsets=int(flavour == "offsets"), pcm_pct=15 if pcm else 0,
                            bypass_pct=15 if pcm else 0)
        f = rec.synth(sp, 2, [0, 1])
        rng = np.random.default_rng(seed)
        start = smooth_picture(p, rng)
        a, b = start.copy(), start.copy()
        arr = host_pic_array({2: a})
        assert oracle().oh_or_pass_deblock(C.byref(f), arr) == 0
        deblocked = a.copy()
        assert oracle().oh_or_pass_sao(C.byref(f), arr) == 0
        scratch = b.copy()
        d, s = plane_ptrs(b)
        d2, _ = plane_ptrs(scratch)
        assert ref().ref_filter_picture(C.byref(f), d, s, d2) == 0
        for c in range(F.n_planes(p)):
            assert np.array_equal(a.visible(c), b.visible(c)), (seed, c)
        touched[0] |= not deblocked.equal(start)
        touched[1] |= not a.equal(deblocked)
    assert touched[0]                              # the deblocking filter really changed samples
    assert touched[1] or w * h <= 64 * 64          # and so did SAO
    rec.close()


def ref_frame(rec, f, pics):
    """whole picture (passes 1-5) through the reference's own kernels: oracle/ref_harness.c::ref_frame.
    pics: {id: HostPic}; pics[f.cur_pic] is reconstructed in place"""
    cur = pics[f.cur_pic]
    d, s = plane_ptrs(cur)
    scratch = cur.copy()
    d2, _ = plane_ptrs(scratch)
    n_refs = max([i for i in range(F.OH_MAX_REFS) if f.ref_pics[i] >= 0] + [-1]) + 1
    RefT = (C.c_void_p * 3) * max(n_refs, 1)
    refs = RefT()
    rs = None
    for i in range(n_refs):
        hp = pics[f.ref_pics[i]]
        for c, pl in enumerate(hp.planes):
            refs[i][c] = pl.ctypes.data
        rs = plane_ptrs(hp)[1]
    if rs is None:
        rs = s
    with ctb_maps_of(rec):
        return ref().ref_frame(C.byref(f), d, s, C.cast(refs, C.c_void_p), n_refs, rs, d2)      # sorts the blocks into decode order itself


class ctb_maps_of:
    """the reference harness works on the slices / tiles of the picture `rec` just finished (none: one slice, one tile)"""

    def __init__(self, rec):
        self.m = rec.ctb_maps()

    def __enter__(self):
        ref().ref_set_ctb_maps(C.byref(self.m) if self.m is not None else None)

    def __exit__(self, *a):
        ref().ref_set_ctb_maps(None)


PIPELINE_CASES = [
    # name, w, h, bd, chroma, log2_ctb, slice_type, params, knobs
    ("b8", 416, 240, 8, 1, 6, 2, {}, {"weighted_pct": 30}),
    ("p10_ctb32", 264, 200, 10, 1, 5, 1, {}, {"tskip_pct": 20}),
    ("b8_far_mv", 200, 136, 8, 1, 6, 2, {}, {"mv_range": 3000, "skip_pct": 60}),
    ("b8_pcm_bypass", 264, 200, 8, 1, 6, 2, {"pcm_loop_filter_disable": 1, "transquant_bypass_enable": 1},
     {"pcm_pct": 12, "bypass_pct": 12, "intra_pct": 30, "vary_deblock_offsets": 1}),
    ("b10_cip", 200, 136, 10, 1, 6, 2, {"constrained_intra_pred": 1}, {"intra_pct": 50}),
    ("b8_422", 200, 136, 8, 2, 5, 2, {}, {"weighted_pct": 25, "intra_pct": 25}),
    ("b10_444", 136, 88, 10, 3, 5, 2, {}, {}),
    ("b12", 136, 88, 12, 1, 5, 2, {}, {}),
    ("i8", 264, 200, 8, 1, 6, 0, {}, {}),
]


@pytest.mark.parametrize("case", PIPELINE_CASES, ids=[c[0] for c in PIPELINE_CASES])
def test_whole_picture_through_reference_kernels(case):
    """passes 1-5 of a work list through the reference's slots and drivers (ref_frame) vs the oracle's oh_or_frame:
    pins the oracle at picture level including the motion-compensation driver arithmetic (MV split, chroma MV
    derivation, edge emulation = clamping, weights) that the slot-level tests cannot see"""
    name, w, h, bd, chroma, lc, st, pk, knobs = case
    p = F.pic_params(w, h, bit_depth=bd, chroma_format_idc=chroma, log2_ctb_size=lc, **pk)
    rec = F.Recorder(p)
    for seed in range(2):
        f = rec.synth(F.synth_params(st, 7000 + seed, **knobs), 2, [0, 1] if st else [])
        rng = np.random.default_rng(seed)
        pics = {0: F.HostPic(p, rng=rng), 1: F.HostPic(p, rng=rng), 2: F.HostPic(p, rng=rng)}
        want = {k: v.copy() for k, v in pics.items()}
        assert oracle().oh_or_frame(C.byref(f), host_pic_array(want)) == 0
        assert ref_frame(rec, f, pics) == 0
        for c in range(F.n_planes(p)):
            assert np.array_equal(want[2].visible(c), pics[2].visible(c)), (name, seed, c)
    rec.close()


# ---- several slices / tiles -------------------------------------------------------------------------------------------
SLICE_CASES = [
    # name, w, h, bd, chroma, log2_ctb, slice_type, synth knobs
    ("slices_lf_off", 416, 240, 8, 1, 5, 2, dict(n_slices=6, slice_knobs=F.SYNTH_NO_LF_ACROSS_SLICES | F.SYNTH_SLICE_OFFSETS)),
    ("slices_deblock_off", 264, 200, 10, 1, 4, 2, dict(n_slices=9, slice_knobs=F.SYNTH_NO_LF_ACROSS_SLICES | F.SYNTH_DEBLOCK_OFF_SLICES)),
    ("slices_intra", 264, 200, 8, 1, 5, 0, dict(n_slices=5, slice_knobs=F.SYNTH_NO_LF_ACROSS_SLICES | F.SYNTH_SLICE_OFFSETS)),
    ("tiles_lf_off", 416, 240, 8, 1, 5, 2, dict(tile_cols=3, tile_rows=2, slice_knobs=F.SYNTH_NO_LF_ACROSS_TILES)),
    ("tiles_slices", 416, 240, 10, 1, 4, 2, dict(tile_cols=2, tile_rows=3, slice_knobs=F.SYNTH_NO_LF_ACROSS_TILES | F.SYNTH_SLICE_PER_TILE |
                                                 F.SYNTH_NO_LF_ACROSS_SLICES | F.SYNTH_SLICE_OFFSETS)),
    ("tiles_lf_on_444", 200, 136, 8, 3, 5, 0, dict(tile_cols=2, tile_rows=2, slice_knobs=F.SYNTH_SLICE_PER_TILE | F.SYNTH_NO_LF_ACROSS_SLICES)),
    ("slices_bs_from_motion", 416, 240, 8, 1, 5, 2, dict(n_slices=6, bs_from_motion=1, slice_knobs=F.SYNTH_NO_LF_ACROSS_SLICES | F.SYNTH_DEBLOCK_OFF_SLICES)),
    ("tiles_bs_from_motion", 416, 240, 10, 2, 5, 1, dict(tile_cols=3, tile_rows=2, bs_from_motion=1, slice_knobs=F.SYNTH_NO_LF_ACROSS_TILES | F.SYNTH_SLICE_PER_TILE)),
]


@pytest.mark.parametrize("case", SLICE_CASES, ids=[c[0] for c in SLICE_CASES])
def test_slices_and_tiles_through_reference_drivers(case):
    """Pictures of several slices / tiles (hevc.c:2592-2642: neighbour availability; hevc_filter.c:206-252: the SAO restore flags
    -> sao_edge_filter[1]; :819-824, :857-862: boundary strengths gated at slice / tile edges; per-slice deblocking off and
    offsets).  The engine-side inputs — OhSaoCtb.edge_flags, the gated BS grids or OhBsInputs.ctb_flags, the intra candidate
    flags — are derived by the recorder / generator from the CTB maps; the reference side gets the RAW maps
    (tab_slice_address, filter_slice_edges, tile ids) and derives everything itself in sao_filter_CTB,
    ff_hevc_deblocking_boundary_strengths and the harness's hls_decode_neighbour."""
    name, w, h, bd, chroma, lc, st, knobs = case
    p = F.pic_params(w, h, bit_depth=bd, chroma_format_idc=chroma, log2_ctb_size=lc)
    rec = F.Recorder(p)
    any_flags = any_off = False
    for seed in range(3):
        f = rec.synth(F.synth_params(st, 8100 + seed, sao_pct=80, intra_pct=25, **knobs), 2, [0, 1] if st else [])
        m = rec.ctb_maps()
        assert m is not None
        n_ctb = ((w + (1 << lc) - 1) >> lc) * ((h + (1 << lc) - 1) >> lc)
        any_flags |= any(f.sao[i].edge_flags for i in range(n_ctb))
        any_off |= any(m.deblock_disabled[i] for i in range(n_ctb))
        rng = np.random.default_rng(seed)
        pics = {0: smooth_picture(p, rng), 1: smooth_picture(p, rng), 2: F.HostPic(p, rng=rng)}
        want = {k: v.copy() for k, v in pics.items()}
        assert oracle().oh_or_frame(C.byref(f), host_pic_array(want)) == 0
        assert ref_frame(rec, f, pics) == 0
        for c in range(F.n_planes(p)):
            assert np.array_equal(want[2].visible(c), pics[2].visible(c)), (name, seed, c)
    if "lf_off" in name or "tiles_slices" in name:
        assert any_flags, "no CTB edge was unfilterable: the case does not exercise sao_edge_filter[1]"
    if "deblock_off" in name:
        assert any_off
    rec.close()
