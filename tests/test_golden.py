"""Pins the oracle against the committed fixtures in tests/golden/ (reference output recorded by
tests/golden/make_golden.py).  Runs everywhere — no reference tree, no GPU."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from openhevc_amd import frame as F
from oracle_lib import host_pic_array, i16p, off_u8p, oracle, u8p

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("bd", [8, 10])
def test_slot_fixtures(bd):
    g = np.load(os.path.join(GOLD, f"slots_{bd}bit.npz"))
    o = oracle()
    bpp = 1 if bd == 8 else 2
    for log2 in (2, 3, 4, 5):
        cin, cout = g[f"idct{log2}_in"], g[f"idct{log2}_out"]
        for k in range(len(cin)):
            c = cin[k].copy()
            o.oh_or_idct(bd, i16p(c), log2)
            assert np.array_equal(c, cout[k])
    for k in range(len(g["dst4_in"])):
        c = g["dst4_in"][k].copy()
        o.oh_or_idct_4x4_luma(bd, i16p(c))
        assert np.array_equal(c, g["dst4_out"][k])
    for t, taps in (("qpel", 8), ("epel", 4)):
        src, src2 = np.ascontiguousarray(g[f"{t}_src"]), np.ascontiguousarray(g[f"{t}_src2"])
        sp = off_u8p(src, 8 * src.strides[0] + 8 * bpp)
        for prm, want in zip(g[f"{t}_params"], g[f"{t}_out"]):
            fx, fy, variant, denom, wx0, wx1, ox0, ox1 = (int(v) for v in prm)
            if variant == 0:
                dst = np.zeros((12, 64), np.int16)
                o.oh_or_mc_put(bd, taps, i16p(dst), 64, sp, src.strides[0], 12, fx, fy, 16)
                got = dst[:, :16]
            else:
                dst = np.zeros((12, 16), src.dtype)
                if variant == 1:
                    o.oh_or_mc_uni(bd, taps, u8p(dst), dst.strides[0], sp, src.strides[0], 12, fx, fy, 16)
                elif variant == 2:
                    o.oh_or_mc_bi(bd, taps, u8p(dst), dst.strides[0], sp, src.strides[0], i16p(src2), 64, 12, fx, fy, 16)
                elif variant == 3:
                    o.oh_or_mc_uni_w(bd, taps, u8p(dst), dst.strides[0], sp, src.strides[0], 12, denom, wx0, ox0, fx, fy, 16)
                else:
                    o.oh_or_mc_bi_w(bd, taps, u8p(dst), dst.strides[0], sp, src.strides[0], i16p(src2), 64, 12,
                                    denom, wx0, wx1, ox0, ox1, fx, fy, 16)
                got = dst
            assert np.array_equal(got.astype(np.int32), want), (t, prm)
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        top, left = np.ascontiguousarray(g[f"pred{log2}_top"]), np.ascontiguousarray(g[f"pred{log2}_left"])
        tp, lp = off_u8p(top, 4 * bpp), off_u8p(left, 4 * bpp)
        k = 0
        for c_idx in (0, 1):
            for mode in range(35):
                dst = np.zeros((n, n), top.dtype)
                if mode == 0:
                    o.oh_or_pred_planar(bd, u8p(dst), tp, lp, n, log2)
                elif mode == 1:
                    o.oh_or_pred_dc(bd, u8p(dst), tp, lp, n, log2, c_idx)
                else:
                    o.oh_or_pred_angular(bd, u8p(dst), tp, lp, n, log2, c_idx, mode)
                assert np.array_equal(dst, g[f"pred{log2}_out"][k]), (log2, c_idx, mode)
                k += 1


def md5_planes(hp):
    return [hashlib.md5(np.ascontiguousarray(hp.visible(c)).tobytes()).hexdigest() for c in range(len(hp.planes))]


def load_picture_cases():
    with open(os.path.join(GOLD, "pictures.json")) as fh:
        return json.load(fh)


def build_case(case):
    """(frame, pics, recorder) for one golden picture case — shared with the GPU parity tests"""
    name, w, h, bd, chroma, lc, st, seed, knobs = case
    pcm, cip = "pcm" in name, "cip" in name
    p = F.pic_params(w, h, bit_depth=bd, chroma_format_idc=chroma, log2_ctb_size=lc,
                     pcm_loop_filter_disable=int(pcm), transquant_bypass_enable=int(pcm), constrained_intra_pred=int(cip))
    rec = F.Recorder(p)
    f = rec.synth(F.synth_params(st, seed, **knobs), 2, [0, 1])
    rng = np.random.default_rng(seed)
    pics = {0: F.HostPic(p, rng=rng), 1: F.HostPic(p, rng=rng), 2: F.HostPic(p)}
    return f, pics, rec


def _n_picture_cases():
    with open(os.path.join(GOLD, "pictures.json")) as fh:
        return len(json.load(fh)["cases"])


@pytest.mark.parametrize("idx", range(_n_picture_cases()))        # every committed case, however many make_golden.py wrote
def test_picture_fixtures(idx):
    """whole pipeline (passes 1-5) of the oracle reproduces the recorded MD5s; the stored final
    MD5 was produced by the reference's own in-loop filter driver"""
    gold = load_picture_cases()
    case = gold["cases"][idx]
    want = gold["expected"][case[0]]
    f, pics, rec = build_case(case)
    assert [int(f.n_pu), int(f.n_tu), int(f.n_intra), int(f.n_levels), int(f.n_coeff)] == want["counts"], \
        "synthetic generator drifted: regenerate tests/golden with make_golden.py"
    arr = host_pic_array(pics)
    assert oracle().oh_or_frame(C.byref(f), arr) == 0
    assert md5_planes(pics[2]) == want["final"]
    rec.close()


def load_upsample_cases():
    with open(os.path.join(GOLD, "upsample.json")) as fh:
        return json.load(fh)


def upsample_inputs(case):
    """(parameters, base-layer picture, EL picture parameters) of one golden up-sampling case"""
    name, bl_size, el_size, win, pa, seed = case
    u = F.upsample_setup(bl_size[0], bl_size[1], el_size[0], el_size[1], tuple(win), pa)
    bl = F.HostPic(F.pic_params(*bl_size), rng=np.random.default_rng(seed))
    return u, bl, F.pic_params(*el_size)


def test_oracle_upsample_matches_reference_md5s():
    """SHVC up-sampling: the oracle's whole-picture routine reproduces the MD5s recorded from
    the reference's upsample_base_layer_frame slot (tests/golden/make_golden.py)"""
    import ctypes as C
    from oracle_lib import OhHostPicC, oracle
    gold = load_upsample_cases()
    for case in gold["cases"]:
        u, bl, pe = upsample_inputs(case)
        el = F.HostPic(pe, fill=0)
        hps = []
        for hp_, p_ in ((bl, bl.params), (el, pe)):
            h = OhHostPicC()
            for c, pl in enumerate(hp_.planes):
                w, hh = F.plane_dims(p_, c)
                h.data[c], h.stride[c], h.width[c], h.height[c] = pl.ctypes.data, pl.strides[0], w, hh
            h.bit_depth = 8
            hps.append(h)
        assert oracle().oh_or_upsample_frame(C.byref(hps[0]), C.byref(hps[1]), C.byref(u)) == 0
        assert md5_planes(el) == gold["expected"][case[0]], case[0]
