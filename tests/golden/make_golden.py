"""Generates tests/golden/*.npz from the reference's OWN kernels (oracle/_ref, built from
/root/reference by oracle/Makefile).  Run in the container that has the reference tree:

    python tests/golden/make_golden.py

The fixtures are DATA (seeded inputs + the reference's outputs); tests/test_golden.py pins the
oracle against them wherever the reference tree is absent (e.g. the GPU box).
"""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from openhevc_amd import frame as F  # noqa: E402
from oracle_lib import (host_pic_array, i16p, intp, off_u8p, oracle, plane_ptrs, rand_pixels, ref, u8p)  # noqa: E402


def slot_vectors(bd):
    rng = np.random.default_rng(7000 + bd)
    r = ref()
    bpp = 1 if bd == 8 else 2
    out = {}
    # residual
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        cin = np.stack([np.clip(rng.laplace(0, 900, (n, n)) * (rng.random((n, n)) < 0.2), -32768, 32767).astype(np.int16)
                        for _ in range(6)] + [rng.integers(-32768, 32768, (n, n)).astype(np.int16)])
        cout = cin.copy()
        for k in range(len(cout)):
            r.ref_idct(bd, log2, i16p(cout[k]), n)
        out[f"idct{log2}_in"], out[f"idct{log2}_out"] = cin, cout
    cin = rng.integers(-20000, 20000, (8, 4, 4)).astype(np.int16)
    cout = cin.copy()
    for k in range(8):
        r.ref_idct_4x4_luma(bd, i16p(cout[k]))
    out["dst4_in"], out["dst4_out"] = cin, cout
    # interpolation: (taps, variant) on one 16x12 block per fractional position
    for epel in (0, 1):
        nfrac = 8 if epel else 4
        src = rand_pixels(rng, (12 + 16, 16 + 16), bd, extreme=True)
        src2 = rng.integers(-(1 << 13), 1 << 14, size=(12, 64)).astype(np.int16)
        res = []
        params = []
        for fx in range(nfrac):
            for fy in range(nfrac):
                denom, wx0, wx1, ox0, ox1 = (int(v) for v in (rng.integers(0, 8), *rng.integers(-100, 128, 2), *rng.integers(-60, 60, 2)))
                for variant in (0, 1, 2, 3, 4):
                    if variant == 0:
                        dst = np.zeros((12, 64), np.int16)
                        r.ref_mc(bd, epel, 0, C.cast(i16p(dst), C.POINTER(C.c_uint8)), 64, off_u8p(src, 8 * src.strides[0] + 8 * bpp),
                                 src.strides[0], None, 0, 12, 0, 0, 0, 0, 0, fx, fy, 16)
                        res.append(dst[:, :16].astype(np.int32))
                    else:
                        dst = np.zeros((12, 16), src.dtype)
                        r.ref_mc(bd, epel, variant, u8p(dst), dst.strides[0], off_u8p(src, 8 * src.strides[0] + 8 * bpp),
                                 src.strides[0], i16p(src2), 64, 12, denom, wx0, wx1, ox0, ox1, fx, fy, 16)
                        res.append(dst.astype(np.int32))
                    params.append((fx, fy, variant, denom, wx0, wx1, ox0, ox1))
        t = "epel" if epel else "qpel"
        out[f"{t}_src"], out[f"{t}_src2"] = src, src2
        out[f"{t}_params"], out[f"{t}_out"] = np.array(params, np.int32), np.stack(res)
    # intra angular / planar / dc from prepared edges
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        top = rand_pixels(rng, (2 * n + 8,), bd)
        left = rand_pixels(rng, (2 * n + 8,), bd)
        left[3] = top[3]
        res = []
        for c_idx in (0, 1):
            for mode in range(35):
                dst = np.zeros((n, n), top.dtype)
                tp, lp = off_u8p(top, 4 * bpp), off_u8p(left, 4 * bpp)
                if mode == 0:
                    r.ref_pred_planar(bd, log2, u8p(dst), tp, lp, n)
                elif mode == 1:
                    r.ref_pred_dc(bd, log2, u8p(dst), tp, lp, n, c_idx)
                else:
                    r.ref_pred_angular(bd, log2, u8p(dst), tp, lp, n, c_idx, mode)
                res.append(dst)
        out[f"pred{log2}_top"], out[f"pred{log2}_left"], out[f"pred{log2}_out"] = top, left, np.stack(res)
    return out


def md5_planes(hp):
    return [hashlib.md5(np.ascontiguousarray(hp.visible(c)).tobytes()).hexdigest() for c in range(len(hp.planes))]


PICTURE_CASES = [
    # name, w, h, bd, chroma, log2_ctb, slice_type, seed, extra synth knobs
    ("i_8b_420", 416, 240, 8, 1, 6, 0, 11, {}),
    ("b_8b_420", 416, 240, 8, 1, 6, 2, 12, {"weighted_pct": 20}),
    ("p_10b_420", 416, 240, 10, 1, 5, 1, 13, {"tskip_pct": 10}),
    ("b_10b_444", 200, 136, 10, 3, 4, 2, 14, {}),
    ("b_8b_pcm", 264, 200, 8, 1, 6, 2, 15, {"pcm_pct": 10, "bypass_pct": 10, "vary_deblock_offsets": 1}),
    # constrained_intra_pred_flag = 1: the intra stage of this one is produced by the reference's intra_pred slots
    ("b_10b_cip", 264, 200, 10, 1, 6, 2, 16, {"intra_pct": 50}),
    # the boundary strengths of this one come from the reference's ff_hevc_deblocking_boundary_strengths over the generator's maps
    ("b_10b_bs_from_motion", 416, 240, 10, 1, 6, 2, 17, {"bs_from_motion": 1, "intra_pct": 20}),
    # several slices / tiles: the reference side works from the raw CTB maps (tab_slice_address, filter_slice_edges, tile ids) and derives
    # the SAO restore flags, the gated boundary strengths and the neighbour availability itself (oracle/ref_harness.c)
    ("b_8b_slices", 416, 240, 8, 1, 5, 2, 18, {"n_slices": 6, "sao_pct": 80, "intra_pct": 25, "slice_knobs": 1 | 4 | 16}),
    ("i_10b_slices", 264, 200, 10, 1, 5, 0, 19, {"n_slices": 5, "sao_pct": 80, "slice_knobs": 1 | 16}),
    ("b_8b_tiles", 416, 240, 8, 1, 5, 2, 20, {"tile_cols": 3, "tile_rows": 2, "sao_pct": 80, "intra_pct": 25, "slice_knobs": 1 | 2 | 8 | 16}),
    ("p_10b_tiles_bs_from_motion", 416, 240, 10, 1, 6, 1, 21, {"tile_cols": 2, "tile_rows": 2, "bs_from_motion": 1, "sao_pct": 80, "slice_knobs": 2}),
]


def picture_case(name, w, h, bd, chroma, lc, st, seed, knobs):
    """Seeded reference pictures and a synthetic work list; the expected picture is REFERENCE OUTPUT end to end:
    passes 1-5 run through the reference's own slots and drivers (oracle/ref_harness.c::ref_frame: put_hevc_{q,e}pel*
    with emulated_edge_mc, idct*/transform_*, put_pcm, hpc.intra_pred[], ff_hevc_hls_filters).  `recon` is the same
    with both in-loop filters switched off."""
    pcm, cip = "pcm" in name, "cip" in name
    p = F.pic_params(w, h, bit_depth=bd, chroma_format_idc=chroma, log2_ctb_size=lc,
                     pcm_loop_filter_disable=int(pcm), transquant_bypass_enable=int(pcm), constrained_intra_pred=int(cip))
    rec = F.Recorder(p)
    f = rec.synth(F.synth_params(st, seed, **knobs), 2, [0, 1])
    rng = np.random.default_rng(seed)
    pics = {0: F.HostPic(p, rng=rng), 1: F.HostPic(p, rng=rng), 2: F.HostPic(p)}
    sys.path.insert(0, os.path.dirname(HERE))
    from test_oracle_picture_vs_ref import ref_frame
    plain = {k: v.copy() for k, v in pics.items()}
    g = F.OhFrame()
    C.memmove(C.byref(g), C.byref(f), C.sizeof(F.OhFrame))
    g.p.deblock_enabled = 0
    g.p.sao_enabled = 0
    assert ref_frame(rec, g, plain) == 0
    assert ref_frame(rec, f, pics) == 0
    return {"recon": md5_planes(plain[2]), "final": md5_planes(pics[2]),
            "counts": [int(f.n_pu), int(f.n_tu), int(f.n_intra), int(f.n_levels), int(f.n_coeff)]}


UPSAMPLE_CASES = [
    # name, BL size, EL size, scaled reference layer window (left, right, top, bottom), phase_align_flag, seed
    ("x2", (208, 120), (416, 240), (0, 0, 0, 0), 0, 31),
    ("x1_5", (176, 96), (264, 144), (0, 0, 0, 0), 0, 32),
    ("snr", (264, 144), (264, 144), (0, 0, 0, 0), 0, 33),
    ("ratio_1_64_window", (200, 112), (328, 200), (8, 4, 4, 8), 0, 34),
    ("x2_phase_align_1080p", (960, 544), (1920, 1088), (0, 0, 0, 0), 1, 35),
]


def upsample_case(name, bl_size, el_size, win, pa, seed):
    """MD5s of the enhancement-layer planes the reference's upsample_base_layer_frame slot produces"""
    (wb, hb), (we, he) = bl_size, el_size
    u = F.upsample_setup(wb, hb, we, he, win, pa)
    bl = F.HostPic(F.pic_params(wb, hb), rng=np.random.default_rng(seed))
    el = F.HostPic(F.pic_params(we, he), fill=0)
    el_p = (C.c_void_p * 3)(*[pl.ctypes.data for pl in el.planes])
    bl_p = (C.c_void_p * 3)(*[pl.ctypes.data for pl in bl.planes])
    el_s = (C.c_int * 3)(*[pl.strides[0] for pl in el.planes])
    bl_s = (C.c_int * 3)(*[pl.strides[0] for pl in bl.planes])
    ref().ref_up_frame(el_p, el_s, we, he, bl_p, bl_s, wb, hb, C.byref(u))
    return md5_planes(el)


def main():
    import json
    with open(os.path.join(HERE, "upsample.json"), "w") as fh:
        json.dump({"cases": [list(c) for c in UPSAMPLE_CASES], "expected": {c[0]: upsample_case(*c) for c in UPSAMPLE_CASES}}, fh, indent=1)
    if "--pictures-only" not in sys.argv:            # the slot vectors only change with the slot set
        for bd in (8, 10):
            np.savez_compressed(os.path.join(HERE, f"slots_{bd}bit.npz"), **slot_vectors(bd))
    import json
    pics = {c[0]: picture_case(*c) for c in PICTURE_CASES}
    with open(os.path.join(HERE, "pictures.json"), "w") as fh:
        json.dump({"cases": [list(c[:8]) + [c[8]] for c in PICTURE_CASES], "expected": pics}, fh, indent=1)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
