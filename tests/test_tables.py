"""The drop-in boundary itself: ff_hevcdsp_init_hip / ff_hevcpred_init_hip / ff_videodsp_init_hip
fill the reference's function-pointer tables with RECORDING slots.  tests/replay_driver.c issues
the slot calls the reference's CTU loop would issue (raw pointers, byte strides, edge emulation
buffers); the re-recorded work list must reconstruct the identical picture."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from openhevc_amd import frame as F
from oracle_lib import host_pic_array, oracle

HERE = os.path.dirname(os.path.abspath(__file__))


def driver():
    F.host()                                                  # builds libohevc_host.so (tables.c inside)
    out = os.path.join(HERE, "_replay_driver.so")
    src = os.path.join(HERE, "replay_driver.c")
    lib = os.path.join(F.PKG_DIR, "libohevc_host.so")
    inc = os.path.join(os.path.dirname(HERE), "oracle", "replay_slots.inc")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(lib), os.path.getmtime(inc)):
        subprocess.check_call(["gcc", "-O1", "-g", "-fPIC", "-shared", "-std=gnu99", "-o", out, src, lib,
                               "-Wl,-rpath," + F.PKG_DIR])
    d = C.CDLL(out)
    d.replay_through_tables.argtypes = [C.POINTER(F.OhFrame), C.c_void_p, C.c_void_p * 3, C.c_int * 3, C.c_void_p, C.c_int, C.c_int * 3]
    return d


def padded_planes(p, rng, pad=96):
    """reference-style frame: planes with an edge border (the reference pads by 32+; MVs resolve
    inside it).  Returns (backing arrays, pointer to sample (0,0) per plane, linesize per plane)"""
    bpp = 2 if p.bit_depth > 8 else 1
    arrs, ptrs, ls = [], [], []
    for c in range(F.n_planes(p)):
        w, h = F.plane_dims(p, c)
        a = rng.integers(0, 1 << p.bit_depth, size=(h + 2 * pad, w + 2 * pad)).astype(np.uint8 if bpp == 1 else np.uint16)
        arrs.append(a)
        ptrs.append(a.ctypes.data + pad * a.strides[0] + pad * bpp)
        ls.append(a.strides[0])
    return arrs, ptrs, ls


def visible(p, arrs, pad=96):
    hp = F.HostPic(p)
    for c, a in enumerate(arrs):
        w, h = F.plane_dims(p, c)
        hp.planes[c][:h, :w] = a[pad:pad + h, pad:pad + w]
    return hp


CASES = [("p8", 8, 1, 1, {}), ("b8", 8, 1, 2, {"weighted_pct": 30}), ("b8_far", 8, 1, 2, {"mv_range": 1200}),
         ("b10", 10, 1, 2, {"tskip_pct": 30, "intra_pct": 30}), ("i8", 8, 1, 0, {}), ("b8_444", 8, 3, 2, {"weighted_pct": 30}),
         ("b8_bypass", 8, 1, 2, {"bypass_pct": 20, "pcm_pct": 10, "intra_pct": 30}),
         ("b8_422", 8, 2, 2, {"weighted_pct": 30, "intra_pct": 25})]


@pytest.mark.parametrize("name,bd,chroma,st,knobs", CASES, ids=[c[0] for c in CASES])
def test_recording_tables_round_trip(name, bd, chroma, st, knobs):
    byp = "bypass" in name
    p = F.pic_params(200, 136, bit_depth=bd, chroma_format_idc=chroma, pcm_loop_filter_disable=int(byp),
                     transquant_bypass_enable=int(byp))
    rec, rec2 = F.Recorder(p), F.Recorder(p)
    f = rec.synth(F.synth_params(st, 99, **knobs), 2, [0, 1])
    rng = np.random.default_rng(3)
    ref_arrs = [padded_planes(p, rng) for _ in range(2)]
    cur_arrs, cur_ptrs, cur_ls = padded_planes(p, rng)
    npl = F.n_planes(p)
    RefT = (C.c_void_p * 3) * 2
    refs = RefT()
    for s in range(2):
        for c in range(npl):
            refs[s][c] = ref_arrs[s][1][c]
    cur = (C.c_void_p * 3)(*(cur_ptrs + [None] * (3 - npl)))
    cls = (C.c_int * 3)(*(cur_ls + [0] * (3 - npl)))
    rls = (C.c_int * 3)(*(ref_arrs[0][2] + [0] * (3 - npl)))
    bad = driver().replay_through_tables(C.byref(f), rec2.h, cur, cls, C.cast(refs, C.c_void_p), 2, rls)
    assert bad == 0, f"{bad} slot calls could not be translated"
    g = F.host().oh_rec_finish(rec2.h).contents
    assert (g.n_pu, g.n_tu, g.n_intra, int(g.n_coeff)) == (f.n_pu, f.n_tu, f.n_intra, int(f.n_coeff))
    # both work lists, run by the checker on the same pictures, give the same result
    base = {0: visible(p, ref_arrs[0][0]), 1: visible(p, ref_arrs[1][0])}
    a = dict(base); a[2] = F.HostPic(p, rng=np.random.default_rng(8))
    b = dict(base); b[2] = a[2].copy()
    assert oracle().oh_or_frame(C.byref(f), host_pic_array(a)) == 0
    assert oracle().oh_or_frame(C.byref(g), host_pic_array(b)) == 0
    assert a[2].equal(b[2])
    rec.close(); rec2.close()
