"""Boundary-strength derivation: the CPU restatement (oracle/oracle.c: oh_or_bs_derive) against the
reference's own ff_hevc_deblocking_boundary_strengths (oracle/_ref, hevc_filter.c:584-941) on the same seeded maps."""
import ctypes as C

import numpy as np
import pytest

from bs_inputs import as_struct, make_inputs
from openhevc_amd import frame as F
from oracle_lib import have_ref, oracle, ref

CASES = [(416, 240, 6, 3, 1), (416, 240, 4, 3, 2), (200, 136, 5, 3, 3), (832, 480, 6, 3, 4), (1920, 1080, 6, 3, 5), (272, 144, 6, 4, 6)]


def derive(lib_fn, p, maps):
    n = F.bs_size(p)
    vbs, hbs = np.full(n, 7, np.uint8), np.full(n, 7, np.uint8)
    st = as_struct(*maps)
    assert lib_fn(C.byref(p), C.byref(st), vbs.ctypes.data_as(C.c_void_p), hbs.ctypes.data_as(C.c_void_p)) == 0
    return vbs, hbs


@pytest.mark.skipif(not have_ref(), reason="reference tree not present")
@pytest.mark.parametrize("w,h,lc,lcb,seed", CASES)
def test_oracle_matches_reference(w, h, lc, lcb, seed):
    p = F.pic_params(w, h, log2_ctb_size=lc, log2_min_cb_size=lcb)
    o, r = oracle().oh_or_bs_derive, ref().ref_bs_derive
    for fn in (o, r):
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        fn.restype = C.c_int
    seen = set()
    for k in range(3):
        maps = make_inputs(p, 1000 * seed + k, intra_pct=(0, 15, 40)[k])
        vo, ho = derive(o, p, maps)
        vr, hr = derive(r, p, maps)
        assert np.array_equal(vo, vr), f"vertical grid differs at {np.nonzero(vo != vr)[0][:8]}"
        assert np.array_equal(ho, hr), f"horizontal grid differs at {np.nonzero(ho != hr)[0][:8]}"
        seen |= set(np.unique(vo)) | set(np.unique(ho))
    assert seen == {0, 1, 2}                                 # every strength occurs
