"""Sparse residual hand-off (OH_TUF_SPARSE, include/ohevc_frame.h): the host passes the quantised
levels residual_coding parsed, de-quantisation (hevc_cabac.c:1478-1494, 1818-1841) moves behind the boundary.
PARITY UNPINNED against the reference for the de-quantisation itself: its statements sit inside
ff_hevc_hls_residual_coding between CABAC reads and cannot be driven from a harness.  What is checked here: the oracle's
restatement against an independent numpy restatement of the same text, saturation, scaling-list indexing per size."""
import ctypes as C

import numpy as np
import pytest

from openhevc_amd import frame as F
from oracle_lib import host_pic_array, oracle

LEVEL_SCALE = [40, 45, 51, 57, 64, 72]          # hevc_cabac.c:1417


def dequant_numpy(f, p):
    """dense coefficient pool with every sparse block de-quantised in numpy"""
    sl = np.ctypeslib.as_array(C.cast(f.scaling, C.POINTER(C.c_uint8)), shape=(4 * 6 * 64 + 12,)) if f.scaling else None
    co = np.ctypeslib.as_array(f.coeffs, shape=(int(f.n_coeff),)).copy()
    for i in range(f.n_tu):
        t = f.tu[i]
        if not t.flags & 16:
            continue
        so = f.tu_sparse[i]
        w0 = f.sparse[so]
        cnt, qp, mid = w0 & 0xffff, (w0 >> 16) & 0xff, w0 >> 24
        l2, n = t.log2_size, 1 << t.log2_size
        shift = p.bit_depth + l2 - 5
        q6 = qp if qp < 74 else 0                          # the reference's rem6[] / div6[] tables end two entries early (hevc_cabac.c:1428-1440)
        add, scale = 1 << (shift - 1), LEVEL_SCALE[q6 % 6] << (q6 // 6)
        blk = np.zeros(n * n, np.int64)
        for k in range(cnt):
            w = f.sparse[so + 1 + k]
            pos, lvl = w & 0xffff, int(np.int16(np.uint16(w >> 16)))
            px, py = pos % n, pos // n
            sm = 16
            if mid != 0xff:
                if px or py or l2 < 4:
                    idx = {2: py * 4 + px, 3: py * 8 + px, 4: (py >> 1) * 8 + (px >> 1), 5: (py >> 2) * 8 + (px >> 2)}[l2]
                    sm = int(sl[((l2 - 2) * 6 + mid) * 64 + idx])
                else:
                    sm = int(sl[4 * 6 * 64 + (l2 - 4) * 6 + mid])
            blk[pos] = max(-32768, min(32767, (lvl * scale * sm + add) >> shift))
        co[t.coeff_off:t.coeff_off + n * n] = blk.astype(np.int16)
    return co


@pytest.mark.parametrize("bd,chroma,lists", [(8, 1, 0), (10, 1, 1), (12, 3, 1), (8, 2, 1)])
def test_oracle_dequant_matches_numpy_restatement(bd, chroma, lists):
    p = F.pic_params(264, 200, bit_depth=bd, chroma_format_idc=chroma)
    rec = F.Recorder(p)
    f = rec.synth(F.synth_params(2, 42 + bd, sparse_pct=70, scaling_list=lists, tskip_pct=20, qp_var=20), 2, [0, 1])
    assert f.n_sparse > 0 and bool(f.scaling) == bool(lists)
    rng = np.random.default_rng(0)
    pics = {0: F.HostPic(p, rng=rng), 1: F.HostPic(p, rng=rng), 2: F.HostPic(p, rng=rng)}
    a = {k: v.copy() for k, v in pics.items()}
    assert oracle().oh_or_frame(C.byref(f), host_pic_array(a)) == 0
    co = dequant_numpy(f, p)
    g = F.OhFrame()
    C.memmove(C.byref(g), C.byref(f), C.sizeof(F.OhFrame))
    tu2 = (F.OhTu * f.n_tu)(*[f.tu[i] for i in range(f.n_tu)])
    for i in range(f.n_tu):
        tu2[i].flags &= ~16
    g.tu = C.cast(tu2, C.POINTER(F.OhTu))
    g.coeffs = co.ctypes.data_as(C.POINTER(C.c_int16))
    g.tu_sparse, g.sparse, g.n_sparse = None, None, 0
    b = {k: v.copy() for k, v in pics.items()}
    assert oracle().oh_or_frame(C.byref(g), host_pic_array(b)) == 0
    assert a[2].equal(b[2])
    rec.close()


def test_sparse_records_are_compact():
    """the point of the hand-off: a sparse work list carries far fewer bytes than its dense pool"""
    p = F.pic_params(416, 240)
    rec = F.Recorder(p)
    f = rec.synth(F.synth_params(2, 5, sparse_pct=100), 2, [0, 1])
    assert all(f.tu[i].flags & 16 for i in range(f.n_tu))
    assert 4 * f.n_sparse < 2 * int(f.n_coeff) // 4
    rec.close()
