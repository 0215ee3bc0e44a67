"""Cross-component prediction (4:4:4 range extension; OH_TUF_CROSS): chroma residual += (res_scale_val * luma residual) >> 3.
PARITY UNPINNED against the reference for this one statement: hevc.c:1319-1331 / 1352-1364 and hevc_cabac.c:1942-1947 are host
code outside the files compiled into oracle/_ref.  Checked here: the oracle's pool arithmetic against numpy, and the recorder's
bookkeeping (flags, links, the zero block that carries the prediction when cbf is 0)."""
import ctypes as C

import numpy as np

from openhevc_amd import frame as F
from oracle_lib import host_pic_array, oracle


def residual_pool(f, p):
    co = np.ctypeslib.as_array(f.coeffs, shape=(int(f.n_coeff),)).copy()
    pics = {f.cur_pic: F.HostPic(p)}
    assert oracle().oh_or_pass_residual(C.byref(f), host_pic_array(pics), co.ctypes.data_as(C.POINTER(C.c_int16))) == 0
    return co


def test_cross_component_residual_arithmetic():
    p = F.pic_params(200, 136, bit_depth=10, chroma_format_idc=3, log2_ctb_size=5)
    rec = F.Recorder(p)
    f = rec.synth(F.synth_params(2, 77, ccp_pct=70, intra_pct=30, tskip_pct=20), 2, [0, 1])
    assert bool(f.tu_cross)
    with_cross = residual_pool(f, p)
    g = F.OhFrame()
    C.memmove(C.byref(g), C.byref(f), C.sizeof(F.OhFrame))
    g.tu_cross = None                                        # same list, links dropped: plain inverse transforms
    plain = residual_pool(g, p)
    n_cross = n_zero = 0
    for i in range(f.n_tu):
        t = f.tu[i]
        n2 = 1 << (2 * t.log2_size)
        c = slice(t.coeff_off, t.coeff_off + n2)
        if not t.flags & 32:
            assert f.tu_cross[i] == 0xffffffff
            assert np.array_equal(with_cross[c], plain[c])
            continue
        ty = f.tu[f.tu_cross[i] & 0xffffff]
        scale = int(np.int8(np.uint8(f.tu_cross[i] >> 24)))
        assert t.c_idx in (1, 2) and ty.c_idx == 0 and ty.log2_size == t.log2_size and ty.flags & 64 and abs(scale) in (1, 2, 4, 8)
        y = plain[ty.coeff_off:ty.coeff_off + n2].astype(np.int64)
        want = (plain[c].astype(np.int64) + ((scale * y) >> 3)).astype(np.int16)
        assert np.array_equal(with_cross[c], want), i
        n_cross += 1
        n_zero += int(t.kind == F.TU_BYPASS and not np.any(np.ctypeslib.as_array(f.coeffs, shape=(int(f.n_coeff),))[c]))
    assert n_cross > 20 and n_zero > 0                       # both forms occur: coded chroma block, and cbf 0
    rec.close()
