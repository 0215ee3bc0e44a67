"""ctypes views of the two CHECKERS (test infrastructure, never the product):

* ``oracle/liboracle.so``          our CPU restatement (oracle/oracle.c)
* ``oracle/_ref/libohevc_ref.so``  the reference's own C kernels compiled from /root/reference
                                   (only exists where that tree is present)
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_TREE = "/root/reference"

c_u8p = C.POINTER(C.c_uint8)
c_i16p = C.POINTER(C.c_int16)
c_intp = C.POINTER(C.c_int)


def u8p(a):
    return a.ctypes.data_as(c_u8p)


def i16p(a):
    return a.ctypes.data_as(c_i16p)


def intp(a):
    return a.ctypes.data_as(c_intp)


def off_u8p(a, byte_off):
    """pointer `byte_off` bytes into array a"""
    return C.cast(a.ctypes.data + int(byte_off), c_u8p)


def off_i16p(a, el_off):
    return C.cast(a.ctypes.data + 2 * int(el_off), c_i16p)


_oracle = None
_ref = None

I, P, U8, I16, IP = C.c_int, C.c_ssize_t, c_u8p, c_i16p, c_intp
V = C.c_void_p

# ptrdiff_t arguments MUST be declared: an undeclared Python int travels as a 32-bit C int
_ORACLE_SIGS = {
    "oh_or_transform_add": [I, U8, I16, P, I],
    "oh_or_transform_skip": [I, I16, I],
    "oh_or_transform_rdpcm": [I16, I, I],
    "oh_or_idct_4x4_luma": [I, I16],
    "oh_or_idct": [I, I16, I],
    "oh_or_idct_dc": [I, I16, I],
    "oh_or_mc_put": [I, I, I16, P, U8, P, I, I, I, I],
    "oh_or_mc_uni": [I, I, U8, P, U8, P, I, I, I, I],
    "oh_or_mc_bi": [I, I, U8, P, U8, P, I16, P, I, I, I, I],
    "oh_or_mc_uni_w": [I, I, U8, P, U8, P, I, I, I, I, I, I, I],
    "oh_or_mc_bi_w": [I, I, U8, P, U8, P, I16, P, I, I, I, I, I, I, I, I, I],
    "oh_or_pred_planar": [I, U8, U8, U8, P, I],
    "oh_or_pred_dc": [I, U8, U8, U8, P, I, I],
    "oh_or_pred_angular": [I, U8, U8, U8, P, I, I, I],
    "oh_or_intra_pred": [V, U8, P, I, I, I, I, I, I, I, I, V],
    "oh_or_loop_filter_luma": [I, U8, P, P, I, IP, U8, U8],
    "oh_or_loop_filter_chroma": [I, U8, P, P, IP, U8, U8],
    "oh_or_sao_band": [I, U8, U8, P, P, I16, I, I, I],
    "oh_or_sao_edge": [I, U8, U8, P, P, I16, I, IP, I, I, I, U8, U8, U8],
    "oh_or_pass_inter": [V, V], "oh_or_pass_residual": [V, V, I16], "oh_or_pass_intra": [V, V, I16],
    "oh_or_pass_deblock": [V, V], "oh_or_pass_sao": [V, V], "oh_or_frame": [V, V],
    "oh_or_up_luma_h": [I, I, I16, P, U8, P, I, I, I, I, I, V], "oh_or_up_cr_h": [I, I, I16, P, U8, P, I, I, I, I, I, V],
    "oh_or_up_luma_v": [I, I, U8, P, I16, P, I, I, I, I, I, I, I, V], "oh_or_up_cr_v": [I, I, U8, P, I16, P, I, I, I, I, I, I, I, V],
    "oh_or_upsample_frame": [V, V, V],
}
_REF_SIGS = {
    "ref_transform_add": [I, I, U8, I16, P],
    "ref_transform_skip": [I, I16, I],
    "ref_transform_rdpcm": [I, I16, I, I],
    "ref_idct_4x4_luma": [I, I16],
    "ref_idct": [I, I, I16, I],
    "ref_idct_dc": [I, I, I16],
    "ref_mc": [I, I, I, U8, P, U8, P, I16, P, I, I, I, I, I, I, I, I, I],
    "ref_emulated_edge_mc": [I, U8, U8, P, P, I, I, I, I, I, I],
    "ref_pred_planar": [I, I, U8, U8, U8, P],
    "ref_pred_dc": [I, I, U8, U8, U8, P, I],
    "ref_pred_angular": [I, I, U8, U8, U8, P, I, I],
    "ref_loop_filter": [I, I, U8, P, I, IP, U8, U8],
    "ref_sao_band": [I, U8, U8, P, P, I16, I, IP, I, I, I],
    "ref_sao_edge": [I, I, U8, U8, P, P, I16, I, IP, I, I, I, U8, U8, U8],
    "ref_intra_picture": [V, V, V, I16],
    "ref_filter_picture": [V, V, V, V],
    "ref_up_block_h": [I, I, I, I16, P, U8, P, I, I, I, I, I, V],
    "ref_up_block_v": [I, I, I, U8, P, I16, P, I, I, I, I, I, I, I, V],
    "ref_up_frame": [V, V, I, I, V, V, I, I, V],
    "ref_frame": [V, V, V, V, I, V, V],
    "ref_up_blocks": [V, V, I, I, V, V, I, I, V, I],
    "ref_set_ctb_maps": [V],
}


def _declare(lib, sigs):
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int if ("pass_" in name or name.endswith("_frame") or name.endswith("_picture") or name == "ref_up_blocks") and name != "ref_up_frame" else None
    return lib


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])


def oracle():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "liboracle.so")
        build_oracle()                       # incremental: keeps the checker in step with include/*.h
        _oracle = _declare(C.CDLL(path), _ORACLE_SIGS)
    return _oracle


def have_ref():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libohevc_ref.so")) or os.path.isdir(REF_TREE)


def ref():
    """The reference kernels.  Built on demand when /root/reference is present."""
    global _ref
    if _ref is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libohevc_ref.so")
        if os.path.isdir(REF_TREE):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "ref"])      # incremental
        elif not os.path.exists(path):
            raise RuntimeError("reference tree not present; use tests/golden fixtures")
        _ref = _declare(C.CDLL(path), _REF_SIGS)
    return _ref


def pix_dtype(bd):
    return np.uint8 if bd == 8 else np.uint16


def rand_pixels(rng, shape, bd, extreme=False):
    """uniform random samples; `extreme` mixes in runs of 0 / max (saturation cases)"""
    mx = (1 << bd) - 1
    a = rng.integers(0, mx + 1, size=shape, dtype=np.int64)
    if extreme:
        m = rng.integers(0, 4, size=shape)
        a = np.where(m == 0, 0, np.where(m == 1, mx, a))
    return a.astype(pix_dtype(bd))


# ---- picture-level helpers -------------------------------------------------------------------
class OhHostPicC(C.Structure):
    _fields_ = [("data", C.c_void_p * 3), ("stride", C.c_ssize_t * 3), ("width", C.c_int32 * 3),
                ("height", C.c_int32 * 3), ("bit_depth", C.c_int32)]


def host_pic_array(pics):
    """pics: dict picture-id -> openhevc_amd.frame.HostPic.  Returns (ctypes array indexed by id, keepalive)."""
    from openhevc_amd import frame as F
    n = max(pics) + 1
    arr = (OhHostPicC * n)()
    for pid, hp in pics.items():
        for c, pl in enumerate(hp.planes):
            w, h = F.plane_dims(hp.params, c)
            arr[pid].data[c] = pl.ctypes.data
            arr[pid].stride[c] = pl.strides[0]
            arr[pid].width[c], arr[pid].height[c] = w, h
        arr[pid].bit_depth = hp.bd
    return arr


def oracle_frame(frame, pics):
    """run the whole oracle pipeline for one work list; pics[frame.cur_pic] is overwritten"""
    arr = host_pic_array(pics)
    rc = oracle().oh_or_frame(C.byref(frame), arr)
    assert rc == 0
    return pics[frame.cur_pic]


def plane_ptrs(hp):
    d = (C.c_void_p * 3)(*[pl.ctypes.data for pl in hp.planes] + [None] * (3 - len(hp.planes)))
    s = (C.c_ssize_t * 3)(*[pl.strides[0] for pl in hp.planes] + [0] * (3 - len(hp.planes)))
    return d, s
