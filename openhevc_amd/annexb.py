"""ctypes view of include/ohevc_annexb.h (libohevc_host.so): access-unit splitter, NAL scan, NAL unescape, picture-hash SEI."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
END_NOT_FOUND = -100


class OhAuScanner(C.Structure):
    _fields_ = [("state64", C.c_uint64), ("frame_start_found", C.c_int32), ("reserved", C.c_int32)]


class OhNal(C.Structure):
    _fields_ = [("offset", C.c_size_t), ("size", C.c_size_t), ("type", C.c_int32), ("layer_id", C.c_int32), ("temporal_id", C.c_int32),
                ("first_slice_segment_in_pic", C.c_int32)]


class OhPictureHash(C.Structure):
    _fields_ = [("present", C.c_int32), ("hash_type", C.c_int32), ("md5", (C.c_uint8 * 16) * 3), ("crc", C.c_uint32 * 3), ("checksum", C.c_uint32 * 3)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libohevc_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C openhevc_amd` (or __graft_entry__.build())")
        L = C.CDLL(path)
        L.oh_au_scanner_init.argtypes = [C.POINTER(OhAuScanner)]
        L.oh_au_scanner_init.restype = None
        L.oh_au_find_frame_end.argtypes = [C.POINTER(OhAuScanner), C.c_char_p, C.c_size_t]
        L.oh_au_find_frame_end.restype = C.c_long
        L.oh_annexb_split.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_size_t]
        L.oh_annexb_split.restype = C.c_long
        L.oh_annexb_nal_units.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(OhNal), C.c_size_t]
        L.oh_annexb_nal_units.restype = C.c_long
        L.oh_nal_unescape.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_int32)]
        L.oh_nal_unescape.restype = C.c_long
        L.oh_sei_picture_hash.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(OhPictureHash)]
        L.oh_sei_picture_hash.restype = C.c_int
        _lib = L
    return _lib


def split(data):
    """[(start, end)] of the access units of an Annex-B buffer (oh_annexb_split)"""
    L = lib()
    cap = 64
    while True:
        off = (C.c_size_t * cap)()
        n = L.oh_annexb_split(data, len(data), off, cap)
        if n >= 0:
            return [(off[i], off[i + 1]) for i in range(n)]
        cap = -n + 1


def nal_units(buf):
    """[(offset, size, type, layer_id, temporal_id, first_slice_segment_in_pic)]"""
    L = lib()
    cap = 64
    while True:
        arr = (OhNal * cap)()
        n = L.oh_annexb_nal_units(buf, len(buf), arr, cap)
        if n < 0:
            raise ValueError("no start code where one is due")
        if n <= cap:
            return [(u.offset, u.size, u.type, u.layer_id, u.temporal_id, u.first_slice_segment_in_pic) for u in arr[:n]]
        cap = n


def unescape(nal):
    """(rbsp bytes, [positions of the bytes before each dropped emulation-prevention byte], bytes consumed)"""
    L = lib()
    dst = C.create_string_buffer(max(len(nal), 1))
    n = C.c_size_t()
    ns = C.c_int32()
    cap = len(nal) // 3 + 1
    pos = (C.c_int32 * cap)()
    used = L.oh_nal_unescape(nal, len(nal), dst, C.byref(n), pos, cap, C.byref(ns))
    return dst.raw[:n.value], list(pos[:ns.value]), used


def picture_hash(nal):
    """None, or (hash_type, [three digests / values]) of the decoded-picture-hash message of one SEI NAL unit"""
    h = OhPictureHash()
    r = lib().oh_sei_picture_hash(nal, len(nal), C.byref(h))
    if r < 0:
        raise ValueError("malformed SEI NAL unit")
    if not r:
        return None
    if h.hash_type == 0:
        return 0, [bytes(h.md5[c]) for c in range(3)]
    return h.hash_type, list(h.crc if h.hash_type == 1 else h.checksum)
