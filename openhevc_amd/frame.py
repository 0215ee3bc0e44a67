"""ctypes mirror of include/ohevc_frame.h / ohevc_recorder.h / ohevc_synth.h and the loader of
libohevc_host.so (recorder + synthetic-stream generator; plain C, no GPU needed)."""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)

OH_MAX_REFS = 16
OH_NO_REF = 0xFF
OH_NO_WP = 0xFFFF
OH_NO_COEFF = 0xFFFFFFFF

TU_IDCT, TU_DST4, TU_SKIP, TU_BYPASS, TU_PCM = range(5)
TUF_ADD_NOW, TUF_RDPCM, TUF_RDPCM_VER, TUF_ROTATE = 1, 2, 4, 8
AV_BOTTOM_LEFT, AV_LEFT, AV_UP_LEFT, AV_UP, AV_UP_RIGHT = 1, 2, 4, 8, 16


class OhPicParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "width", "height", "bit_depth", "chroma_format_idc", "log2_ctb_size", "log2_min_cb_size",
        "log2_min_tb_size", "log2_min_pu_size", "pcm_loop_filter_disable", "transquant_bypass_enable",
        "strong_intra_smoothing", "intra_smoothing_disabled", "cb_qp_offset", "cr_qp_offset",
        "sao_enabled", "deblock_enabled", "constrained_intra_pred")] + [("reserved", C.c_int32 * 3)]


class OhPu(C.Structure):
    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("w", C.c_uint8), ("h", C.c_uint8),
                ("ref", C.c_uint8 * 2), ("mv", (C.c_int16 * 2) * 2), ("wp", C.c_uint16), ("reserved", C.c_uint16)]


class OhWeights(C.Structure):
    _fields_ = [("w", (C.c_int16 * 3) * 2), ("o", (C.c_int16 * 3) * 2), ("log2_denom", C.c_uint8 * 2),
                ("reserved", C.c_uint8 * 2)]


class OhTu(C.Structure):
    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("c_idx", C.c_uint8), ("log2_size", C.c_uint8),
                ("kind", C.c_uint8), ("flags", C.c_uint8), ("coeff_off", C.c_uint32)]


class OhIntra(C.Structure):
    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("c_idx", C.c_uint8), ("log2_size", C.c_uint8),
                ("mode", C.c_uint8), ("avail", C.c_uint8), ("tu", C.c_uint32)]


class OhIntraCtu(C.Structure):
    _fields_ = [("sub_first", C.c_uint32), ("n_sub", C.c_uint16), ("ctu", C.c_uint16)]


class OhDeblockCtb(C.Structure):
    _fields_ = [("beta_offset", C.c_int8), ("tc_offset", C.c_int8)]


class OhSaoCtb(C.Structure):
    _fields_ = [("offset_val", (C.c_int16 * 5) * 3), ("band_position", C.c_uint8 * 3), ("eo_class", C.c_uint8 * 3),
                ("type_idx", C.c_uint8 * 3), ("edge_flags", C.c_uint8)]


class OhFrame(C.Structure):
    _fields_ = [
        ("p", OhPicParams), ("cur_pic", C.c_int32), ("ref_pics", C.c_int32 * OH_MAX_REFS),
        ("n_pu", C.c_uint32), ("pu", C.POINTER(OhPu)),
        ("n_wp", C.c_uint32), ("wp", C.POINTER(OhWeights)),
        ("n_tu", C.c_uint32), ("tu", C.POINTER(OhTu)),
        ("n_coeff", C.c_uint64), ("coeffs", C.POINTER(C.c_int16)),
        ("n_intra", C.c_uint32), ("intra", C.POINTER(OhIntra)),
        ("n_ictu", C.c_uint32), ("ictu", C.POINTER(OhIntraCtu)),
        ("n_sub", C.c_uint32), ("sub_start", C.POINTER(C.c_uint32)),
        ("n_levels", C.c_uint32), ("level_start", C.POINTER(C.c_uint32)),
        ("bs_size", C.c_uint32), ("vertical_bs", C.POINTER(C.c_uint8)), ("horizontal_bs", C.POINTER(C.c_uint8)),
        ("qp_y_tab", C.POINTER(C.c_int8)), ("is_pcm", C.POINTER(C.c_uint8)),
        ("deblock", C.POINTER(OhDeblockCtb)), ("sao", C.POINTER(OhSaoCtb)), ("is_intra", C.POINTER(C.c_uint8)),
        ("n_sparse", C.c_uint32), ("sparse", C.POINTER(C.c_uint32)), ("tu_sparse", C.POINTER(C.c_uint32)),
        ("scaling", C.c_void_p), ("tu_cross", C.POINTER(C.c_uint32)), ("bs_in", C.c_void_p),
        ("sao_pending", C.POINTER(C.c_uint8)), ("flags", C.c_uint32),
    ]


OH_FRAME_PINNED, OH_FRAME_BS_PACKED = 1, 2                  # OhFrame.flags (include/ohevc_frame.h)


def bs_size(p):
    """oh_bs_size() of include/ohevc_frame.h: bytes of one boundary-strength grid (the reference's padded allocation)"""
    hs = 1 if p.chroma_format_idc in (1, 2) else 0
    vs = 1 if p.chroma_format_idc == 1 else 0
    bw, bh = p.width >> 2, p.height >> 2
    return max((bw + 4 * (1 << hs)) * bh, bw * (bh + 4 * (1 << vs))) + 64


class OhMvField(C.Structure):
    _fields_ = [("mv", (C.c_int16 * 2) * 2), ("poc", C.c_int32 * 2), ("pred_flag", C.c_uint32), ("ref_idx", C.c_uint8 * 2),
                ("pad", C.c_uint8 * 2)]


class OhBsInputs(C.Structure):
    _fields_ = [("mvf", C.c_void_p), ("cbf_luma", C.c_void_p), ("call_log2", C.c_void_p), ("ctb_flags", C.c_void_p),
                ("loop_filter_across_tiles", C.c_int32)]


class OhSynthParams(C.Structure):
    _fields_ = [("seed", C.c_uint64)] + [(n, C.c_int32) for n in (
        "slice_type", "n_refs", "intra_pct", "skip_pct", "bi_pct", "frac_mv_pct", "mv_range", "cbf_pct",
        "weighted_pct", "split_pct", "qp_base", "qp_var", "sao_pct", "tskip_pct", "pcm_pct", "bypass_pct",
        "vary_deblock_offsets", "sparse_pct", "scaling_list", "ccp_pct", "bs_from_motion", "n_slices", "tile_cols", "tile_rows", "slice_knobs")]


# OhSynthParams.slice_knobs (include/ohevc_synth.h)
SYNTH_NO_LF_ACROSS_SLICES, SYNTH_NO_LF_ACROSS_TILES, SYNTH_DEBLOCK_OFF_SLICES, SYNTH_SLICE_PER_TILE, SYNTH_SLICE_OFFSETS = 1, 2, 4, 8, 16


class OhCtbMaps(C.Structure):
    """include/ohevc_recorder.h: per-CTB slice / tile maps (raster order)"""
    _fields_ = [("slice_addr", C.POINTER(C.c_int32)), ("filter_slice_edges", C.POINTER(C.c_uint8)), ("deblock_disabled", C.POINTER(C.c_uint8)),
                ("tile_id", C.POINTER(C.c_int32)), ("tiles_enabled", C.c_int32), ("loop_filter_across_tiles", C.c_int32)]


assert C.sizeof(OhPu) == 20 and C.sizeof(OhWeights) == 28 and C.sizeof(OhTu) == 12
assert C.sizeof(OhIntra) == 12 and C.sizeof(OhSaoCtb) == 40 and C.sizeof(OhDeblockCtb) == 2


class OhUpsample(C.Structure):
    """include/ohevc_frame.h: parameters of the SHVC inter-layer up-sampling (UpsamplInf + scaled window)"""
    _fields_ = [(n, C.c_int32) for n in ("add_x_lum", "add_y_lum", "scale_x_lum", "scale_y_lum", "add_x_cr", "add_y_cr",
                                         "scale_x_cr", "scale_y_cr", "idx", "win_left", "win_right", "win_top", "win_bottom")]


OH_UP_DEFAULT, OH_UP_X2, OH_UP_X1_5, OH_UP_SNR = 0, 1, 2, 3


def upsample_setup(width_bl, height_bl, width_el, height_el, win=(0, 0, 0, 0), phase_align_flag=0):
    """oh_upsample_setup (hevc.c:446-501) restated for Python callers; win = (left, right, top, bottom)"""
    u = OhUpsample()
    wl, wr, wt, wb = win
    w_el, h_el = width_el - wl - wr, height_el - wt - wb
    u.win_left, u.win_right, u.win_top, u.win_bottom = wl, wr, wt, wb
    u.scale_x_lum = ((width_bl << 16) + (w_el >> 1)) // w_el
    u.scale_y_lum = ((height_bl << 16) + (h_el >> 1)) // h_el
    ph = phase_align_flag << 1
    u.add_x_lum = ((ph * u.scale_x_lum + 2) >> 2) + (1 << 11)
    u.add_y_lum = ((ph * u.scale_y_lum + 2) >> 2) + (1 << 11)
    u.add_x_cr = (((0 + phase_align_flag) * u.scale_x_lum + 2) >> 2) + (1 << 11)
    u.add_y_cr = (((1 + phase_align_flag) * u.scale_y_lum + 2) >> 2) + (1 << 11)
    u.scale_x_cr, u.scale_y_cr = u.scale_x_lum, u.scale_y_lum
    sx, sy = u.scale_x_lum, u.scale_y_lum
    u.idx = OH_UP_SNR if (sx, sy) == (65536, 65536) else OH_UP_X2 if (sx, sy) == (32768, 32768) else \
        OH_UP_X1_5 if (sx, sy) == (43691, 43691) else OH_UP_DEFAULT
    return u


def pic_params(width, height, bit_depth=8, chroma_format_idc=1, log2_ctb_size=6, log2_min_cb_size=3,
               log2_min_tb_size=2, sao=1, deblock=1, strong_intra_smoothing=1, pcm_loop_filter_disable=0,
               transquant_bypass_enable=0, cb_qp_offset=0, cr_qp_offset=0, intra_smoothing_disabled=0,
               constrained_intra_pred=0):
    p = OhPicParams()
    p.width, p.height, p.bit_depth, p.chroma_format_idc = width, height, bit_depth, chroma_format_idc
    p.log2_ctb_size, p.log2_min_cb_size, p.log2_min_tb_size = log2_ctb_size, log2_min_cb_size, log2_min_tb_size
    p.log2_min_pu_size = log2_min_cb_size - 1
    p.sao_enabled, p.deblock_enabled, p.strong_intra_smoothing = sao, deblock, strong_intra_smoothing
    p.pcm_loop_filter_disable, p.transquant_bypass_enable = pcm_loop_filter_disable, transquant_bypass_enable
    p.cb_qp_offset, p.cr_qp_offset, p.intra_smoothing_disabled = cb_qp_offset, cr_qp_offset, intra_smoothing_disabled
    p.constrained_intra_pred = constrained_intra_pred
    assert width % (1 << log2_min_cb_size) == 0 and height % (1 << log2_min_cb_size) == 0
    return p


def hshift(p, c):
    return int(bool(c) and p.chroma_format_idc in (1, 2))


def vshift(p, c):
    return int(bool(c) and p.chroma_format_idc == 1)


def plane_dims(p, c):
    return p.width >> hshift(p, c), p.height >> vshift(p, c)


def n_planes(p):
    return 3 if p.chroma_format_idc else 1


def half_layout(p):
    """python twin of oh_pic_half_layout(): (half_bytes, [stride in samples], [plane offset in bytes])"""
    bpp = 2 if p.bit_depth > 8 else 1
    total, strides, offs = 0, [], []
    for c in range(n_planes(p)):
        w, h = plane_dims(p, c)
        row = (w * bpp + 255) // 256 * 256
        strides.append(row // bpp)
        offs.append(total)
        total += (row * h + 255) // 256 * 256
    return total, strides, offs


_host = None


def build_host():
    subprocess.check_call(["make", "-s", "-C", PKG_DIR, "libohevc_host.so"])


def host():
    """libohevc_host.so: recorder + synth (host-only C)."""
    global _host
    if _host is None:
        path = os.path.join(PKG_DIR, "libohevc_host.so")
        if not os.path.exists(path) or os.path.exists(os.path.join(PKG_DIR, "Makefile")) and os.access(PKG_DIR, os.W_OK):
            build_host()                     # incremental make; a no-op when up to date
        lib = C.CDLL(path)
        V, I = C.c_void_p, C.c_int
        lib.oh_rec_create.restype = V
        lib.oh_rec_create.argtypes = [C.POINTER(OhPicParams)]
        lib.oh_rec_destroy.argtypes = [V]
        lib.oh_rec_begin.argtypes = [V, I, C.POINTER(C.c_int32), I]
        lib.oh_rec_pu.argtypes = [V] + [I] * 10 + [C.POINTER(OhWeights)]
        lib.oh_rec_tu.argtypes = [V, I, I, I, I, I, I, C.POINTER(C.c_int16)]
        lib.oh_rec_tu.restype = C.c_uint32
        lib.oh_rec_tu_sparse.argtypes = [V, I, I, I, I, I, I, I, I, I, C.POINTER(C.c_uint32)]
        lib.oh_rec_tu_sparse.restype = C.c_uint32
        lib.oh_rec_scaling_list.argtypes = [V]
        lib.oh_rec_scaling_list.restype = C.c_void_p
        lib.oh_rec_tu_cross.argtypes = [V, C.c_uint32, C.c_uint32, I]
        lib.oh_rec_intra.argtypes = [V, I, I, I, I, I, I, C.c_uint32]
        lib.oh_rec_finish.argtypes = [V]
        lib.oh_rec_finish.restype = C.POINTER(OhFrame)
        lib.oh_rec_avail.argtypes = [V, I, I, I, I]
        lib.oh_rec_ctb_maps.argtypes = [V]
        lib.oh_rec_ctb_maps.restype = C.POINTER(OhCtbMaps)
        lib.oh_rec_ctb_maps_in_use.argtypes = [V]
        lib.oh_rec_ctb_maps_in_use.restype = C.POINTER(OhCtbMaps)
        lib.oh_rec_mark_decoded.argtypes = [V, I, I, I, I]
        for n, t in (("oh_rec_vertical_bs", C.c_uint8), ("oh_rec_horizontal_bs", C.c_uint8), ("oh_rec_qp_y_tab", C.c_int8),
                     ("oh_rec_is_pcm", C.c_uint8), ("oh_rec_is_intra", C.c_uint8), ("oh_rec_deblock", OhDeblockCtb), ("oh_rec_sao", OhSaoCtb)):
            getattr(lib, n).argtypes = [V]
            getattr(lib, n).restype = C.POINTER(t)
        lib.oh_synth_defaults.argtypes = [C.POINTER(OhSynthParams), I, C.c_uint64]
        lib.oh_synth_picture.argtypes = [V, C.POINTER(OhSynthParams), I, C.POINTER(C.c_int32), I]
        lib.oh_synth_picture.restype = C.POINTER(OhFrame)
        _host = lib
    return _host


class Recorder:
    """Owns one OhRecorder; `frame` is the last finished OhFrame (valid until the next begin)."""

    def __init__(self, params):
        self.lib = host()
        self.params = params
        self.h = self.lib.oh_rec_create(C.byref(params))
        self.frame = None

    def close(self):
        if self.h:
            self.lib.oh_rec_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def ctb_maps(self):
        """the slice / tile maps of the picture just finished (None: one slice, one tile)"""
        m = self.lib.oh_rec_ctb_maps_in_use(self.h)
        return m.contents if m else None

    def synth(self, sp, cur_pic, ref_pics=()):
        refs = (C.c_int32 * OH_MAX_REFS)(*([int(r) for r in ref_pics] + [-1] * (OH_MAX_REFS - len(ref_pics))))
        f = self.lib.oh_synth_picture(self.h, C.byref(sp), int(cur_pic), refs, len(ref_pics))
        if not f:
            raise RuntimeError("oh_synth_picture failed")
        self.frame = f.contents
        return self.frame


def synth_params(slice_type, seed, **kw):
    sp = OhSynthParams()
    host().oh_synth_defaults(C.byref(sp), slice_type, seed)
    for k, v in kw.items():
        if not hasattr(sp, k):
            raise AttributeError(k)
        setattr(sp, k, v)
    return sp


# ---- host pictures (numpy planes) ----
class HostPic:
    """planes as numpy arrays with a row stride >= width (stride in samples is plane.shape[1])"""

    def __init__(self, params, fill=None, rng=None, pad=0):
        self.params = params
        self.bd = params.bit_depth
        dt = np.uint8 if self.bd == 8 else np.uint16
        self.planes = []
        for c in range(n_planes(params)):
            w, h = plane_dims(params, c)
            st = (w + pad + 63) // 64 * 64
            if rng is not None:
                a = rng.integers(0, 1 << self.bd, size=(h, st)).astype(dt)
            else:
                a = np.full((h, st), (1 << (self.bd - 1)) if fill is None else fill, dt)
            self.planes.append(a)

    def copy(self):
        o = HostPic.__new__(HostPic)
        o.params, o.bd = self.params, self.bd
        o.planes = [p.copy() for p in self.planes]
        return o

    def visible(self, c):
        w, h = plane_dims(self.params, c)
        return self.planes[c][:h, :w]

    def equal(self, other):
        return all(np.array_equal(self.visible(c), other.visible(c)) for c in range(len(self.planes)))


class FrameCopy:
    """Deep copy of a finished work list: every array the OhFrame points at is copied into memory this object owns, so the
    list outlives the recorder's next picture.  `.frame` is an OhFrame over the copies (cur_pic / ref_pics as recorded)."""

    def __init__(self, f, pinned_by=None):
        """pinned_by: the engine library (ctypes handle with oh_host_alloc / oh_host_free): every array is copied into page-locked memory
        from oh_host_alloc and the boundary-strength grids are packed four to the byte (oh_pack_bs), i.e. the list is what a recorder
        that writes into blocks lent by the engine hands over: OhFrame.flags = OH_FRAME_PINNED | OH_FRAME_BS_PACKED, the engine
        copies it to the GPU by DMA from where it lies."""
        p = f.p
        self.keep = []
        self._pinned, self._lib = [], pinned_by
        g = OhFrame()
        C.memmove(C.byref(g), C.byref(f), C.sizeof(OhFrame))

        def dup(ptr, nbytes, ctype, data=None):
            if not ptr or nbytes == 0:
                return C.cast(None, C.POINTER(ctype))
            src = data if data is not None else np.frombuffer(C.string_at(ptr, int(nbytes)), dtype=np.uint8)
            if pinned_by is None:
                buf = src.copy()
                self.keep.append(buf)
                return C.cast(buf.ctypes.data, C.POINTER(ctype))
            mem = pinned_by.oh_host_alloc(C.c_size_t(len(src)))
            if not mem:
                raise MemoryError("oh_host_alloc")
            self._pinned.append((mem, len(src)))
            C.memmove(mem, src.ctypes.data, len(src))
            return C.cast(mem, C.POINTER(ctype))

        def packed(ptr, n):                                    # oh_pack_bs: entry i in bits 2 (i & 3) of byte i >> 2
            a = np.frombuffer(C.string_at(ptr, int(n)), dtype=np.uint8) & 3
            a = np.concatenate([a, np.zeros((-len(a)) % 4, np.uint8)]).reshape(-1, 4)
            return (a[:, 0] | a[:, 1] << 2 | a[:, 2] << 4 | a[:, 3] << 6).astype(np.uint8)

        hs, vs = hshift(p, 1), vshift(p, 1)
        n_ctb = ((p.width + (1 << p.log2_ctb_size) - 1) >> p.log2_ctb_size) * ((p.height + (1 << p.log2_ctb_size) - 1) >> p.log2_ctb_size)
        n_pu = (p.width >> p.log2_min_pu_size) * (p.height >> p.log2_min_pu_size)
        n_tb = (p.width >> p.log2_min_tb_size) * (p.height >> p.log2_min_tb_size)
        n_qp = ((p.width >> p.log2_min_cb_size) + 1) * ((p.height >> p.log2_min_cb_size) + 1)
        g.pu = dup(f.pu, f.n_pu * C.sizeof(OhPu), OhPu)
        g.wp = dup(f.wp, f.n_wp * C.sizeof(OhWeights), OhWeights)
        g.tu = dup(f.tu, f.n_tu * C.sizeof(OhTu), OhTu)
        g.coeffs = dup(f.coeffs, f.n_coeff * 2, C.c_int16)
        g.intra = dup(f.intra, f.n_intra * C.sizeof(OhIntra), OhIntra)
        g.ictu = dup(f.ictu, f.n_ictu * C.sizeof(OhIntraCtu), OhIntraCtu)
        g.sub_start = dup(f.sub_start, (f.n_sub + 1) * 4 if f.n_intra else 0, C.c_uint32)
        g.level_start = dup(f.level_start, (f.n_levels + 1) * 4 if f.n_intra else 0, C.c_uint32)
        if pinned_by is not None and f.vertical_bs and f.horizontal_bs and not (f.flags & OH_FRAME_BS_PACKED):
            g.vertical_bs = dup(f.vertical_bs, f.bs_size, C.c_uint8, packed(f.vertical_bs, f.bs_size))
            g.horizontal_bs = dup(f.horizontal_bs, f.bs_size, C.c_uint8, packed(f.horizontal_bs, f.bs_size))
            g.flags |= OH_FRAME_BS_PACKED
        else:
            nb = (f.bs_size + 3) // 4 if f.flags & OH_FRAME_BS_PACKED else f.bs_size
            g.vertical_bs = dup(f.vertical_bs, nb, C.c_uint8)
            g.horizontal_bs = dup(f.horizontal_bs, nb, C.c_uint8)
        g.qp_y_tab = dup(f.qp_y_tab, n_qp, C.c_int8)
        g.is_pcm = dup(f.is_pcm, n_pu, C.c_uint8)
        g.deblock = dup(f.deblock, n_ctb * C.sizeof(OhDeblockCtb), OhDeblockCtb)
        g.sao = dup(f.sao, n_ctb * C.sizeof(OhSaoCtb), OhSaoCtb)
        g.is_intra = dup(f.is_intra, n_pu, C.c_uint8)
        g.sparse = dup(f.sparse, f.n_sparse * 4, C.c_uint32)
        g.tu_sparse = dup(f.tu_sparse, f.n_tu * 4 if f.sparse else 0, C.c_uint32)
        g.tu_cross = dup(f.tu_cross, f.n_tu * 4, C.c_uint32)
        g.sao_pending = dup(f.sao_pending, n_ctb, C.c_uint8)
        if f.scaling:
            g.scaling = C.cast(dup(f.scaling, 4 * 6 * 64 + 12, C.c_uint8), C.c_void_p).value
        if f.bs_in:
            src = C.cast(f.bs_in, C.POINTER(OhBsInputs)).contents
            bi = OhBsInputs()
            for name, n in (("mvf", n_pu * C.sizeof(OhMvField)), ("cbf_luma", n_tb), ("call_log2", n_tb), ("ctb_flags", n_ctb)):
                setattr(bi, name, C.cast(dup(getattr(src, name), int(n), C.c_uint8), C.c_void_p).value)
            bi.loop_filter_across_tiles = src.loop_filter_across_tiles
            self.keep.append(bi)
            g.bs_in = C.addressof(bi)
        if pinned_by is not None:
            g.flags |= OH_FRAME_PINNED
        self.frame = g
        self.bytes = sum(b.nbytes for b in self.keep if isinstance(b, np.ndarray)) + sum(n for _, n in self._pinned)

    def __del__(self):
        for mem, _ in getattr(self, "_pinned", []):
            try:
                self._lib.oh_host_free(C.c_void_p(mem))
            except Exception:      # noqa: BLE001 - interpreter shutdown
                pass
        self._pinned = []

    def with_ids(self, cur_pic, ref_pics):
        """header copy whose picture ids are replaced (the arrays are shared)"""
        g = OhFrame()
        C.memmove(C.byref(g), C.byref(self.frame), C.sizeof(OhFrame))
        g.cur_pic = int(cur_pic)
        for i in range(OH_MAX_REFS):
            g.ref_pics[i] = int(ref_pics[i]) if i < len(ref_pics) else -1
        return g


# ---- work lists as files (fixtures recorded where the reference decoder is available, replayed where it is not) ----
_FRAME_ARRAYS = (("pu", "n_pu", OhPu), ("wp", "n_wp", OhWeights), ("tu", "n_tu", OhTu), ("intra", "n_intra", OhIntra), ("ictu", "n_ictu", OhIntraCtu))


def frame_to_arrays(f):
    """{name: numpy uint8 array} holding everything an OhFrame points at plus its scalar fields (dense work lists: no sparse records,
    no cross-component links, no bs_in)"""
    assert not f.sparse and not f.bs_in and not f.scaling, "only the dense hand-over is stored (cross-component links included)"
    p = f.p
    n_ctb = ((p.width + (1 << p.log2_ctb_size) - 1) >> p.log2_ctb_size) * ((p.height + (1 << p.log2_ctb_size) - 1) >> p.log2_ctb_size)
    n_pu = (p.width >> p.log2_min_pu_size) * (p.height >> p.log2_min_pu_size)
    n_qp = ((p.width >> p.log2_min_cb_size) + 1) * ((p.height >> p.log2_min_cb_size) + 1)

    def grab(ptr, nbytes):
        return np.frombuffer(C.string_at(ptr, int(nbytes)), dtype=np.uint8).copy() if ptr and nbytes else np.zeros(0, np.uint8)
    out = {"params": np.frombuffer(bytes(p), dtype=np.uint8).copy(),
           "ids": np.array([f.cur_pic] + [f.ref_pics[i] for i in range(OH_MAX_REFS)], np.int32),
           "counts": np.array([f.n_pu, f.n_wp, f.n_tu, f.n_coeff, f.n_intra, f.n_ictu, f.n_sub, f.n_levels, f.bs_size], np.int64)}
    for name, cnt, typ in _FRAME_ARRAYS:
        out[name] = grab(getattr(f, name), getattr(f, cnt) * C.sizeof(typ))
    out["coeffs"] = np.frombuffer(C.string_at(f.coeffs, int(f.n_coeff) * 2), dtype=np.int16).copy() if f.n_coeff else np.zeros(0, np.int16)
    out["sub_start"] = grab(f.sub_start, (f.n_sub + 1) * 4 if f.n_intra else 0)
    out["level_start"] = grab(f.level_start, (f.n_levels + 1) * 4 if f.n_intra else 0)
    out["vertical_bs"] = grab(f.vertical_bs, f.bs_size)
    out["horizontal_bs"] = grab(f.horizontal_bs, f.bs_size)
    out["qp_y_tab"] = grab(f.qp_y_tab, n_qp)
    out["is_pcm"] = grab(f.is_pcm, n_pu)
    out["is_intra"] = grab(f.is_intra, n_pu)
    out["deblock"] = grab(f.deblock, n_ctb * C.sizeof(OhDeblockCtb))
    out["sao"] = grab(f.sao, n_ctb * C.sizeof(OhSaoCtb))
    if f.tu_cross:
        out["tu_cross"] = np.frombuffer(C.string_at(f.tu_cross, int(f.n_tu) * 4), dtype=np.uint32).copy()
    if f.sao_pending:
        out["sao_pending"] = grab(f.sao_pending, n_ctb)
    return out


class FrameFromArrays:
    """an OhFrame over arrays produced by frame_to_arrays (kept alive by this object)"""

    def __init__(self, a):
        self.a = {k: np.ascontiguousarray(v) for k, v in a.items()}
        f = OhFrame()
        C.memmove(C.byref(f.p), self.a["params"].ctypes.data, C.sizeof(OhPicParams))
        ids, cnt = self.a["ids"], self.a["counts"]
        f.cur_pic = int(ids[0])
        for i in range(OH_MAX_REFS):
            f.ref_pics[i] = int(ids[1 + i])
        f.n_pu, f.n_wp, f.n_tu, f.n_coeff, f.n_intra, f.n_ictu, f.n_sub, f.n_levels, f.bs_size = (int(v) for v in cnt)

        def ptr(name, typ):
            v = self.a[name]
            return C.cast(v.ctypes.data, C.POINTER(typ)) if v.size else C.cast(None, C.POINTER(typ))
        for name, _, typ in _FRAME_ARRAYS:
            setattr(f, name, ptr(name, typ))
        f.coeffs = ptr("coeffs", C.c_int16)
        f.sub_start, f.level_start = ptr("sub_start", C.c_uint32), ptr("level_start", C.c_uint32)
        f.vertical_bs, f.horizontal_bs = ptr("vertical_bs", C.c_uint8), ptr("horizontal_bs", C.c_uint8)
        f.qp_y_tab, f.is_pcm, f.is_intra = ptr("qp_y_tab", C.c_int8), ptr("is_pcm", C.c_uint8), ptr("is_intra", C.c_uint8)
        f.deblock, f.sao = ptr("deblock", OhDeblockCtb), ptr("sao", OhSaoCtb)
        if "sparse" in self.a:                             # optional: the sparse hand-over (uint32 records, per-TU offsets, scaling lists)
            f.n_sparse = int(self.a["sparse"].size)
            f.sparse, f.tu_sparse = ptr("sparse", C.c_uint32), ptr("tu_sparse", C.c_uint32)
            assert self.a["tu_sparse"].size == f.n_tu
        if "sao_pending" in self.a:                       # optional: the filter-call order of a tiled picture (OhFrame.sao_pending)
            f.sao_pending = ptr("sao_pending", C.c_uint8)
        if "tu_cross" in self.a:                          # optional: cross-component links (luma TU index | res_scale_val << 24 per TU)
            assert self.a["tu_cross"].size == f.n_tu
            f.tu_cross = ptr("tu_cross", C.c_uint32)
        if "scaling" in self.a:
            assert self.a["scaling"].nbytes == 4 * 6 * 64 + 2 * 6
            f.scaling = self.a["scaling"].ctypes.data
        self.frame = f
