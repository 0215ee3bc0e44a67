"""ctypes binding of libohevc_hip.so (include/ohevc_hip.h).  Plumbing only: no arithmetic here.

There is no CPU fallback: creating an Engine without the built library or without a GPU raises.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import frame as F

OH_N_PASSES = 6
PASS_NAMES = ("inter", "residual", "intra", "deblock_v", "deblock_h", "sao")

_lib = None


class EngineError(RuntimeError):
    pass


class OhWindow(C.Structure):                                  # include/ohevc_hip.h
    _fields_ = [("left", C.c_int32), ("right", C.c_int32), ("top", C.c_int32), ("bottom", C.c_int32)]


def lib_path():
    return os.path.join(F.PKG_DIR, "libohevc_hip.so")


def build():
    subprocess.check_call(["make", "-s", "-C", F.PKG_DIR, "libohevc_hip.so", "libohevc_host.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(lib_path()):
            raise EngineError("libohevc_hip.so is not built (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(lib_path())
        V, I = C.c_void_p, C.c_int
        PP = C.POINTER(C.c_void_p)
        L.oh_engine_create.argtypes = [PP, I]
        L.oh_engine_create_on_stream.argtypes = [PP, I, V]
        L.oh_pic_bytes.argtypes = [C.POINTER(F.OhPicParams)]
        L.oh_pic_bytes.restype = C.c_size_t
        L.oh_pic_wrap.argtypes = [V, C.POINTER(F.OhPicParams), V, V, C.c_size_t, C.POINTER(I)]
        L.oh_pic_final_half.argtypes = [V, I]
        L.oh_pic_set_final_half.argtypes = [V, I, I]
        L.oh_engine_destroy.argtypes = [V]
        L.oh_engine_destroy.restype = None
        L.oh_engine_last_error.argtypes = [V]
        L.oh_engine_last_error.restype = C.c_char_p
        L.oh_engine_sync.argtypes = [V]
        L.oh_pic_alloc.argtypes = [V, C.POINTER(F.OhPicParams), C.POINTER(I)]
        L.oh_pic_free.argtypes = [V, I]
        L.oh_pic_upload.argtypes = [V, I, C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t)]
        L.oh_pic_download.argtypes = [V, I, C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t)]
        L.oh_pic_download_window.argtypes = [V, I, C.POINTER(OhWindow), C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t)]
        L.oh_pic_download_start.argtypes = [V, I, C.POINTER(OhWindow), C.POINTER(C.c_void_p)]
        L.oh_download_finish.argtypes = [V, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_ssize_t)]
        L.oh_pic_upsample_ctbs.argtypes = [V, I, I, C.c_void_p, I, C.POINTER(C.c_uint32), I]
        L.oh_pics_md5.argtypes = [V, C.POINTER(C.c_int), I, C.POINTER(C.c_uint8)]
        L.oh_frame_upload.argtypes = [V, C.POINTER(F.OhFrame), PP]
        L.oh_frames_upload.argtypes = [V, C.POINTER(C.POINTER(F.OhFrame)), I, PP]
        L.oh_frame_execute.argtypes = [V, V]
        L.oh_pic_upsample.argtypes = [V, C.c_int, C.c_int, V]
        L.oh_frames_execute.argtypes = [V, C.POINTER(C.c_void_p), C.c_int]
        L.oh_frame_free.argtypes = [V, V]
        L.oh_frame_release.argtypes = [V, V]
        L.oh_frame_download_bs.argtypes = [V, V, V, V, C.c_size_t]
        L.oh_frame_submit.argtypes = [V, C.POINTER(F.OhFrame)]
        L.oh_engine_profile.argtypes = [V, I]
        L.oh_engine_pass_times.argtypes = [V, C.POINTER(C.c_double), C.POINTER(C.c_uint64), I]
        L.oh_engine_intra_launch_times.argtypes = [V, C.POINTER(C.c_double), C.POINTER(C.c_uint64), I]
        L.oh_engine_host_times.argtypes = [V, C.POINTER(C.c_double), C.POINTER(C.c_uint64), I, I]
        L.oh_engine_memory.argtypes = [V, C.POINTER(C.c_uint64)]
        L.oh_engine_upload_bytes.argtypes = [V, I]
        L.oh_engine_upload_bytes.restype = C.c_uint64
        L.oh_engine_stream.argtypes = [V]
        L.oh_engine_stream.restype = V
        L.oh_host_alloc.argtypes = [C.c_size_t]
        L.oh_host_alloc.restype = V
        L.oh_host_free.argtypes = [V]
        L.oh_host_free.restype = None
        L.oh_pic_device_planes.argtypes = [V, I, C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        _lib = L
    return _lib


class Engine:
    def __init__(self, device=0, stream=None):
        """stream: optional hipStream_t handle (int), e.g. torch.cuda.current_stream().cuda_stream"""
        self.L = lib()
        h = C.c_void_p()
        if stream is None:
            rc = self.L.oh_engine_create(C.byref(h), device)
        else:
            rc = self.L.oh_engine_create_on_stream(C.byref(h), device, C.c_void_p(stream))
        if rc != 0:
            raise EngineError(f"oh_engine_create(device={device}) failed with {rc}: no usable MI355X / HIP device")
        self.h = h

    def _chk(self, rc, what):
        if rc != 0:
            raise EngineError(f"{what} failed ({rc}): {self.L.oh_engine_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.oh_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        self._chk(self.L.oh_engine_sync(self.h), "oh_engine_sync")

    # ---- pictures ----
    def pic_alloc(self, params):
        pid = C.c_int(-1)
        self._chk(self.L.oh_pic_alloc(self.h, C.byref(params), C.byref(pid)), "oh_pic_alloc")
        return pid.value

    def pic_wrap(self, params, half0_ptr, half1_ptr, half_bytes):
        pid = C.c_int(-1)
        self._chk(self.L.oh_pic_wrap(self.h, C.byref(params), C.c_void_p(half0_ptr), C.c_void_p(half1_ptr), half_bytes,
                                     C.byref(pid)), "oh_pic_wrap")
        return pid.value

    def pic_final_half(self, pid):
        r = self.L.oh_pic_final_half(self.h, pid)
        if r < 0:
            raise EngineError(f"oh_pic_final_half({pid}) failed")
        return r

    def pic_set_final_half(self, pid, half):
        self._chk(self.L.oh_pic_set_final_half(self.h, pid, half), "oh_pic_set_final_half")

    def pic_free(self, pid):
        self._chk(self.L.oh_pic_free(self.h, pid), "oh_pic_free")

    @staticmethod
    def _plane_args(hp):
        n = len(hp.planes)
        d = (C.c_void_p * 3)(*[pl.ctypes.data for pl in hp.planes] + [None] * (3 - n))
        s = (C.c_ssize_t * 3)(*[pl.strides[0] for pl in hp.planes] + [0] * (3 - n))
        return d, s

    def pic_upload(self, pid, hp):
        d, s = self._plane_args(hp)
        self._chk(self.L.oh_pic_upload(self.h, pid, d, s), "oh_pic_upload")

    def pic_download(self, pid, params):
        hp = F.HostPic(params)
        d, s = self._plane_args(hp)
        self._chk(self.L.oh_pic_download(self.h, pid, d, s), "oh_pic_download")
        return hp

    def pic_download_start(self, pid, left=0, right=0, top=0, bottom=0):
        """first half of the output fetch (oh_pic_download_start): the copies are enqueued behind the picture's batch; returns the handle
        oh_download_finish takes — on any thread, while this one keeps driving the engine"""
        d = C.c_void_p()
        win = OhWindow(left, right, top, bottom)
        self._chk(self.L.oh_pic_download_start(self.h, pid, C.byref(win), C.byref(d)), "oh_pic_download_start")
        return d

    def download_finish(self, handle, params, left=0, right=0, top=0, bottom=0, planes=True):
        """second half (oh_download_finish): waits for the copies, returns the packed planes; planes=False: hands no destination over
        (the call must return OH_E_ARG and still release the staging buffer) and returns the error code"""
        dt = np.uint8 if params.bit_depth <= 8 else np.uint16
        W, H = params.width - left - right, params.height - top - bottom
        if not planes:
            return self.L.oh_download_finish(self.h, handle, None, None)
        out = []
        for c in range(F.n_planes(params)):
            hs = 1 if c and params.chroma_format_idc in (1, 2) else 0
            vs = 1 if c and params.chroma_format_idc == 1 else 0
            out.append(np.zeros((H >> vs, W >> hs), dt))
        n = len(out)
        d = (C.c_void_p * 3)(*[pl.ctypes.data for pl in out] + [None] * (3 - n))
        s = (C.c_ssize_t * 3)(*[pl.strides[0] for pl in out] + [0] * (3 - n))
        self._chk(self.L.oh_download_finish(self.h, handle, d, s), "oh_download_finish")
        return out

    def pic_download_window(self, pid, params, left=0, right=0, top=0, bottom=0, pad=0):
        """the picture inside its conformance window as packed numpy planes (pad: extra bytes per destination row, to
        exercise pitches larger than the row)"""
        dt = np.uint8 if params.bit_depth <= 8 else np.uint16
        W, H = params.width - left - right, params.height - top - bottom
        planes = []
        for c in range(F.n_planes(params)):
            hs = 1 if c and params.chroma_format_idc in (1, 2) else 0
            vs = 1 if c and params.chroma_format_idc == 1 else 0
            planes.append(np.zeros((H >> vs, (W >> hs) + pad // dt().itemsize), dt))
        n = len(planes)
        d = (C.c_void_p * 3)(*[pl.ctypes.data for pl in planes] + [None] * (3 - n))
        s = (C.c_ssize_t * 3)(*[pl.strides[0] for pl in planes] + [0] * (3 - n))
        win = OhWindow(left, right, top, bottom)
        self._chk(self.L.oh_pic_download_window(self.h, pid, C.byref(win), d, s), "oh_pic_download_window")
        return [pl[:, :pl.shape[1] - pad // dt().itemsize] if pad else pl for pl in planes]

    def pics_md5(self, pids):
        """[[16-byte MD5 of plane 0, 1, 2]] of finished pictures, computed on the GPU (oh_pics_md5)"""
        n = len(pids)
        ids = (C.c_int * max(n, 1))(*pids)
        out = (C.c_uint8 * (48 * max(n, 1)))()
        self._chk(self.L.oh_pics_md5(self.h, ids, n, out), "oh_pics_md5")
        raw = bytes(out)
        return [[raw[48 * i + 16 * c:48 * i + 16 * c + 16] for c in range(3)] for i in range(n)]

    def pic_device_planes(self, pid):
        p = (C.c_void_p * 3)()
        st, w, h = (C.c_int32 * 3)(), (C.c_int32 * 3)(), (C.c_int32 * 3)()
        self._chk(self.L.oh_pic_device_planes(self.h, pid, p, st, w, h), "oh_pic_device_planes")
        return [(p[c], st[c], w[c], h[c]) for c in range(3)]

    def pic_upsample(self, dst_pid, src_pid, u):
        """SHVC inter-layer reference: resample picture src_pid into dst_pid (u: frame.upsample_setup)"""
        self._chk(self.L.oh_pic_upsample(self.h, dst_pid, src_pid, C.byref(u)), "oh_pic_upsample")

    def memory(self):
        """oh_engine_memory: what the engine holds for work lists (arenas alive, their bytes, free in the pool, staging buffers, their bytes, deferred lists)"""
        out = (C.c_uint64 * 6)()
        self._chk(self.L.oh_engine_memory(self.h, out), "oh_engine_memory")
        return dict(zip(("arenas", "arena_bytes", "arenas_free", "stages", "stage_bytes", "deferred"), map(int, out)))

    # ---- work lists ----
    def pic_upsample_ctbs(self, dst_pid, src_pid, u, log2_ctb_size, ctb_addrs):
        a = (C.c_uint32 * max(len(ctb_addrs), 1))(*ctb_addrs)
        self._chk(self.L.oh_pic_upsample_ctbs(self.h, dst_pid, src_pid, C.byref(u), log2_ctb_size, a, len(ctb_addrs)), "oh_pic_upsample_ctbs")

    def frame_upload(self, frame):
        df = C.c_void_p()
        self._chk(self.L.oh_frame_upload(self.h, C.byref(frame), C.byref(df)), "oh_frame_upload")
        return df

    def frames_upload(self, frames):
        """n work lists, one set of preparation launches (oh_frames_upload); returns the device frames"""
        n = len(frames)
        fs = (C.POINTER(F.OhFrame) * n)(*[C.pointer(f) for f in frames])
        out = (C.c_void_p * n)()
        self._chk(self.L.oh_frames_upload(self.h, fs, n, out), "oh_frames_upload")
        return [C.c_void_p(out[i]) for i in range(n)]

    def frame_execute(self, df):
        self._chk(self.L.oh_frame_execute(self.h, df), "oh_frame_execute")
        self.n_batches = getattr(self, "n_batches", 0) + 1

    def frames_execute(self, dfs):
        """one launch per pass over all the (mutually independent) pictures in dfs"""
        arr = (C.c_void_p * len(dfs))(*[d.value if isinstance(d, C.c_void_p) else d for d in dfs])
        self.n_batches = getattr(self, "n_batches", 0) + (len(dfs) + 31) // 32
        self._chk(self.L.oh_frames_execute(self.h, arr, len(dfs)), "oh_frames_execute")

    def frame_download_bs(self, df, params):
        """(vertical, horizontal) boundary-strength grids of an uploaded work list as the deblock pass reads them"""
        n = F.bs_size(params)
        v, h = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        self._chk(self.L.oh_frame_download_bs(self.h, df, v.ctypes.data, h.ctypes.data, n), "oh_frame_download_bs")
        return v, h

    def frame_free(self, df):
        self._chk(self.L.oh_frame_free(self.h, df), "oh_frame_free")

    def frame_release(self, df):
        """stream-ordered free (no host wait): call after the last execute of df was enqueued"""
        self._chk(self.L.oh_frame_release(self.h, df), "oh_frame_release")

    def frame_submit(self, frame):
        self._chk(self.L.oh_frame_submit(self.h, C.byref(frame)), "oh_frame_submit")

    # ---- profiling ----
    def profile(self, enable):
        self._chk(self.L.oh_engine_profile(self.h, int(enable)), "oh_engine_profile")

    def pass_times(self, reset=False):
        ms = (C.c_double * OH_N_PASSES)()
        n = C.c_uint64()
        self._chk(self.L.oh_engine_pass_times(self.h, ms, C.byref(n), int(reset)), "oh_engine_pass_times")
        return dict(zip(PASS_NAMES, list(ms))), n.value

    def intra_launch_times(self, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        self._chk(self.L.oh_engine_intra_launch_times(self.h, C.byref(ms), C.byref(n), int(reset)), "oh_engine_intra_launch_times")
        return ms.value, n.value

    HOST_TIME_NAMES = ("upload", "upload_count_loops", "upload_arena", "upload_wait_for_staging_buffer", "upload_memcpy_to_pinned",
                       "upload_enqueue", "execute", "execute_wait_for_preparation", "release")

    def host_times(self, reset=False):
        """{slot: (milliseconds, calls)} of host wall time inside the hand-over path (include/ohevc_hip.h: OhHostTime)"""
        n = len(self.HOST_TIME_NAMES)
        ms, calls = (C.c_double * n)(), (C.c_uint64 * n)()
        self._chk(self.L.oh_engine_host_times(self.h, ms, calls, n, int(reset)), "oh_engine_host_times")
        return {k: (ms[i], calls[i]) for i, k in enumerate(self.HOST_TIME_NAMES)}

    def upload_bytes(self, reset=False):
        return int(self.L.oh_engine_upload_bytes(self.h, int(reset)))

    def stream(self):
        return self.L.oh_engine_stream(self.h)


def remap_frame(frame, id_map):
    """copy of an OhFrame header whose picture ids are translated through id_map (host id -> engine id)"""
    g = F.OhFrame()
    C.memmove(C.byref(g), C.byref(frame), C.sizeof(F.OhFrame))
    g.cur_pic = id_map[frame.cur_pic]
    for i in range(F.OH_MAX_REFS):
        r = frame.ref_pics[i]
        g.ref_pics[i] = id_map.get(r, -1)
    return g
