"""CPU backend for the wave schedule, built on the CHECKER (oracle/).  Test infrastructure:
used by the gloo multi-process tests and by bench.py's cpu_baseline leg, never by the product."""
import ctypes as C

import numpy as np
import torch

from openhevc_amd import frame as F
from openhevc_amd import parallel as P
from oracle_lib import OhHostPicC, oracle, ref


class OracleBackend(P.Backend):
    def __init__(self, params, plan, knobs=None, group=None):
        self.params, self.plan = params, plan
        self.store = P.PictureStore(torch, torch.device("cpu"), params, plan, len(plan.waves), len(plan.tail), group=group)
        names = self.store.names()
        self.ids = {n: i for i, n in enumerate(names)}
        self.final = {n: 0 for n in names}
        self.bpp = 2 if params.bit_depth > 8 else 1
        knobs = dict(P.default_synth_knobs(), **(knobs or {}))
        self.recs, self.frames = {}, {}
        for pic in plan.pictures():
            rec = F.Recorder(params)                       # one recorder per picture keeps its arrays alive
            sp = F.synth_params(pic.slice_type, pic.seed, n_refs=len(pic.refs), **knobs)
            self.frames[pic.name] = rec.synth(sp, self.ids[pic.name], [self.ids[r] for r in pic.refs])
            self.recs[pic.name] = rec
        self.executed = 0

    def _pic_array(self):
        """OhHostPic table: every picture's FINAL half, except that the picture being decoded is
        handled in execute() (decode into half 0, SAO result copied to half 1 like the engine)"""
        arr = (OhHostPicC * len(self.ids))()
        for name, i in self.ids.items():
            self._fill(arr[i], name, self.final[name])
        return arr

    def _fill(self, slot, name, half):
        t = self.store.halves(name)[half]
        base = t.data_ptr()
        for c in range(F.n_planes(self.params)):
            w, h = F.plane_dims(self.params, c)
            slot.data[c] = base + self.store.offsets[c]
            slot.stride[c] = self.store.strides[c] * self.bpp
            slot.width[c], slot.height[c] = w, h
        slot.bit_depth = self.params.bit_depth

    def wave_tensor(self, wave):
        return self.store.waves[wave]

    def execute(self, name):
        f = self.frames[name]
        arr = self._pic_array()
        half = 1 if (self.params.sao_enabled and f.sao) else 0
        self._fill(arr[self.ids[name]], name, half)        # the oracle works in place: decode into the final half
        assert oracle().oh_or_frame(C.byref(f), arr) == 0
        self.final[name] = half
        self.executed += 1

    def final_half(self, name):
        return self.final[name]

    def set_final_half(self, name, half):
        self.final[name] = half

    def picture(self, name):
        """visible samples of the finished picture, list of numpy planes"""
        t = self.store.halves(name)[self.final[name]].numpy()
        out = []
        for c in range(F.n_planes(self.params)):
            w, h = F.plane_dims(self.params, c)
            st = self.store.strides[c] * self.bpp
            raw = t[self.store.offsets[c]: self.store.offsets[c] + st * h].reshape(h, st)
            out.append(raw.view(np.uint16 if self.bpp == 2 else np.uint8)[:, :w].copy())
        return out


class _PaddedHalf:
    """one picture buffer with slack on both sides: the reference's kernels assume its edge-padded frames (32 samples
    and more around every plane, libavcodec get_buffer) and read a little past a bare plane"""
    PAD = 1 << 16

    def __init__(self, nbytes):
        self.buf = np.zeros(nbytes + 2 * self.PAD, np.uint8)
        self.nbytes = nbytes

    def data_ptr(self):
        return self.buf.ctypes.data + self.PAD

    def numpy(self):
        return self.buf[self.PAD:self.PAD + self.nbytes]


class _PaddedStore:
    """PictureStore's interface (single rank) over padded buffers"""

    def __init__(self, params, plan):
        self.half_bytes, self.strides, self.offsets = F.half_layout(params)
        self.plan = plan
        self._names = [("ref", w, 0) for w in range(len(plan.waves))] + [p.name for p in plan.tail]
        self._h = {n: (_PaddedHalf(self.half_bytes), _PaddedHalf(self.half_bytes)) for n in self._names}

    def halves(self, name):
        return self._h[name]

    def names(self):
        return list(self._names)


class RefBackend(OracleBackend):
    """Same schedule, executed by the REFERENCE's own C kernels (oracle/_ref/libohevc_ref.so: put_hevc_{q,e}pel*,
    idct*, intra_pred, ff_hevc_hls_filters ... driven by oracle/ref_harness.c::ref_frame).  bench.py's cpu_baseline of
    kind "reference"; the library is built in the container from /root/reference and travels to the GPU box as a
    git-ignored binary."""

    def __init__(self, params, plan, knobs=None):
        assert plan.world == 1, "CPU baseline: one rank"
        super().__init__(params, plan, knobs)
        self.store = _PaddedStore(params, plan)            # same names / ids as the store it replaces
        self.lib = ref()
        self._scratch = _PaddedHalf(self.store.half_bytes)

    def _planes(self, name, half):
        base = self.store.halves(name)[half].data_ptr()
        n = F.n_planes(self.params)
        d = (C.c_void_p * 3)(*[base + self.store.offsets[c] for c in range(n)] + [None] * (3 - n))
        s = (C.c_ssize_t * 3)(*[self.store.strides[c] * self.bpp for c in range(n)] + [0] * (3 - n))
        return d, s

    def execute(self, name):
        f = self.frames[name]
        half = 1 if (self.params.sao_enabled and f.sao) else 0
        d, s = self._planes(name, half)
        by_id = {i: n for n, i in self.ids.items()}
        n_refs = max([k for k in range(F.OH_MAX_REFS) if f.ref_pics[k] >= 0] + [-1]) + 1
        refs = ((C.c_void_p * 3) * max(n_refs, 1))()
        for k in range(n_refs):
            rn = by_id[f.ref_pics[k]]
            rd, _ = self._planes(rn, self.final[rn])
            for c in range(3):
                refs[k][c] = rd[c]
        sb = self._scratch.data_ptr()
        d2 = (C.c_void_p * 3)(*[sb + self.store.offsets[c] for c in range(F.n_planes(self.params))] + [None] * (3 - F.n_planes(self.params)))
        assert self.lib.ref_frame(C.byref(f), d, s, C.cast(refs, C.c_void_p), n_refs, s, d2) == 0
        self.final[name] = half
        self.executed += 1
