"""CPU backend for the wave schedule, built on the CHECKER (oracle/).  Test infrastructure:
used by the gloo multi-process tests and by bench.py's cpu_baseline leg, never by the product."""
import ctypes as C

import numpy as np
import torch

from openhevc_amd import frame as F
from openhevc_amd import parallel as P
from oracle_lib import OhHostPicC, oracle


class OracleBackend(P.Backend):
    def __init__(self, params, plan, knobs=None):
        self.params, self.plan = params, plan
        self.store = P.PictureStore(torch, torch.device("cpu"), params, plan, len(plan.waves), len(plan.tail))
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
