#!/usr/bin/env python3
"""Diagnostic: where does an intra CTU workgroup spend its cycles?  Needs the -DOH_STAMPS build
(make -C openhevc_amd libohevc_hip_stamps.so) and a GPU.  Prints, per launch class, shader
clock, cycles per sub-level and the split between block work and barrier wait."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OHEVC_STAMPS"] = "1"
from openhevc_amd import engine as E  # noqa: E402
from openhevc_amd import frame as F  # noqa: E402

E.lib_path = lambda: os.path.join(F.PKG_DIR, "libohevc_hip_stamps.so")


def main():
    w, h = (int(v) for v in (sys.argv[1:3] or (1920, 1080)))
    knobs = dict(kv.split("=") for kv in sys.argv[3:])          # e.g. split_pct=100
    knobs = {k: int(v) for k, v in knobs.items()}
    p = F.pic_params(w, h)
    rec = F.Recorder(p)
    eng = E.Engine(0)
    eng.L.oh_debug_read.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
    ids = [eng.pic_alloc(p) for _ in range(3)]
    rng = np.random.default_rng(0)
    for i in ids[:2]:
        eng.pic_upload(i, F.HostPic(p, rng=rng))
    for st, name in ((0, "I picture"), (2, "B picture")):
        f = rec.synth(F.synth_params(st, 7, **knobs), ids[2], ids[:2])
        df = eng.frame_upload(f)
        for _ in range(3):
            eng.frame_execute(df)
        eng.sync()
        buf = (C.c_uint64 * (16 + 4000 * 16))()
        eng.L.oh_debug_read(eng.h, buf, len(buf))
        n = min(int(buf[0]), 4000)
        r = np.array(buf[16:16 + n * 16], dtype=np.float64).reshape(n, 16)
        r = r[r[:, 0] > 0]
        nsub, cyc, rt = r[:, 0], r[:, 1], r[:, 2]
        clk = cyc.sum() / rt.sum() * 100.0
        print(f"{name}: {n} launches stamped, shader clock ~{clk:.0f} MHz")
        print(f"  cycles per sub-level (wave 0 of WG 0): {cyc.sum() / nsub.sum():.0f}   "
              f"= block work {r[:, 3].sum() / nsub.sum():.0f} + barrier wait {r[:, 4].sum() / nsub.sum():.0f}")
        print(f"  inside a block: gather+substitute+smooth {r[:, 5].sum() / nsub.sum():.0f}, publish {r[:, 6].sum() / nsub.sum():.0f}, "
              f"predict+store {r[:, 7].sum() / nsub.sum():.0f}  (per sub-level, wave 0 only)")
        print(f"  inside a four-block pass: descriptors+gather+smooth {r[:, 11].sum() / nsub.sum():.0f}, publish+DC {r[:, 12].sum() / nsub.sum():.0f}, "
              f"predict {r[:, 13].sum() / nsub.sum():.0f}, add+store {r[:, 14].sum() / nsub.sum():.0f}  (per sub-level, wave 0 only)")
        big = r[np.argsort(-cyc)[:5]]
        for row in big:
            print(f"    n_sub {int(row[0]):3d} blocks {int(row[9]):3d} grid {int(row[8]):3d}: {row[1]:.0f} cycles, {row[2] * 10:.0f} ns, "
                  f"{row[1] / row[0]:.0f} cyc/sub-level")
        eng.frame_free(df)
    eng.close()


if __name__ == "__main__":
    main()
