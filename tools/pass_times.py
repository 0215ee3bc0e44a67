#!/usr/bin/env python3
"""Per-pass device time of ONE picture executed alone (no other stream), averaged over repeats.
usage: pass_times.py WIDTH HEIGHT BITDEPTH SLICE_TYPE [knob=value ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openhevc_amd import frame as F  # noqa: E402
from openhevc_amd import parallel as P  # noqa: E402
from openhevc_amd.engine import Engine  # noqa: E402


def main():
    w, h, bd, st = (int(v) for v in sys.argv[1:5])
    knobs = dict(P.default_synth_knobs())
    for kv in sys.argv[5:]:
        k, v = kv.split("=")
        knobs[k] = int(v)
    p = F.pic_params(w, h, bit_depth=bd)
    rec = F.Recorder(p)
    eng = Engine(0)
    ids = [eng.pic_alloc(p) for _ in range(3)]
    rng = np.random.default_rng(0)
    for i in ids[:2]:
        eng.pic_upload(i, F.HostPic(p, rng=rng))
    f = rec.synth(F.synth_params(st, 7, **knobs), ids[2], ids[:2])
    st_ = P.frame_stats(f)
    df = eng.frame_upload(f)
    for _ in range(3):
        eng.frame_execute(df)
    eng.sync()
    eng.profile(True)
    n = 20
    for _ in range(n):
        eng.frame_execute(df)
    ms, cnt = eng.pass_times(reset=True)
    b = 2 if bd > 8 else 1
    ab = P.algorithmic_bytes(st_, b)
    print(f"{w}x{h} {bd}-bit slice_type {st} knobs {knobs}")
    for k in ms:
        us = ms[k] / cnt * 1e3
        print(f"  {k:10s} {us:9.1f} us   algorithmic {ab[k] / 1e6:8.2f} MB -> {ab[k] / max(us, 1e-9) / 1e3:8.1f} GB/s")
    eng.frame_free(df)
    eng.close()


if __name__ == "__main__":
    main()
