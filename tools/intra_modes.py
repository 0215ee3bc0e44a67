#!/usr/bin/env python3
"""Experiment: a batch of N pictures of one slice type on S engines (streams, one host thread each), timed as a whole and per pass.
The intra pass's form is chosen by the environment (OHEVC_INTRA_MODE = levels | dag | direct | unset: automatic), read once per
process, so this runs once per form.
usage: intra_modes.py SLICE_TYPE(0 I, 1 P, 2 B) BATCH STREAMS [WIDTH HEIGHT BITDEPTH] [knob=value ...]"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openhevc_amd import frame as F  # noqa: E402
from openhevc_amd import parallel as P  # noqa: E402
from openhevc_amd.engine import Engine  # noqa: E402


def main():
    st, nb, ns = (int(v) for v in sys.argv[1:4])
    rest = sys.argv[4:]
    w, h, bd = 3840, 2160, 10
    if len(rest) >= 3 and "=" not in rest[0]:
        w, h, bd = (int(v) for v in rest[:3])
        rest = rest[3:]
    knobs = dict(P.default_synth_knobs())
    for kv in rest:
        k, v = kv.split("=")
        knobs[k] = int(v)
    p = F.pic_params(w, h, bit_depth=bd)
    rec = F.Recorder(p)
    rng = np.random.default_rng(0)
    ref_host = [F.HostPic(p, rng=rng) for _ in range(2)]
    distinct = 4
    engines, batches = [], []
    lists = [F.FrameCopy(rec.synth(F.synth_params(st, 100 + k, **knobs), 0, [1, 2] if st else [])) for k in range(distinct)]
    stats = P.frame_stats(lists[0].frame)
    for s in range(ns):
        eng = Engine(0)
        refs = [eng.pic_alloc(p) for _ in range(2)]
        for i, hp in zip(refs, ref_host):
            eng.pic_upload(i, hp)
        dfs = []
        for k in range(nb):
            cur = eng.pic_alloc(p)
            dfs.append(eng.frame_upload(lists[(k + s) % distinct].with_ids(cur, refs if st else [])))
        engines.append(eng)
        batches.append(dfs)
    reps = 6

    def run(eng, dfs, n):
        for _ in range(n):
            eng.frames_execute(dfs)
        eng.sync()

    for eng, dfs in zip(engines, batches):
        run(eng, dfs, 2)
    for eng in engines:
        eng.profile(True)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(e, d, reps)) for e, d in zip(engines, batches)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    pics = reps * nb * ns
    mode = os.environ.get("OHEVC_INTRA_MODE", "auto")
    print(f"mode {mode:7s} slice_type {st} batch {nb} streams {ns}: {dt / reps * 1e3:8.2f} ms per round of batches, {pics / dt:8.1f} pictures/s, "
          f"{pics * w * h / dt / 1e9:6.2f} Gpix/s   (picture 0: {stats['n_intra']} intra blocks covering {stats['intra'] / stats['samples'] * 100:.0f} % of the samples, {stats['n_levels']} levels)")
    tot = {}
    for eng in engines:
        ms, cnt = eng.pass_times(reset=True)
        for k in ms:
            tot[k] = tot.get(k, 0.0) + ms[k] / reps
    print("   per batch, mean over streams (ms): " + "  ".join(f"{k} {v / ns:.3f}" for k, v in tot.items()))
    for eng in engines:
        eng.close()


if __name__ == "__main__":
    main()
