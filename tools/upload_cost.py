#!/usr/bin/env python3
"""Host-side cost of handing one picture to the engine: oh_frame_upload (validation, MC job / descriptor building, staging
copy, H2D) per 4K Main10 picture, dense and sparse work lists.  usage: upload_cost.py [WIDTH HEIGHT BITDEPTH]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openhevc_amd import frame as F  # noqa: E402
from openhevc_amd import parallel as P  # noqa: E402
from openhevc_amd.engine import Engine  # noqa: E402


def main():
    w, h, bd = (int(v) for v in (sys.argv[1:4] or (3840, 2160, 10)))
    p = F.pic_params(w, h, bit_depth=bd)
    rec = F.Recorder(p)
    eng = Engine(0)
    ids = [eng.pic_alloc(p) for _ in range(3)]
    for st, name in ((2, "B"), (0, "I")):
        for sparse in (0, 100):
            t0 = time.perf_counter()
            f = rec.synth(F.synth_params(st, 7, **dict(P.default_synth_knobs(), sparse_pct=sparse)), ids[2], ids[:2] if st else [])
            t_synth = time.perf_counter() - t0
            eng.frame_free(eng.frame_upload(f))               # warm
            eng.sync()
            n = 10
            t0 = time.perf_counter()
            dfs = [eng.frame_upload(f) for _ in range(n)]
            eng.sync()
            t_up = (time.perf_counter() - t0) / n
            for d in dfs:
                eng.frame_free(d)
            mb = (f.n_pu * 20 + f.n_tu * 12 + f.n_intra * 12 + 2 * f.bs_size + (0 if sparse == 100 else 2 * f.n_coeff) + 4 * f.n_sparse) / 1e6
            print(f"{name} picture, sparse_pct {sparse:3d}: work list {mb:6.2f} MB, oh_frame_upload {t_up * 1e3:6.2f} ms "
                  f"({1 / t_up:6.0f} pictures/s per host thread); synthetic generation {t_synth * 1e3:6.1f} ms")
    eng.close()


if __name__ == "__main__":
    main()
