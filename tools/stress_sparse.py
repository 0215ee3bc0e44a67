#!/usr/bin/env python3
"""stress: the sparse-pin cases through the engine many times; prints where a picture differs from the reference decoder's"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import test_sparse_pin as T
from openhevc_amd import frame as F
from openhevc_amd.engine import Engine, remap_frame
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
prep = {c[0]: T.sparse_work_lists(c) for c in T.CASES}
bad = 0
for trial in range(n):
    for name, (lists, want) in prep.items():
        eng = Engine(0); ids = {}
        for k, (a, cur, _) in enumerate(lists):
            ff = F.FrameFromArrays(a); f = ff.frame
            for i in [cur] + [f.ref_pics[r] for r in range(F.OH_MAX_REFS) if f.ref_pics[r] >= 0]:
                if i not in ids: ids[i] = eng.pic_alloc(f.p)
            eng.frame_submit(remap_frame(f, ids))
            got = eng.pic_download(ids[cur], f.p)
            for c in range(3):
                d = np.argwhere(got.visible(c) != want[k][c])
                if len(d):
                    bad += 1
                    print("trial", trial, name, "picture", k, "plane", c, "mismatches", len(d), "first", d[:6].tolist(), "bbox", d.min(0).tolist(), d.max(0).tolist(), flush=True)
        eng.close()
print("done", n, "trials,", bad, "bad planes")
