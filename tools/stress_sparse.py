#!/usr/bin/env python3
"""stress: the sparse-pin cases through the engine many times, lists prepared afresh every time (as the test does); prints where a
picture differs from the reference decoder's and whether the prepared lists themselves changed between repetitions"""
import hashlib
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np
import test_sparse_pin as T
from openhevc_amd import frame as F
from openhevc_amd.engine import Engine, remap_frame
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seen = {}
bad = 0
for trial in range(n):
    for case in T.CASES:
        name = case[0]
        lists, want = T.sparse_work_lists(case)
        h = hashlib.md5()
        for a, cur, _ in lists:
            for k in sorted(a):
                h.update(k.encode()); h.update(np.ascontiguousarray(a[k]).tobytes())
        for p in want:
            for pl in p: h.update(np.ascontiguousarray(pl).tobytes())
        if seen.setdefault(name, h.hexdigest()) != h.hexdigest():
            print("trial", trial, name, "HOST SIDE CHANGED: lists / reference pictures differ from the first preparation", flush=True)
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
                    print("trial", trial, name, "picture", k, "plane", c, "mismatches", len(d), "first", d[:6].tolist(), "bbox", d.min(0).tolist(), d.max(0).tolist(),
                          "got", [int(got.visible(c)[tuple(q)]) for q in d[:4]], "want", [int(want[k][c][tuple(q)]) for q in d[:4]], flush=True)
        eng.close()
print("done", n, "trials,", bad, "bad planes")
