#!/usr/bin/env python3
"""One-off, longer version of tests/test_gpu_parity.py::test_random_configurations (needs a GPU and the built oracle):
352 more seeded configurations up to 2400x1360, every picture bit-exact against the CPU checker.  Run from the repo root
after kernel rewrites; last run: end of round 1 (a third of the pictures with bs_from_motion), all ok."""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_parity as T
from openhevc_amd.engine import Engine
eng = Engine(0)
fn = T.test_random_configurations
for seed, count, mw, mh in [(777001, 150, 34, 26), (777002, 150, 40, 30), (777003, 40, 160, 90), (777004, 12, 300, 170)]:
    fn.__wrapped__(eng, seed, count, mw, mh) if hasattr(fn, "__wrapped__") else fn(eng, seed, count, mw, mh)
    print("sweep", seed, count, "ok", flush=True)
eng.close()
