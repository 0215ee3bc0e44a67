#!/usr/bin/env python3
"""GPU box: boundary-strength grids the engine derives from a real stream's motion field (hooked front end, bs_in) against the grids the
reference decoder derived itself, cell by cell.  usage: check_bs_stream.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import refdec  # noqa: E402
import streamgen  # noqa: E402
from openhevc_amd import frame as F  # noqa: E402
from openhevc_amd.engine import Engine, remap_frame  # noqa: E402


def main():
    kw = dict(n_pictures=3, gop=2)
    data, _ = streamgen.write_stream(416, 240, 7, **kw)
    grids = []
    refdec.record_work_lists(data, lambda f, cur, poc: grids.append((np.ctypeslib.as_array(f.vertical_bs, (f.bs_size,)).copy(),
                                                                     np.ctypeslib.as_array(f.horizontal_bs, (f.bs_size,)).copy())))
    eng = Engine(0)
    ids, k = {}, [0]

    def on_picture(f, cur, poc):
        for i in [cur] + [f.ref_pics[r] for r in range(F.OH_MAX_REFS) if f.ref_pics[r] >= 0]:
            if i not in ids:
                ids[i] = eng.pic_alloc(f.p)
        df = eng.frame_upload(remap_frame(f, ids))
        v, h = eng.frame_download_bs(df, f.p)
        wv, wh = grids[k[0]]
        n = min(len(v), len(wv))
        dv, dh = np.nonzero(v[:n] != wv[:n])[0], np.nonzero(h[:n] != wh[:n])[0]
        print("picture", k[0], "cells", n, "vertical differ", len(dv), dv[:8], v[dv[:8]], wv[dv[:8]], "horizontal differ", len(dh), dh[:8], h[dh[:8]], wh[dh[:8]],
              "nonzero want", int((wv != 0).sum()), "got", int((v != 0).sum()))
        eng.frame_execute(df)
        eng.sync()
        eng.frame_free(df)
        k[0] += 1
    refdec.record_work_lists(data, on_picture, bs_from_motion=True)


if __name__ == "__main__":
    main()
