"""Random but CONSISTENT inputs of the boundary-strength derivation (test infrastructure): a coding quadtree per CTB whose
leaves are the blocks ff_hevc_deblocking_boundary_strengths is called for (a whole coding block, or the transform units of a
coded one), prediction units (2Nx2N / 2NxN / Nx2N / NxN / asymmetric) with motion drawn from a small pool so that equal,
close and distant vectors, same and swapped reference pictures all occur, cbf per transform unit, and per-CTB slice / tile
edge flags."""
import ctypes as C

import numpy as np

from openhevc_amd import frame as F

MVF_DT = np.dtype([("mv", "<i2", (2, 2)), ("poc", "<i4", (2,)), ("pred_flag", "<u4"), ("ref_idx", "u1", (2,)), ("pad", "u1", (2,))])
assert MVF_DT.itemsize == 24 == C.sizeof(F.OhMvField)


def make_inputs(p, seed, intra_pct=15, flags_pct=30, disabled_pct=5):
    rng = np.random.default_rng(seed)
    lpu, ltu, lc, lcb = p.log2_min_pu_size, p.log2_min_tb_size, p.log2_ctb_size, p.log2_min_cb_size
    mpw, mph = p.width >> lpu, p.height >> lpu
    mtw, mth = p.width >> ltu, p.height >> ltu
    ctbw, ctbh = (p.width + (1 << lc) - 1) >> lc, (p.height + (1 << lc) - 1) >> lc
    mvf = np.zeros((mph, mpw), MVF_DT)
    cbf = np.zeros((mth, mtw), np.uint8)
    call = np.zeros((mth, mtw), np.uint8)
    flags = np.zeros((ctbh, ctbw), np.uint8)
    pool = [(int(rng.integers(-64, 64)), int(rng.integers(-64, 64))) for _ in range(6)]
    pocs = [0, 4, 8, 8]                                      # two lists may hold the same picture

    def motion():
        m = np.zeros((), MVF_DT)
        if rng.integers(0, 100) < intra_pct:
            return m                                         # PF_INTRA: everything 0 like the reference's default value
        pf = int(rng.choice([1, 2, 3, 3]))
        for l in range(2):
            if pf & (1 << l):
                bx, by = pool[int(rng.integers(0, len(pool)))]
                m["mv"][l] = (bx + int(rng.integers(-5, 6)), by + int(rng.integers(-5, 6)))
                ri = int(rng.integers(0, len(pocs)))
                m["ref_idx"][l] = ri
                m["poc"][l] = pocs[ri]
        m["pred_flag"] = pf
        return m

    def fill_pu(x, y, w, h):
        mvf[y >> lpu:(y + h) >> lpu, x >> lpu:(x + w) >> lpu] = motion()

    def tu_tree(x, y, log2, depth, enabled):
        if x >= p.width or y >= p.height:
            return
        if log2 > ltu and (log2 > 5 or (depth < 2 and rng.integers(0, 100) < 40)):
            for k in range(4):
                tu_tree(x + (k & 1) * (1 << (log2 - 1)), y + (k >> 1) * (1 << (log2 - 1)), log2 - 1, depth + 1, enabled)
            return
        n = 1 << log2
        if rng.integers(0, 100) < 55:
            cbf[y >> ltu:(y + n) >> ltu, x >> ltu:(x + n) >> ltu] = 1
        if enabled:
            call[y >> ltu:(y + n) >> ltu, x >> ltu:(x + n) >> ltu] = log2

    def cu_tree(x, y, log2, enabled):
        if x >= p.width or y >= p.height:
            return
        n = 1 << log2
        if log2 > lcb and (x + n > p.width or y + n > p.height or rng.integers(0, 100) < 50):
            for k in range(4):
                cu_tree(x + (k & 1) * (n >> 1), y + (k >> 1) * (n >> 1), log2 - 1, enabled)
            return
        part = int(rng.integers(0, 8)) if log2 > 3 else int(rng.integers(0, 4))
        q = n >> 2
        rects = {0: [(0, 0, n, n)], 1: [(0, 0, n, n >> 1), (0, n >> 1, n, n >> 1)], 2: [(0, 0, n >> 1, n), (n >> 1, 0, n >> 1, n)],
                 3: [(0, 0, n >> 1, n >> 1), (n >> 1, 0, n >> 1, n >> 1), (0, n >> 1, n >> 1, n >> 1), (n >> 1, n >> 1, n >> 1, n >> 1)],
                 4: [(0, 0, n, q), (0, q, n, n - q)], 5: [(0, 0, n, n - q), (0, n - q, n, q)],
                 6: [(0, 0, q, n), (q, 0, n - q, n)], 7: [(0, 0, n - q, n), (n - q, 0, q, n)]}[part]
        if part == 3 and log2 == 3 and lpu > 2:
            rects = [(0, 0, n, n)]
        first = motion()
        if first["pred_flag"] == 0:                          # an intra CU is intra as a whole
            mvf[y >> lpu:(y + n) >> lpu, x >> lpu:(x + n) >> lpu] = first
        else:
            for (rx, ry, rw, rh) in rects:
                m = motion()
                while m["pred_flag"] == 0:
                    m = motion()
                mvf[(y + ry) >> lpu:(y + ry + rh) >> lpu, (x + rx) >> lpu:(x + rx + rw) >> lpu] = m
        if rng.integers(0, 100) < 60:                        # coded: the function is called per transform unit
            tu_tree(x, y, log2, 0, enabled)
        elif enabled:                                        # skipped / no residual: one call for the coding block
            call[y >> ltu:(y + n) >> ltu, x >> ltu:(x + n) >> ltu] = log2

    for cy in range(ctbh):
        for cx in range(ctbw):
            fl = 0
            if rng.integers(0, 100) < flags_pct:
                fl = int(rng.integers(0, 32))
            if cy == 0:
                fl &= ~3
            if cx == 0:
                fl &= ~12
            flags[cy, cx] = fl
            cu_tree(cx << lc, cy << lc, lc, rng.integers(0, 100) >= disabled_pct)
    return mvf, cbf, call, flags, int(rng.integers(0, 2))


def as_struct(mvf, cbf, call, flags, across_tiles):
    """OhBsInputs over the numpy arrays (keep the arrays alive while it is used)"""
    return F.OhBsInputs(mvf.ctypes.data, cbf.ctypes.data, call.ctypes.data, flags.ctypes.data, across_tiles)
