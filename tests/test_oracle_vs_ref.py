"""Pins the CPU oracle (oracle/oracle.c) against the reference's OWN C kernels.

oracle/_ref/libohevc_ref.so is hevcdsp.c / hevcpred.c / videodsp.c / hevc_filter.c compiled
straight from /root/reference (oracle/Makefile) and driven through the reference's tables
(ff_hevc_dsp_init, ff_hevc_pred_init).  Every comparison is bit-exact.  These tests run only
where the reference tree exists (this container); tests/test_golden.py pins the same oracle
against committed fixtures everywhere else.
"""
import ctypes as C

import numpy as np
import pytest

from oracle_lib import (have_ref, i16p, intp, off_i16p, off_u8p, oracle, pix_dtype, rand_pixels, ref, u8p)

pytestmark = pytest.mark.skipif(not have_ref(), reason="reference tree / oracle/_ref not present")

DEPTHS = [8, 10, 12]


def laplacian_coeffs(rng, n, dense):
    """dequantised-looking coefficient blocks: sparse low-frequency or dense full-range"""
    if dense:
        return rng.integers(-32768, 32768, size=(n, n)).astype(np.int16)
    c = np.zeros((n, n), np.int64)
    k = rng.integers(1, max(2, n * n // 4))
    ys = np.minimum(rng.geometric(0.35, k) - 1, n - 1)
    xs = np.minimum(rng.geometric(0.35, k) - 1, n - 1)
    c[ys, xs] = rng.laplace(0, 600, k).astype(np.int64)
    return np.clip(c, -32768, 32767).astype(np.int16)


@pytest.mark.parametrize("bd", DEPTHS)
def test_idct_all_sizes(bd):
    rng = np.random.default_rng(100 + bd)
    o, r = oracle(), ref()
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        for trial in range(40):
            c = laplacian_coeffs(rng, n, dense=trial % 4 == 0)
            a, b = c.copy(), c.copy()
            o.oh_or_idct(bd, i16p(a), log2)
            r.ref_idct(bd, log2, i16p(b), n)            # col_limit = n: nothing skipped
            assert np.array_equal(a, b), (bd, log2, trial)


@pytest.mark.parametrize("bd", DEPTHS)
def test_idct_col_limit_is_only_a_zero_skip(bd):
    """hevc_cabac.c:1927-1934 passes col_limit = last_x + last_y + 4 (capped); with all coefficients
    beyond that anti-diagonal zero the reference result equals the full transform."""
    rng = np.random.default_rng(200 + bd)
    o, r = oracle(), ref()
    for log2 in (3, 4, 5):
        n = 1 << log2
        for trial in range(30):
            lx, ly = rng.integers(0, n, 2)
            c = np.zeros((n, n), np.int16)
            c[: ly + 1, : lx + 1] = rng.integers(-3000, 3000, size=(ly + 1, lx + 1))
            c[ly, lx] = 77
            m = max(lx, ly)
            col_limit = int(lx + ly + 4)
            if m < 4:
                col_limit = min(4, col_limit)
            elif m < 8:
                col_limit = min(8, col_limit)
            elif m < 12:
                col_limit = min(24, col_limit)
            a, b = c.copy(), c.copy()
            o.oh_or_idct(bd, i16p(a), log2)
            r.ref_idct(bd, log2, i16p(b), col_limit)
            assert np.array_equal(a, b), (bd, log2, lx, ly)


@pytest.mark.parametrize("bd", DEPTHS)
def test_idct_dc_dst_skip_rdpcm(bd):
    rng = np.random.default_rng(300 + bd)
    o, r = oracle(), ref()
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        for trial in range(20):
            c = np.zeros((n, n), np.int16)
            c[0, 0] = rng.integers(-32768, 32768)
            a, b, full = c.copy(), c.copy(), c.copy()
            o.oh_or_idct_dc(bd, i16p(a), log2)
            r.ref_idct_dc(bd, log2, i16p(b))
            o.oh_or_idct(bd, i16p(full), log2)
            assert np.array_equal(a, b)
            assert np.array_equal(a, full)              # DC-only == full transform (OH_TU_IDCT covers both)
            c = rng.integers(-32768, 32768, size=(n, n)).astype(np.int16)
            a, b = c.copy(), c.copy()
            o.oh_or_transform_skip(bd, i16p(a), log2)
            r.ref_transform_skip(bd, i16p(b), log2)
            assert np.array_equal(a, b)
            for mode in (0, 1):
                a, b = c.copy(), c.copy()
                o.oh_or_transform_rdpcm(i16p(a), log2, mode)
                r.ref_transform_rdpcm(bd, i16p(b), log2, mode)
                assert np.array_equal(a, b)
    for trial in range(60):
        c = laplacian_coeffs(rng, 4, dense=trial % 3 == 0)
        a, b = c.copy(), c.copy()
        o.oh_or_idct_4x4_luma(bd, i16p(a))
        r.ref_idct_4x4_luma(bd, i16p(b))
        assert np.array_equal(a, b)


@pytest.mark.parametrize("bd", DEPTHS)
def test_transform_add_saturates(bd):
    rng = np.random.default_rng(400 + bd)
    o, r = oracle(), ref()
    bpp = 1 if bd == 8 else 2
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        for trial in range(10):
            plane = rand_pixels(rng, (n + 3, n + 9), bd, extreme=True)
            res = rng.integers(-(2 << bd), 2 << bd, size=(n, n)).astype(np.int16)
            a, b = plane.copy(), plane.copy()
            stride = plane.strides[0]
            o.oh_or_transform_add(bd, off_u8p(a, stride + 4 * bpp), i16p(res), stride, log2)
            r.ref_transform_add(bd, log2, off_u8p(b, stride + 4 * bpp), i16p(res.copy()), stride)
            assert np.array_equal(a, b)


MC_WIDTHS = [4, 8, 12, 16, 24, 32, 48, 64]


@pytest.mark.parametrize("bd", DEPTHS)
@pytest.mark.parametrize("epel", [0, 1])
def test_mc_all_variants(bd, epel):
    """put / uni / bi / uni_w / bi_w x every fractional position x the reference's width set"""
    rng = np.random.default_rng(500 + bd * 2 + epel)
    o, r = oracle(), ref()
    taps = 4 if epel else 8
    nfrac = 8 if epel else 4
    bpp = 1 if bd == 8 else 2
    pad = 8
    for w in MC_WIDTHS:
        for h in (4, 8, w) if w <= 16 else (8, w):
            src = rand_pixels(rng, (h + 2 * pad, w + 2 * pad), bd, extreme=True)
            sstride = src.strides[0]
            soff = pad * sstride + pad * bpp
            src2 = rng.integers(-(1 << 13), 1 << 14, size=(h, 64)).astype(np.int16)
            fracs = [(0, 0), (rng.integers(1, nfrac), 0), (0, rng.integers(1, nfrac)),
                     (rng.integers(1, nfrac), rng.integers(1, nfrac)), (nfrac - 1, nfrac - 1), (1, nfrac // 2)]
            for fx, fy in fracs:
                fx, fy = int(fx), int(fy)
                denom = int(rng.integers(0, 8))
                wx0, wx1 = (int(v) for v in rng.integers(-128, 128, 2))
                ox0, ox1 = (int(v) for v in rng.integers(-128, 128, 2))
                # put -> int16
                a = np.zeros((h, 64), np.int16); b = a.copy()
                o.oh_or_mc_put(bd, taps, i16p(a), 64, off_u8p(src, soff), sstride, h, fx, fy, w)
                r.ref_mc(bd, epel, 0, C.cast(i16p(b), C.POINTER(C.c_uint8)), 64, off_u8p(src, soff),
                         sstride, None, 0, h, 0, 0, 0, 0, 0, fx, fy, w)
                assert np.array_equal(a[:, :w], b[:, :w]), ("put", bd, epel, w, h, fx, fy)
                dst0 = rand_pixels(rng, (h, w + 8), bd)
                dstride = dst0.strides[0]
                for variant in (1, 2, 3, 4):
                    a, b = dst0.copy(), dst0.copy()
                    if variant == 1:
                        o.oh_or_mc_uni(bd, taps, u8p(a), dstride, off_u8p(src, soff), sstride, h, fx, fy, w)
                    elif variant == 2:
                        o.oh_or_mc_bi(bd, taps, u8p(a), dstride, off_u8p(src, soff), sstride, i16p(src2), 64, h, fx, fy, w)
                    elif variant == 3:
                        o.oh_or_mc_uni_w(bd, taps, u8p(a), dstride, off_u8p(src, soff), sstride, h, denom, wx0, ox0, fx, fy, w)
                    else:
                        o.oh_or_mc_bi_w(bd, taps, u8p(a), dstride, off_u8p(src, soff), sstride, i16p(src2), 64, h,
                                        denom, wx0, wx1, ox0, ox1, fx, fy, w)
                    r.ref_mc(bd, epel, variant, u8p(b), dstride, off_u8p(src, soff), sstride,
                             i16p(src2), 64, h, denom, wx0, wx1, ox0, ox1, fx, fy, w)
                    assert np.array_equal(a, b), (variant, bd, epel, w, h, fx, fy, denom, wx0, wx1, ox0, ox1)


@pytest.mark.parametrize("bd", [8, 10])
def test_edge_emulation_is_coordinate_clamping(bd):
    """videodsp_template.c:26-101 == reading the source with clamped coordinates (what the MC
    pass does instead of materialising edge_emu_buffer)"""
    rng = np.random.default_rng(600 + bd)
    r = ref()
    bpp = 1 if bd == 8 else 2
    W, H, pad = 40, 24, 96
    big = rand_pixels(rng, (H + 2 * pad, W + 2 * pad), bd)
    pic = big[pad:pad + H, pad:pad + W]
    for trial in range(200):
        bw, bh = int(rng.integers(5, 72)), int(rng.integers(5, 72))
        sx, sy = int(rng.integers(-90, W + 20)), int(rng.integers(-90, H + 20))
        if sx + bw > W + pad or sy + bh > H + pad:
            continue
        buf = np.zeros((80, 80), pix_dtype(bd))
        src_ptr = off_u8p(big, (pad + sy) * big.strides[0] + (pad + sx) * bpp)
        r.ref_emulated_edge_mc(bd, u8p(buf), src_ptr, buf.strides[0], big.strides[0],
                               bw, bh, sx, sy, W, H)
        ys = np.clip(np.arange(sy, sy + bh), 0, H - 1)
        xs = np.clip(np.arange(sx, sx + bw), 0, W - 1)
        assert np.array_equal(buf[:bh, :bw], pic[np.ix_(ys, xs)]), (bw, bh, sx, sy)


@pytest.mark.parametrize("bd", DEPTHS)
def test_pred_planar_dc_angular(bd):
    rng = np.random.default_rng(700 + bd)
    o, r = oracle(), ref()
    bpp = 1 if bd == 8 else 2
    for log2 in (2, 3, 4, 5):
        n = 1 << log2
        for trial in range(6):
            top = rand_pixels(rng, (2 * n + 8,), bd, extreme=trial == 0)
            left = rand_pixels(rng, (2 * n + 8,), bd, extreme=trial == 0)
            left[3] = top[3]                               # [-1] is shared
            tp, lp = off_u8p(top, 4 * bpp), off_u8p(left, 4 * bpp)
            blank = rand_pixels(rng, (n, n + 4), bd)
            st = blank.strides[0] // bpp                  # pred_* slots take the stride in PIXELS
            for c_idx in (0, 1):
                a, b = blank.copy(), blank.copy()
                o.oh_or_pred_planar(bd, u8p(a), tp, lp, st, log2)
                r.ref_pred_planar(bd, log2, u8p(b), tp, lp, st)
                assert np.array_equal(a, b)
                a, b = blank.copy(), blank.copy()
                o.oh_or_pred_dc(bd, u8p(a), tp, lp, st, log2, c_idx)
                r.ref_pred_dc(bd, log2, u8p(b), tp, lp, st, c_idx)
                assert np.array_equal(a, b)
                for mode in range(2, 35):
                    a, b = blank.copy(), blank.copy()
                    o.oh_or_pred_angular(bd, u8p(a), tp, lp, st, log2, c_idx, mode)
                    r.ref_pred_angular(bd, log2, u8p(b), tp, lp, st, c_idx, mode)
                    assert np.array_equal(a, b), (bd, log2, c_idx, mode)


def smooth_edge_block(rng, bd, flat):
    """16x16 block around an edge; `flat` gives low-activity content so that the strong / normal
    filters actually trigger (pure noise is always skipped by the beta test)"""
    mx = (1 << bd) - 1
    if flat:
        base = rng.integers(mx // 4, 3 * mx // 4)
        blk = base + rng.integers(-2, 3, size=(16, 16)) * (1 << (bd - 8))
        step = rng.integers(-12, 13) * (1 << (bd - 8))
        blk[:, 8:] += step
        blk = blk + (np.arange(16)[None, :] * rng.integers(-1, 2))
    else:
        blk = rng.integers(0, mx + 1, size=(16, 16))
    return np.clip(blk, 0, mx).astype(pix_dtype(bd))


@pytest.mark.parametrize("bd", DEPTHS)
def test_loop_filters(bd):
    rng = np.random.default_rng(800 + bd)
    o, r = oracle(), ref()
    bpp = 1 if bd == 8 else 2
    hits = 0
    for trial in range(600):
        blk = smooth_edge_block(rng, bd, flat=trial % 5 != 0)
        vertical = trial % 2                      # 1: vertical edge (filter across x)
        if not vertical:
            blk = np.ascontiguousarray(blk.T)
        st = blk.strides[0]
        beta = int(rng.integers(0, 65))
        tc = np.array(rng.integers(0, 25, 2), np.int32)
        if trial % 7 == 0:
            tc[rng.integers(0, 2)] = 0
        no_p = np.array(rng.integers(0, 2, 2) * (trial % 3 == 0), np.uint8)
        no_q = np.array(rng.integers(0, 2, 2) * (trial % 3 == 0), np.uint8)
        off = (4 * st + 8 * bpp) if vertical else (8 * st + 4 * bpp)
        xs, ys = (bpp, st) if vertical else (st, bpp)
        a, b = blk.copy(), blk.copy()
        o.oh_or_loop_filter_luma(bd, off_u8p(a, off), xs, ys, beta, intp(tc), u8p(no_p), u8p(no_q))
        r.ref_loop_filter(bd, (1 if vertical else 0) + (4 if trial % 3 == 0 else 0), off_u8p(b, off), st,
                          beta, intp(tc), u8p(no_p), u8p(no_q))
        assert np.array_equal(a, b), ("luma", bd, trial)
        hits += int(not np.array_equal(a, blk))
        a, b = blk.copy(), blk.copy()
        o.oh_or_loop_filter_chroma(bd, off_u8p(a, off), xs, ys, intp(tc), u8p(no_p), u8p(no_q))
        r.ref_loop_filter(bd, (3 if vertical else 2) + (4 if trial % 3 == 0 else 0), off_u8p(b, off), st,
                          0, intp(tc), u8p(no_p), u8p(no_q))
        assert np.array_equal(a, b), ("chroma", bd, trial)
    assert hits > 100                              # the luma filter really ran (strong and normal paths)


@pytest.mark.parametrize("bd", DEPTHS)
def test_sao_band_and_edge(bd):
    rng = np.random.default_rng(900 + bd)
    o, r = oracle(), ref()
    bpp = 1 if bd == 8 else 2
    for trial in range(120):
        w, h = int(rng.integers(1, 9)) * 8, int(rng.integers(1, 9)) * 8
        if trial % 6 == 0:
            w, h = 64, 64
        src = rand_pixels(rng, (h + 2, w + 2 + 6), bd)
        # smooth-ish content makes every edge category occur
        src = (src.astype(np.int64) // (4 if trial % 2 else 1)).astype(pix_dtype(bd))
        dst0 = rand_pixels(rng, (h + 2, w + 2 + 6), bd)
        st = src.strides[0]
        soff = st + bpp
        offs = np.zeros(5, np.int16)
        offs[1:] = rng.integers(-7, 8, 4) * (1 << (bd - 8 if bd <= 10 else 2))
        c_idx = int(rng.integers(0, 3))
        borders = np.array(rng.integers(0, 2, 4) * (trial % 2), np.int32)
        a, b = dst0.copy(), dst0.copy()
        band = int(rng.integers(0, 32))
        o.oh_or_sao_band(bd, off_u8p(a, soff), off_u8p(src, soff), st, st, i16p(offs), band, w, h)
        r.ref_sao_band(bd, off_u8p(b, soff), off_u8p(src, soff), st, st, i16p(offs), band,
                       intp(borders), w, h, c_idx)
        assert np.array_equal(a, b), ("band", bd, trial)
        for eo in range(4):
            for variant in (0, 1):
                ve = np.zeros(2, np.uint8); he = np.zeros(2, np.uint8); de = np.zeros(4, np.uint8)
                if variant:
                    # flags exist only where the CTB has a neighbour (hevc_filter.c:224-252)
                    ve[:] = rng.integers(0, 2, 2) * (1 - borders[[0, 2]])
                    he[:] = rng.integers(0, 2, 2) * (1 - borders[[1, 3]])
                    de[0] = rng.integers(0, 2) * (1 - borders[0]) * (1 - borders[1])
                    de[1] = rng.integers(0, 2) * (1 - borders[1]) * (1 - borders[2])
                    de[2] = rng.integers(0, 2) * (1 - borders[2]) * (1 - borders[3])
                    de[3] = rng.integers(0, 2) * (1 - borders[0]) * (1 - borders[3])
                a, b = dst0.copy(), dst0.copy()
                o.oh_or_sao_edge(bd, off_u8p(a, soff), off_u8p(src, soff), st, st, i16p(offs), eo,
                                 intp(borders), w, h, variant, u8p(ve), u8p(he), u8p(de))
                r.ref_sao_edge(bd, variant, off_u8p(b, soff), off_u8p(src, soff), st, st, i16p(offs),
                               eo, intp(borders), w, h, c_idx, u8p(ve), u8p(he), u8p(de))
                assert np.array_equal(a, b), ("edge", bd, trial, eo, variant, borders, ve, he, de)
