"""tests/golden/streams/*.npz: work lists recorded INSIDE the reference decoder (its CTU loop with this repository's hooks,
oracle/ref_hooked_unit.c) while it decoded synthetic streams, plus the MD5s of the pictures the unmodified reference decoder
output for the same streams (tests/golden/make_stream_golden.py).  Runs everywhere: the CPU checker here, the HIP engine under
-m gpu — both must reproduce reference output of real bitstream decoding."""
import ctypes as C
import glob
import hashlib
import os

import numpy as np
import pytest

from openhevc_amd import frame as F
from oracle_lib import host_pic_array, oracle

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "streams")
FIXTURES = sorted(glob.glob(os.path.join(GOLD, "*.npz")))


def pictures_of(path):
    z = np.load(path)
    n = int(z["n_pictures"][0])
    frames = []
    for k in range(n):
        pre = f"pic{k}_"
        frames.append(F.FrameFromArrays({key[len(pre):]: z[key] for key in z.files if key.startswith(pre)}))
    return frames, z["md5"].tobytes(), z["stream_md5"].tobytes()


def md5s(hp):
    return b"".join(hashlib.md5(np.ascontiguousarray(hp.visible(c)).tobytes()).digest() for c in range(3))


def test_fixtures_exist():
    assert len(FIXTURES) >= 6


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_checker_reproduces_reference_output(path):
    frames, want, _ = pictures_of(path)
    pics, got = {}, b""
    for ff in frames:
        f = ff.frame
        for i in [f.cur_pic] + [f.ref_pics[k] for k in range(F.OH_MAX_REFS) if f.ref_pics[k] >= 0]:
            if i not in pics:
                pics[i] = F.HostPic(f.p)
        assert oracle().oh_or_frame(C.byref(f), host_pic_array(pics)) == 0
        got += md5s(pics[f.cur_pic])
    assert got == want


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_writer_is_deterministic_across_machines(path):
    """the stream a fixture was recorded from is what the writer produces here (the fixtures' streams are not stored)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(GOLD)))
    import make_stream_golden as M
    import streamgen
    name = os.path.basename(path)[:-4]
    case = [c for c in M.STREAM_CASES if c[0] == name][0]
    data, _ = streamgen.write_stream(case[1], case[2], case[3], **case[4])
    assert hashlib.md5(data).digest() == pictures_of(path)[2]


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_engine_reproduces_reference_output(path):
    from openhevc_amd.engine import Engine, remap_frame
    frames, want, _ = pictures_of(path)
    eng = Engine(0)
    ids, got = {}, b""
    for ff in frames:
        f = ff.frame
        for i in [f.cur_pic] + [f.ref_pics[k] for k in range(F.OH_MAX_REFS) if f.ref_pics[k] >= 0]:
            if i not in ids:
                ids[i] = eng.pic_alloc(f.p)
        eng.frame_submit(remap_frame(f, ids))
        eng.sync()
        got += md5s(eng.pic_download(ids[f.cur_pic], f.p))
    eng.close()
    assert got == want
