"""CPU-side checks of the C ABI: the built library exports every symbol the public headers
declare, and the product has no route into oracle/."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")


def declared(header, prefix):
    txt = open(os.path.join(INC, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", txt)))


def test_hip_library_exports_the_declared_abi():
    path = os.path.join(ROOT, "openhevc_amd", "libohevc_hip.so")
    if not os.path.exists(path):
        pytest.skip("libohevc_hip.so not built (run __graft_entry__.build())")
    lib = C.CDLL(path)                       # loading needs no GPU; no compute call is made
    names = declared("ohevc_hip.h", "oh_") + declared("ohevc_recorder.h", "oh_rec_")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ but not exported"


def test_host_library_exports_recorder_and_synth():
    from openhevc_amd import frame as F
    lib = F.host()
    for n in declared("ohevc_recorder.h", "oh_rec_") + declared("ohevc_synth.h", "oh_synth_"):
        assert hasattr(lib, n), n


def test_engine_fails_loudly_without_a_gpu():
    """no silent CPU fallback: without a HIP device engine creation must raise"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    path = os.path.join(ROOT, "openhevc_amd", "libohevc_hip.so")
    if not os.path.exists(path):
        pytest.skip("libohevc_hip.so not built")
    from openhevc_amd.engine import Engine, EngineError
    with pytest.raises(EngineError):
        Engine(0)


def test_product_never_references_the_oracle():
    pkg = os.path.join(ROOT, "openhevc_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".c", ".h", ".hip", ".cpp", "Makefile")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "liboracle" not in txt and "oracle/" not in txt and "oh_or_" not in txt, os.path.join(dirpath, fn)
