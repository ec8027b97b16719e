"""The C-ABI library loads on a CPU-only host and exports every symbol include/volseg_hip.h declares."""
import ctypes
import re
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (REPO / "include" / "volseg_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from volume_segmantics_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 35
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    assert set(_lib._SIGS) <= set(syms), sorted(set(_lib._SIGS) - set(syms))   # every bound symbol is declared in the header
    assert _lib.lib.vs_version() >= 100


def test_errors_are_reported_not_fatal():
    from volume_segmantics_amd import _lib
    h = ctypes.c_void_p()
    rc = _lib.lib.vs_unet_create(ctypes.byref(h), 0, 2, 1, 100, 64)   # 100 is not a multiple of 32
    assert rc == -1 and "multiples of 32" in _lib.last_error()
    assert _lib.lib.vs_unet_num_tensors(99) < 0 and "classes" in _lib.last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    import sys
    monkeypatch.setenv("VOLSEG_HIP_LIB", str(tmp_path / "nope.so"))
    saved = sys.modules.pop("volume_segmantics_amd._lib")
    try:
        try:
            importlib.import_module("volume_segmantics_amd._lib")
            raised = False
        except ImportError as e:
            raised = "no CPU fallback" in str(e)
        assert raised
    finally:
        sys.modules["volume_segmantics_amd._lib"] = saved


def test_engine_refuses_cpu_forward():
    import pytest
    import torch
    from volume_segmantics_amd.engine import VolSegUnet
    m = VolSegUnet(2)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 32, 32))
