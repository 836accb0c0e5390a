"""CPU test: the product library builds for gfx950, loads without a GPU and exports every symbol that
include/lfvio.h and include/lfvio_debug.h declare (no compute calls here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lfvio_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported():
    import __graft_entry__ as ge

    ge.build()
    from lfvio import abi

    lib = ctypes.CDLL(abi.HIP_LIB_PATH)
    names = declared("lfvio.h") + declared("lfvio_debug.h")
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
    # the Python-side list used by the loader is the header's list
    assert set(abi.HIP_SYMBOLS) == set(declared("lfvio.h"))


def test_create_fails_loudly_without_gpu():
    """No CPU fallback: on a machine without a HIP device lfvio_create returns NULL and the Engine raises."""
    import torch

    if torch.cuda.is_available():
        return  # GPU box: covered by the gpu suite
    import pytest
    from lfvio.engine import Engine

    with pytest.raises(RuntimeError):
        Engine(0)
