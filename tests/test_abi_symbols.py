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


def test_missing_rccl_is_a_clean_error():
    """with no loadable librccl the group entry points return an error / NULL — they used to build the message
    from two dlerror() calls (the first clears the error, the second returns NULL: std::string + NULL).  The loader keeps
    its result for the life of the process, so this runs in a process of its own.  LFVIO_RCCL_LIB is the only candidate
    when it is set (include/lfvio.h), which hides the system's copies."""
    import subprocess
    import sys

    from lfvio import abi

    code = (
        "import ctypes, sys\n"
        f"lib = ctypes.CDLL({abi.HIP_LIB_PATH!r})\n"
        "buf = ctypes.create_string_buffer(128)\n"
        "rc = lib.lfvio_group_unique_id(buf)\n"
        "assert rc < 0, rc\n"
        "lib.lfvio_group_create.restype = ctypes.c_void_p\n"
        "lib.lfvio_group_create_rank.restype = ctypes.c_void_p\n"
        "assert not lib.lfvio_group_create(ctypes.c_uint(1))\n"
        "assert not lib.lfvio_group_create_rank(0, 0, 1, buf)\n"
        "print('clean')\n"
    )
    env = dict(os.environ, LFVIO_RCCL_LIB="/nonexistent/librccl.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "clean" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-500:])
