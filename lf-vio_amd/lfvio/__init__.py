"""lfvio — Python plumbing over the C-ABI of the MI355X sliding-window solver.

`abi`   ctypes mirror of include/lfvio.h + loader of liblfvio_hip.so (no fallback)
`synth` seeded synthetic windows of the BASELINE.json shape
`engine` thin object wrapper over the C-ABI entry points
"""
from . import abi, synth  # noqa: F401
