import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd")); sys.path.insert(0, ROOT)
from lfvio import abi, synth
from lfvio.engine import Engine
from oracle import binding as ob
np.set_printoptions(linewidth=200, precision=6)
eng = Engine(0)
w = synth.make_window_with_prior(0, 300, lambda x, f: ob.optimize(x, f))[0]
res = {}
for blk in (0, 1):
    eng.set_block_solve(blk)
    eng.set_linw(0)
    eng.batch_reserve(2, w.N, w.M)
    eng.batch_upload(0, w); eng.batch_upload(1, w)
    res[blk] = eng.resident_pass(2, 0, w.N)
a, b = res[0], res[1]
print("q dense", a["q"]); print("q block", b["q"]); print("q rel", np.abs(a["q"] - b["q"]) / np.maximum(np.abs(a["q"]), 1e-300))
d = np.abs(a["gn_p"] - b["gn_p"]); print("gn_p max abs diff", d.max(), "at", d.argmax(), "vals", a["gn_p"][d.argmax()], b["gn_p"][d.argmax()])
print("gn_p rel per entry worst", (d / np.maximum(np.abs(a["gn_p"]), 1e-300)).max())
for cnt in (1, 2, 8):
    rr = {}
    for blk in (0, 1):
        eng.set_block_solve(blk)
        eng.batch_reserve(cnt, w.N, w.M)
        for k in range(cnt): eng.batch_upload(k, w)
        rr[blk] = eng.resident_pass(cnt, 0, w.N)
        print('   blk', blk, 'kernel', eng.solve_kernel(cnt), 'gn_p[:3]', rr[blk]['gn_p'][:3], 'sb', rr[blk]['gn_p'][73:76], rr[blk]['gn_p'][163:166])
    d = np.abs(rr[0]['gn_p'] - rr[1]['gn_p'])
    print('count', cnt, 'gn_p max abs diff', d.max(), 'of', np.abs(rr[0]['gn_p']).max(), 'q diff', np.abs(rr[0]['q']-rr[1]['q']).max())
import ctypes as C
buf = (C.c_longlong * 64)()
eng.lib.lfvio_debug_read_clocks.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
eng.lib.lfvio_debug_read_clocks(eng.ctx, buf)
d=list(buf)[:32]
print('stamps', [d[k+1]-d[k] for k in range(7)], 'tail', d[24]-d[5], d[25]-d[24], d[26]-d[25], d[6]-d[26])
print('chain per wave (wave0 step | col | syrk | barrier):', [d[8+4*w:12+4*w] for w in range(4)])
print('jtrace', np.frombuffer(buf, dtype=np.float64)[32:64])
for blk in (0, 1):
    eng.set_block_solve(blk)
    s = eng.solve(w)
    print("blk", blk, "iters", s.c.num_iterations, "term", s.c.termination)
    for t in s.trace():
        print("   ", {k: (f"{v:.9e}" if isinstance(v, float) else v) for k, v in t.items()})
