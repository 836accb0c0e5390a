"""One-shot GPU diagnostics: HIP path vs oracle, verbose (used during bring-up)."""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
from oracle import binding as ob

def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)

def stage(name, fn):
    print(f"==== {name}", flush=True)
    try:
        fn()
    except Exception:
        traceback.print_exc()
    sys.stdout.flush()

eng = Engine(0)
print(eng.lib.lfvio_version())
N = int(os.environ.get("DBG_N", "300"))
w = synth.make_window(0, N)

def s_lin():
    A = eng.linearize(w); B = ob.linearize(w)
    print("cost", A["cost"], B["cost"])
    for k in ("H", "g", "a", "b", "W"):
        print(k, rel(A[k], B[k]))
    d = np.abs(A["H"] - B["H"]); i = np.unravel_index(d.argmax(), d.shape); print("worst H entry", i, A["H"][i], B["H"][i])
    # per-block summary
    blk = lambda M, r, c, n, m: np.abs(M[r:r+n, c:c+m]).max()
    print("Hcc diff", blk(d,0,0,73,73), "Hsb diff", blk(d,73,73,99,99), "Hc-sb diff", blk(d,73,0,99,73))
    dW = np.abs(A["W"] - B["W"]); print("W worst", np.unravel_index(dW.argmax(), dW.shape), dW.max())
stage("linearize", s_lin)

def s_solve():
    eng.set_graph(False)
    t = time.time(); so = eng.solve(w); print("gpu solve wall", time.time() - t)
    ro = ob.solve(w)
    print("iters", so.c.num_iterations, ro.c.num_iterations, "term", so.c.termination, ro.c.termination)
    for a, b in zip(so.trace(), ro.trace()):
        print("  gpu cost %.10g rad %.6g ok %d rd %.4g | ref cost %.10g rad %.6g ok %d rd %.4g" % (a["cost"], a["radius"], a["successful"], a["relative_decrease"], b["cost"], b["radius"], b["successful"], b["relative_decrease"]))
    print("pose diff", np.abs(so.pose - ro.pose).max(), "sb", np.abs(so.speed_bias - ro.speed_bias).max(), "lam rel", rel(so.lam, ro.lam), "td", so.td - ro.td)
    eng.set_graph(True)
    t = time.time(); s2 = eng.solve(w); print("graph solve wall", time.time() - t)
    t = time.time(); s2 = eng.solve(w); print("graph solve wall 2nd", time.time() - t)
    print("graph == direct", np.array_equal(s2.pose, so.pose), np.array_equal(s2.lam, so.lam))
stage("solve", s_solve)

def s_marg():
    sol, pr = ob.optimize(w, 0)
    w2 = abi.apply_solution(w, sol)
    ref, Ar, br = ob.marginalize(w2, 0, want_Ab=True)
    p = eng.marginalize(w2, 0)
    print("m n nb", p.m, p.n, p.num_blocks, "|", ref.m, ref.n, ref.num_blocks, p.block_list() == ref.block_list())
    A, b = eng.marg_system(p.n)
    print("A rel", rel(A, Ar), "b rel", rel(b, br))
    J, r = p.J(), p.r()
    print("JtJ-A", rel(J.T @ J, Ar), "Jtr-b", np.abs(J.T @ r - br).max() / np.abs(br).max())
    w3 = w2.copy(prior=ref)
    for flag in (0, 1):
        ref2, A2, b2 = ob.marginalize(w3, flag, want_Ab=True)
        p2 = eng.marginalize(w3, flag)
        if ref2.valid and p2.valid == 1 and p2.n == ref2.n:
            A, b = eng.marg_system(p2.n)
            print("flag", flag, "with prior: n", p2.n, ref2.n, "A rel", rel(A, A2), "b rel", rel(b, b2), p2.block_list() == ref2.block_list())
        else:
            print("flag", flag, "valid", p2.valid, ref2.valid, p2.n, ref2.n)
stage("marginalize", s_marg)

def s_opt():
    rs, rp = ob.optimize(w, 0)
    t = time.time(); s, p = eng.optimize(w, 0); print("optimize wall", time.time() - t)
    print("pose diff", np.abs(s.pose - rs.pose).max(), "lam rel", rel(s.lam, rs.lam), "prior blocks", p.block_list() == rp.block_list())
    J, Jr = p.J(), rp.J()
    print("JtJ rel", rel(J.T @ J, Jr.T @ Jr))
    t = time.time()
    for _ in range(20): eng.batch_optimize(1, 0)
    print("resident optimize avg ms", (time.time() - t) / 20 * 1e3)
stage("optimize", s_opt)
