"""Measures the quantities whose bars in tests/test_gpu_parity.py are wider than 1e-6 (run on the GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "lf-vio_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
from lfvio import abi, synth
from lfvio.engine import Engine
from oracle import binding as ob
import marg_ref

eng = Engine(0)
rel = lambda a, b: np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)

def prior_metrics(p, pr):
    J, r, Jr, rr = p.J(), p.r(), pr.J(), pr.r()
    A, Ar = J.T @ J, Jr.T @ Jr
    return dict(A=rel(A, Ar), Jtr=np.abs(J.T @ r - Jr.T @ rr).max() / np.abs(Jr.T @ rr).max(), kept=(marg_ref.kept_directions(p), marg_ref.kept_directions(pr)),
                slack=marg_ref.kept_count_slack(Ar, A))

print("== full chain")
win, warm = synth.make_window_with_prior(0, 300, lambda w, f: ob.optimize(w, f))
for w in (warm, win):
    rs, rp = ob.optimize(w, abi.MARGIN_OLD); gs, gp = eng.optimize(w, abi.MARGIN_OLD)
    print(" ", rel(gs.lam, rs.lam), prior_metrics(gp, rp))
print("== marginalize vs oracle on identical inputs")
for seed, n in [(0, 300), (7, 60), (8, 1000)]:
    w = synth.make_window(seed, n); sol, _ = ob.optimize(w, abi.MARGIN_OLD); w2 = abi.apply_solution(w, sol)
    ref, Aref, bref = ob.marginalize(w2, abi.MARGIN_OLD, want_Ab=True); p = eng.marginalize(w2, abi.MARGIN_OLD)
    A, b = eng.marg_system(p.n)
    J, r = p.J(), p.r()
    s, V = np.linalg.eigh(0.5 * (Aref + Aref.T))
    keep = s > 1e-8
    proj_b = V[:, keep] @ (V[:, keep].T @ bref)
    print(" ", n, "A'", rel(A, Aref), "b'", np.abs(b - bref).max() / np.abs(bref).max(), "JtJ", rel(J.T @ J, Aref), "Jtr vs b'", np.abs(J.T @ r - bref).max() / np.abs(bref).max(),
          "Jtr vs P b'", np.abs(J.T @ r - proj_b).max() / np.abs(bref).max(), prior_metrics(p, ref))
print("== randomized sweep")
rng = np.random.default_rng(20260928)
worst = []
for case in range(60):
    seed = int(rng.integers(0, 10_000)); n = int(rng.choice([1, 2, 5, 9, 17, 33, 64, 65, 128, 300]))
    kw = dict(estimate_extrinsic=int(rng.integers(0, 2)), estimate_td=int(rng.integers(0, 2)), tr=float(rng.choice([0.0, 0.02])), max_num_iterations=int(rng.choice([1, 3, 8, 12])))
    flag = int(rng.choice([abi.MARGIN_OLD, abi.MARGIN_SECOND_NEW]))
    if rng.integers(0, 2):
        w = synth.make_window_with_prior(seed, n, lambda x, f: ob.optimize(x, f), **kw)[0]
    else:
        w = synth.make_window(seed, n, **kw)
    rs, rp = ob.optimize(w, flag); gs, gp = eng.optimize(w, flag)
    dl = rel(gs.lam, rs.lam)
    dA = None
    if rp.valid == 1:
        Ar = rp.J().T @ rp.J()
        dA = rel(gp.J().T @ gp.J(), Ar) if np.abs(Ar).max() > 1.0 else None
    if dl > 3e-7 or (dA is not None and dA > 3e-7):
        lin = ob.linearize(abi.apply_solution(w, rs))
        # condition number of the Gauss-Newton Hessian at the solution (Jacobi-scaled), landmarks eliminated
        H, a, W = lin["H"], lin["a"], lin["W"]
        S = H.copy(); S[:73, :73] -= (W / a[:, None]).T @ W
        d = 1.0 / (1.0 + np.sqrt(np.maximum(np.diag(H), 0))); Ss = S * d[:, None] * d[None, :]
        ev = np.abs(np.linalg.eigvalsh(0.5 * (Ss + Ss.T)))
        print(f"  case {case} n={n} {kw} flag={flag} prior_in={w.prior is not None}: lam {dl:.2e} A {dA} min a {a.min():.2e} eig(S) [{ev.min():.1e}, {ev.max():.1e}] pose {np.abs(gs.pose-rs.pose).max():.1e}")
print("== sequence")
seed, n, steps = 6, 200, 6
scene = synth.Scene(seed, n_total=11 + steps); rng = np.random.default_rng([seed, 104729]); prior, st = None, None
for k in range(steps):
    kw = {} if k == 0 else dict(prior=prior, init_state=st)
    w = synth.make_window(seed, n, kf0=k, scene=scene, **kw)
    sol, prior = ob.optimize(w, abi.MARGIN_OLD)
    sg, pg = eng.optimize(w, abi.MARGIN_OLD)
    print(" ", k, prior_metrics(pg, prior))
    st = synth.continue_state(scene, k + 1, sol.pose, sol.speed_bias, sol.ex_pose, sol.td, rng)
