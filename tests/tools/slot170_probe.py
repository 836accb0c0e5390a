"""Which A' is off on the batch512 window of seed 170: the device's structured elimination or the dense eigen route? (GPU box)"""
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
rel = lambda a, b: np.abs(a - b).max() / np.abs(b).max()
for seed in (170, 3):
    w = synth.make_window_with_prior(seed, 300, lambda x, f: ob.optimize(x, f))[0]
    sol, _ = ob.optimize(w, abi.MARGIN_OLD)
    w2 = abi.apply_solution(w, sol)
    pref, Aref, bref = ob.marginalize(w2, abi.MARGIN_OLD, want_Ab=True)
    p = eng.marginalize(w2, abi.MARGIN_OLD)
    A_gpu, _ = eng.marg_system(p.n)
    eng.force_eig(True); p2 = eng.marginalize(w2, abi.MARGIN_OLD); A_gpu_eig, _ = eng.marg_system(p2.n); eng.force_eig(False)
    lin = ob.linearize(marg_ref.frame0_subwindow(w2))
    A_np, b_np, s, kept = marg_ref.structured_marg_old(lin, pref.block_list())
    # dense route in numpy: full m x m block with the landmarks, eigh + eps cut (marginalization_factor.cpp:267-272)
    a, b, W, H, g = lin["a"], lin["b"], lin["W"], lin["H"], lin["g"]
    drop = marg_ref.tangent_cols(abi.BLOCK_POSE, 0) + marg_ref.tangent_cols(abi.BLOCK_SPEEDBIAS, 0)
    keep = []
    for kind, frame, idx in pref.block_list():
        keep += marg_ref.tangent_cols(kind, frame + 1 if kind in (0, 1) else frame)
    N0 = len(a)
    m = 15 + N0
    Wk = np.zeros((N0, 172)); Wk[:, :73] = W
    Amm = np.zeros((m, m)); Amm[:15, :15] = H[np.ix_(drop, drop)]; Amm[:15, 15:] = Wk[:, drop].T; Amm[15:, :15] = Wk[:, drop]; Amm[15:, 15:] = np.diag(a)
    Arm = np.concatenate([H[np.ix_(keep, drop)], Wk[:, keep].T], axis=1)
    lam, V = np.linalg.eigh(0.5 * (Amm + Amm.T))
    inv = (V * np.where(lam > 1e-8, 1.0 / np.where(lam > 1e-8, lam, 1), 0)) @ V.T
    A_dense = H[np.ix_(keep, keep)] - Arm @ inv @ Arm.T
    print(f"seed {seed}: N0 {N0} a in [{a.min():.2e}, {a.max():.2e}] eig(A_mm full) [{lam.min():.2e}, {lam.max():.2e}]")
    print("   gpu(chol) vs oracle", rel(A_gpu, Aref), " gpu(eig) vs oracle", rel(A_gpu_eig, Aref), " gpu chol vs eig", rel(A_gpu, A_gpu_eig))
    print("   numpy structured vs oracle", rel(A_np, Aref), " vs gpu", rel(A_np, A_gpu))
    print("   numpy dense(eigh m x m) vs oracle", rel(A_dense, Aref), " vs gpu", rel(A_dense, A_gpu), " vs numpy structured", rel(A_dense, A_np))
    print("   |A'| max", np.abs(Aref).max(), " |A_rr| max", np.abs(H[np.ix_(keep, keep)]).max())
