"""Landmark-sharded (multi-GPU) path.

CPU (gloo, world_size 2): the partition, and the algebra the sharding rests on — per-rank Gauss-Newton
blocks of a landmark range sum, through a real all-reduce, to the blocks of the whole window, and the
reduced (Schur) system built from the reduced sums equals the unsharded one.
GPU (one device): two library contexts play two ranks on the same GPU, the all-reduce is emulated by
adding the two exchange buffers; the result must equal lfvio_solve() of the whole window."""
import os
import sys

import numpy as np
import pytest

from lfvio import abi, synth
from lfvio.sharded import partition_landmarks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sub_window(w, b, e, pose_side):
    o0, o1 = int(w.obs_offset[b]), int(w.obs_offset[e])
    sw = w.copy(start_frame=w.start_frame[b:e], obs_offset=w.obs_offset[b:e + 1] - o0, inv_depth=w.inv_depth[b:e],
                obs_point=w.obs_point[o0:o1], obs_velocity=w.obs_velocity[o0:o1], obs_cur_td=w.obs_cur_td[o0:o1],
                obs_uv_y=w.obs_uv_y[o0:o1])
    if not pose_side:  # IMU + prior live on one rank only
        imu = []
        for p in w.imu:
            q = abi.preint_from_array(abi.preint_to_array(p))
            q.sum_dt = 1e9
            imu.append(q)
        sw = sw.copy(imu=imu, prior=None)
    return sw


def test_partition_balances_observations():
    w = synth.make_window(1, 1000)
    for world in (1, 2, 3, 8):
        r = partition_landmarks(w.obs_offset, world)
        assert r[0][0] == 0 and r[-1][1] == w.N and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        loads = [int(w.obs_offset[e] - w.obs_offset[b]) for b, e in r]
        assert max(loads) - min(loads) <= 2 * 11  # each cut is off by at most one track (<= 11 observations)
    # fewer landmarks than ranks: empty ranges are legal
    r = partition_landmarks(synth.make_window(2, 3).obs_offset, 8)
    assert sum(e - b for b, e in r) == 3


def _gloo_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.make_window_with_prior(3, 200, lambda x, f: ob.optimize(x, f))[0]
    b, e = partition_landmarks(w.obs_offset, world)[rank]
    lin = ob.linearize(sub_window(w, b, e, pose_side=(rank == 0)))
    # the exchange payload: pose-side Hessian/gradient + Schur sums of the local landmarks + cost
    a = np.where(lin["a"] > 0, lin["a"], 1.0)
    schur = (lin["W"] / a[:, None]).T @ lin["W"]
    z = (lin["W"] / a[:, None]).T @ lin["b"]
    payload = torch.from_numpy(np.concatenate([lin["H"].ravel(), lin["g"], schur.ravel(), z, [lin["cost"]]]))
    dist.all_reduce(payload)  # SUM over ranks
    dist.barrier()
    if rank == 0:
        q.put(payload.numpy())
    dist.destroy_process_group()


def test_shards_sum_to_the_whole_window_gloo(oracle):
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = synth.make_window_with_prior(3, 200, lambda x, f: oracle.optimize(x, f))[0]
    lin = oracle.linearize(w)
    a = lin["a"]
    schur = (lin["W"] / a[:, None]).T @ lin["W"]
    z = (lin["W"] / a[:, None]).T @ lin["b"]
    want = np.concatenate([lin["H"].ravel(), lin["g"], schur.ravel(), z, [lin["cost"]]])
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    # and the reduced camera-side system assembled from the REDUCED sums equals the unsharded one
    H = got[:172 * 172].reshape(172, 172)
    S_red = H[:73, :73] - got[172 * 172 + 172: 172 * 172 + 172 + 73 * 73].reshape(73, 73)
    S_ref = lin["H"][:73, :73] - schur
    assert np.abs(S_red - S_ref).max() <= 1e-12 * np.abs(S_ref).max()


def _gloo_worker_camera_payload(rank, world, port, q):
    """round 5: what an lfvio_group rank sends and keeps (csrc/group.inc XR_SYSTEM, kernels_lin.h pose_terms_here,
    kernels_solve.h k_lm_cb2), restated with the oracle's linearization and a real all-reduce"""
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "lf-vio_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = synth.make_window_with_prior(3, 200, lambda x, f: ob.optimize(x, f))[0]
    b, e = partition_landmarks(w.obs_offset, world)[rank]
    KC = 73
    # EVERY rank evaluates the pose-side factors (IMU, prior) with its landmark range ...
    lin = ob.linearize(sub_window(w, b, e, pose_side=True))
    pose = ob.linearize(sub_window(w, 0, 0, pose_side=True))  # (... which alone are this)
    H, g = lin["H"].copy(), lin["g"].copy()
    if rank != 0:  # ... but only rank 0 adds them to the part that travels
        H[:KC, :KC] -= pose["H"][:KC, :KC]
        g[:KC] -= pose["g"][:KC]
    # Ceres' scaling and damping of the landmark blocks (Jacobi scale at this point, mu = 1e-4, no landmark on the clamp)
    mu = 1e-4
    a, bl, W = lin["a"], lin["b"], lin["W"]
    s = 1.0 / (1.0 + np.sqrt(a))
    d2 = np.clip(s * s * a, 1e-6, 1e32)
    assert np.all(d2 == s * s * a)
    el = s * s * a + mu * d2
    c = s * s / el
    schur, z1, cb2 = (W * c[:, None]).T @ W, (W * c[:, None]).T @ bl, float(np.sum(c * bl * bl))
    payload = torch.from_numpy(np.concatenate([H[:KC, :KC].ravel(), g[:KC], schur.ravel(), z1, [cb2]]))
    dist.all_reduce(payload)
    p = payload.numpy()
    Hc, gc = p[:KC * KC].reshape(KC, KC), p[KC * KC:KC * KC + KC]
    o = KC * KC + KC
    S_all, z_all, cb_all = p[o:o + KC * KC].reshape(KC, KC), p[o + KC * KC:o + KC * KC + KC], p[-1]
    # the system this rank solves: the reduced camera part, its OWN speed / bias rows
    Hfull, gfull = H.copy(), g.copy()
    Hfull[:KC, :KC], gfull[:KC] = Hc, gc
    # and the landmark parts of the Gauss-Newton step's norms for a camera direction N_c, from the reduced sums
    Nc = np.random.default_rng(5).normal(size=KC) * 1e-2
    lgn = (cb_all + 2.0 * z_all @ Nc + Nc @ S_all @ Nc) / (1.0 + mu)
    lgg = -(cb_all + z_all @ Nc)
    # ... against the same sums taken landmark by landmark over this rank's range, then all-reduced
    y = s * (bl + W @ Nc) / el
    gn = -np.sqrt(d2) * y
    direct = torch.tensor([float(np.sum(gn * gn)), float(np.sum((s * bl / np.sqrt(d2)) * gn))], dtype=torch.float64)
    dist.all_reduce(direct)
    dist.barrier()
    q.put((rank, Hfull, gfull, lgn, lgg, direct.numpy()))
    dist.destroy_process_group()


def test_camera_payload_and_landmark_norms_of_the_group_gloo(oracle):
    """What an lfvio_group rank all-reduces is the camera part only; the speed / bias rows are every rank's own copy; the dogleg's
    landmark norms come from the reduced Schur sums.  Two processes, a real (gloo) all-reduce: every rank ends
    with the unsharded system, and the quadratic forms equal the per-landmark sums."""
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker_camera_payload, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = synth.make_window_with_prior(3, 200, lambda x, f: oracle.optimize(x, f))[0]
    lin = oracle.linearize(w)
    for rank, Hfull, gfull, lgn, lgg, direct in got:
        assert np.abs(Hfull - lin["H"]).max() <= 1e-12 * np.abs(lin["H"]).max(), rank
        assert np.abs(gfull - lin["g"]).max() <= 1e-12 * np.abs(lin["g"]).max(), rank
        assert abs(lgn - direct[0]) <= 1e-9 * abs(direct[0]) and abs(lgg - direct[1]) <= 1e-9 * abs(direct[1]), (rank, lgn, lgg, direct)


@pytest.mark.gpu
# 20000: the two-level reductions under sharding; (8, 100000): BASELINE configs[3] at its full size on 8 emulated ranks
@pytest.mark.parametrize("world,n", [(2, 300), (3, 300), (2, 20000), (8, 100000)])
def test_two_contexts_emulate_ranks_on_one_gpu(oracle, world, n):
    import torch
    from lfvio.engine import Engine
    from lfvio.sharded import exchange_tensor

    ref = Engine(0)
    # the warm-up step that produces the prior: the oracle where its dense marginalization reaches, the GPU path beyond
    warm = (lambda x, f: oracle.optimize(x, f)) if n <= 1000 else (lambda x, f: ref.optimize(x, f))
    w = synth.make_window_with_prior(4, n, warm)[0]
    want = ref.solve(w)
    want_opt, want_prior = ref.optimize(w, abi.MARGIN_OLD)  # solve -> gauge fix -> marginalization on one GPU
    ref.close()
    engs = [Engine(0) for _ in range(world)]
    ranges = partition_landmarks(w.obs_offset, world)
    for r, eng in enumerate(engs):
        eng.shard_begin(w, ranges[r][0], ranges[r][1], add_pose_side=(r == 0))
    bufs = [exchange_tensor(e) for e in engs]

    def all_reduce(tensors):  # emulated sum-all-reduce over the "ranks"
        torch.cuda.synchronize()
        total = torch.stack(tensors).sum(dim=0)
        for t in tensors:
            t.copy_(total)
        torch.cuda.synchronize()

    states, guard = [0] * world, 0
    while any(s != 2 for s in states):
        rcs = [e.shard_phase("linearize") for e in engs]
        assert len(set(rcs)) == 1
        if rcs[0] == 1:
            all_reduce([b for b, _ in bufs])
        rcs = [e.shard_phase("solve") for e in engs]
        if rcs[0] == 1:
            all_reduce([b[o:] for b, o in bufs])
        rcs = [e.shard_phase("candidate") for e in engs]
        if rcs[0] == 1:
            all_reduce([b[o:] for b, o in bufs])
        states = [e.shard_decide() for e in engs]
        assert len(set(states)) == 1  # identical decision on every rank
        guard += 1
        assert guard < 64
    # ---- marginalization of the sharded window: one more all-reduce, identical prior on every rank
    rcs = [e.shard_marginalize_linearize(abi.MARGIN_OLD) for e in engs]
    assert rcs == [1] * world
    all_reduce([b for b, _ in bufs])
    priors = [e.shard_marginalize_finish(abi.MARGIN_OLD) for e in engs]
    for p in priors:
        assert (p.valid, p.m, p.n, p.num_blocks) == (1, want_prior.m, want_prior.n, want_prior.num_blocks)
        assert p.block_list() == want_prior.block_list()
        assert np.array_equal(p.J(), priors[0].J()) and np.array_equal(p.r(), priors[0].r())
    J, r, Jw, rw = priors[0].J(), priors[0].r(), want_prior.J(), want_prior.r()
    Aw = Jw.T @ Jw
    assert np.abs(J.T @ J - Aw).max() < 1e-6 * np.abs(Aw).max()
    assert np.abs(J.T @ r - Jw.T @ rw).max() < 1e-6 * np.abs(Jw.T @ rw).max()
    for i in range(priors[0].num_blocks):
        assert np.abs(priors[0].x0(i) - want_prior.x0(i)).max() < 1e-6
    sols = [e.shard_finish(w.N) for e in engs]  # the state after the gauge fix now
    assert np.abs(sols[0].pose - want_opt.pose).max() < 1e-6 * max(1.0, np.abs(want_opt.pose).max())
    want = want_opt
    lam = np.zeros(w.N)
    for r, s in enumerate(sols):
        b, e = ranges[r]
        lam[b:e] = s.inv_depth[b:e]
        assert np.array_equal(s.pose, sols[0].pose)  # replicated state is bit-identical across ranks
        assert s.c.num_iterations == want.c.num_iterations
    # summation order differs from the single-GPU path (per-rank partial sums): tolerance, not bits
    assert np.abs(sols[0].pose - want.pose).max() < 1e-6 * max(1.0, np.abs(want.pose).max())
    assert np.abs(lam - want.lam).max() < 1e-6 * np.abs(want.lam).max()
    assert abs(sols[0].c.final_cost - want.c.final_cost) <= 1e-7 * want.c.final_cost
    for e in engs:
        e.close()


# ---------------------------------------------------------------------------
# The stream-ordered driver (lfvio.sharded.ShardedWindow): what bench.py --gpus N runs over RCCL
# ---------------------------------------------------------------------------
class _ThreadAllReduce:
    """Sum-all-reduce between `world` threads of one process (one library context and stream each): the stand-in for
    RCCL when the ranks share a GPU.  Fixed summation order, the same on every rank."""

    def __init__(self, world):
        import threading

        self.world, self.bar, self.slots = world, threading.Barrier(world), [None] * world

    def rank(self, r):
        import torch

        def all_reduce(t):
            torch.cuda.current_stream().synchronize()  # this rank's exchange buffer is complete
            self.slots[r] = t
            self.bar.wait()
            if r == 0:
                total = torch.stack(self.slots).sum(dim=0)
                for s in self.slots:
                    s.copy_(total)
                torch.cuda.synchronize()
            self.bar.wait()

        return all_reduce


def _check_sharded_result(w, ranges, results, want, want_prior):
    sols = [r[0] for r in results]
    priors = [r[2] for r in results]
    for p in priors:
        assert (p.valid, p.m, p.n, p.num_blocks) == (1, want_prior.m, want_prior.n, want_prior.num_blocks)
        assert p.block_list() == want_prior.block_list()
        assert np.array_equal(p.J(), priors[0].J()) and np.array_equal(p.r(), priors[0].r())  # identical on every rank
    J, Jw = priors[0].J(), want_prior.J()
    Aw = Jw.T @ Jw
    assert np.abs(J.T @ J - Aw).max() < 1e-6 * np.abs(Aw).max()
    lam = np.zeros(w.N)
    for r, s in enumerate(sols):
        b, e = ranges[r]
        lam[b:e] = s.inv_depth[b:e]
        assert np.array_equal(s.pose, sols[0].pose) and np.array_equal(s.speed_bias, sols[0].speed_bias)
        assert (s.c.num_iterations, s.c.termination) == (want.c.num_iterations, want.c.termination)
    assert np.abs(sols[0].pose - want.pose).max() < 1e-6 * max(1.0, np.abs(want.pose).max())
    assert np.abs(sols[0].speed_bias - want.speed_bias).max() < 1e-6
    assert np.abs(lam - want.lam).max() < 1e-6 * np.abs(want.lam).max()
    assert abs(sols[0].c.final_cost - want.c.final_cost) <= 1e-7 * want.c.final_cost


@pytest.mark.gpu
@pytest.mark.parametrize("world,n", [(2, 300), (4, 3000), (8, 100000)])  # (8, 100000): BASELINE configs[3] at full size
def test_stream_ordered_driver_on_emulated_ranks(oracle, world, n):
    """`world` threads, one context each on the same GPU, run ShardedWindow.run() — seven enqueues per pass, one pass in
    flight behind the decision being read — twice (the second time from the resident shard, as the bench does); the
    result equals the single-GPU optimization() of the whole window."""
    import threading

    from lfvio.engine import Engine
    from lfvio.sharded import ShardedWindow

    ref = Engine(0)
    warm = (lambda x, f: oracle.optimize(x, f)) if n <= 1000 else (lambda x, f: ref.optimize(x, f))
    w = synth.make_window_with_prior(4, n, warm)[0]
    want, want_prior = ref.optimize(w, abi.MARGIN_OLD)
    ref.close()
    ar = _ThreadAllReduce(world)
    engs = [Engine(0) for _ in range(world)]
    results, errors = [[None, None] for _ in range(world)], []

    def work(r):
        try:
            sw = ShardedWindow(engs[r], w, r, world, ar.rank(r))
            results[r][0] = sw.run(abi.MARGIN_OLD)
            results[r][1] = sw.run(abi.MARGIN_OLD)  # shard_restart(): no upload
        except Exception as ex:  # noqa: BLE001
            errors.append((r, repr(ex)))
            ar.bar.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=600)
    assert not errors, errors
    ranges = partition_landmarks(w.obs_offset, world)
    for k in (0, 1):
        _check_sharded_result(w, ranges, [results[r][k] for r in range(world)], want, want_prior)
    for r in range(world):  # the resident re-run is the same computation: bit-identical
        assert np.array_equal(results[r][0][0].pose, results[r][1][0].pose)
        assert np.array_equal(results[r][0][2].J(), results[r][1][2].J())
    for e in engs:
        e.close()


def _two_process_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    for p in (ROOT, os.path.join(ROOT, "lf-vio_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from lfvio.engine import Engine
    from lfvio.sharded import ShardedWindow
    from oracle import binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    w = synth.make_window_with_prior(3, 300, lambda x, f: ob.optimize(x, f))[0]

    def all_reduce(t):  # gloo over a host staging copy; ordered on the current (= the context's) stream
        h = t.cpu()
        dist.all_reduce(h)
        t.copy_(h)

    eng = Engine(0)
    sol, rng, prior = ShardedWindow(eng, w, rank, world, all_reduce).run(abi.MARGIN_OLD)
    q.put((rank, rng, sol.pose, sol.speed_bias, sol.inv_depth.copy(), sol.c.num_iterations, sol.c.termination, sol.c.final_cost,
           prior.J(), prior.r(), prior.block_list()))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_processes_drive_the_sharded_solve_over_gloo(eng, oracle):
    """Two REAL processes (one library context each, both on GPU 0) run the product driver with torch.distributed
    collectives — gloo on a host staging copy here, RCCL on the device buffer in bench.py; the result must equal the
    single-process optimization() of the whole window and be bit-identical on both ranks."""
    import torch.multiprocessing as mp

    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_two_process_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in range(world)], key=lambda g: g[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    w = synth.make_window_with_prior(3, 300, lambda x, f: oracle.optimize(x, f))[0]
    want, want_prior = eng.optimize(w, abi.MARGIN_OLD)
    ref, ref_prior = oracle.optimize(w, abi.MARGIN_OLD)
    lam = np.zeros(w.N)
    for rank, (b, e), pose, sb, inv_depth, iters, term, cost, J, r, blocks in got:
        lam[b:e] = inv_depth[b:e]
        assert np.array_equal(pose, got[0][2]) and np.array_equal(J, got[0][8]) and np.array_equal(r, got[0][9])
        assert (iters, term) == (want.c.num_iterations, want.c.termination) == (ref.c.num_iterations, ref.c.termination)
        assert blocks == want_prior.block_list() == ref_prior.block_list()
        assert np.abs(pose - want.pose).max() < 1e-6 and np.abs(pose - ref.pose).max() < 1e-6
        assert abs(cost - ref.c.final_cost) <= 1e-7 * ref.c.final_cost
    assert np.abs(lam - ref.lam).max() < 1e-6 * np.abs(ref.lam).max()
    A, Aref = got[0][8].T @ got[0][8], ref_prior.J().T @ ref_prior.J()
    assert np.abs(A - Aref).max() < 1e-6 * np.abs(Aref).max()
