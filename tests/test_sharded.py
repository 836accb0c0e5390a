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
    assert np.abs(J.T @ r - Jw.T @ rw).max() < 1e-4 * np.abs(Jw.T @ rw).max()
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
