"""Landmark-sharded solve across the GPUs of one node.

One process per GPU.  Every rank holds the whole (small) pose-side state and a contiguous CSR range of
the landmarks balanced on observation count; per trust-region iteration the ranks exchange ONE
sum-all-reduce of [H_pp | g_p | Schur sums | scalars] (151 KB, latency-bound on xGMI) and two 128-byte
scalar all-reduces; the reduced 172x172 system is solved redundantly on every rank, so no broadcast is
needed and every rank takes the identical accept/reject decision.
"""
import numpy as np


def partition_landmarks(obs_offset, world):
    """Contiguous landmark ranges [b_r, e_r) with ~equal observation counts (the sweep cost is per
    observation).  Returns a list of (begin, end); ranges may be empty when N < world."""
    obs_offset = np.asarray(obs_offset, dtype=np.int64)
    N = len(obs_offset) - 1
    M = int(obs_offset[-1]) if N > 0 else 0
    cuts = [0]
    for r in range(1, world):
        target = M * r / world
        # first landmark whose start offset reaches the target, kept monotone
        c = int(np.searchsorted(obs_offset[:-1], target, side="left"))
        cuts.append(min(max(c, cuts[-1]), N))
    cuts.append(N)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


class _CudaArray:
    """Minimal __cuda_array_interface__ view of a raw device pointer (float64, 1-D)."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr="<f8", data=(int(ptr), False), version=3, strides=None)


def exchange_tensor(engine):
    """torch view (no copy) of the engine's device exchange buffer + the offset of its scalar tail."""
    import torch

    ptr, n, soff = engine.shard_exchange()
    t = torch.as_tensor(_CudaArray(ptr, n), device="cuda")
    return t, soff


def solve_sharded(engine, win, rank, world, all_reduce, marg_flag=None):
    """Run the sharded trust-region loop.  all_reduce(tensor) must sum `tensor` in place over the ranks
    (torch.distributed.all_reduce on the RCCL backend).  Returns (solution, (begin, end)) — inv_depth is filled for
    the local landmark range only — or, with marg_flag (abi.MARGIN_OLD / MARGIN_SECOND_NEW), the whole
    optimization(): (solution after the gauge fix, (begin, end), next prior); the marginalization costs one more
    all-reduce of the exchange buffer and every rank ends up with the identical prior."""
    ranges = partition_landmarks(win.obs_offset, world)
    b, e = ranges[rank]
    engine.shard_begin(win, b, e, add_pose_side=(rank == 0))
    buf, soff = exchange_tensor(engine)
    tail = buf[soff:]
    state, guard = 0, 0
    while state != 2:
        if engine.shard_phase("linearize") == 1:
            all_reduce(buf)
        if engine.shard_phase("solve") == 1:
            all_reduce(tail)
        if engine.shard_phase("candidate") == 1:
            all_reduce(tail)
        state = engine.shard_decide()
        guard += 1
        if guard > 4 * (win.max_num_iterations + 8):
            raise RuntimeError("sharded loop did not terminate")
    if marg_flag is None:
        return engine.shard_finish(win.N), (b, e)
    if engine.shard_marginalize_linearize(marg_flag) == 1:
        all_reduce(buf)
    prior = engine.shard_marginalize_finish(marg_flag)
    return engine.shard_finish(win.N), (b, e), prior
