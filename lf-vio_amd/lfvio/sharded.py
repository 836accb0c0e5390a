"""Landmark-sharded solve across the GPUs of one node — the PYTHON reference driver.

The product's driver is C++ inside the library (lfvio_group, lf-vio_amd/csrc/group.inc: the loop below with the collective an
ncclAllReduce the library issues itself); bench.py and the host side use that one.  This module stays as the second,
independent driver the tests hold it against (torch.distributed / emulated collectives over the exposed exchange buffer).

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


class ShardedWindow:
    """One rank's share of a landmark-sharded window, driven stream-ordered: per pass seven enqueues (four kernel phases,
    three sum-all-reduces on the context's own stream) and ONE wait — for the decision of the pass before the one just
    enqueued, so the device never idles behind the host.  Every rank reads identical flags and therefore issues the
    identical sequence of collectives.

    all_reduce(tensor) must sum `tensor` in place over the ranks, ordered on the current torch stream (for
    torch.distributed on RCCL that is what dist.all_reduce does); world 1 may pass None.
    """

    def __init__(self, engine, win, rank, world, all_reduce):
        import torch

        self.eng, self.win, self.rank, self.world = engine, win, rank, world
        self.range = partition_landmarks(win.obs_offset, world)[rank]
        engine.shard_begin(win, self.range[0], self.range[1], add_pose_side=(rank == 0))
        self.buf, soff = exchange_tensor(engine)
        self.tail = self.buf[soff:]
        self.stream = torch.cuda.ExternalStream(engine.stream())
        self.all_reduce = all_reduce if (all_reduce is not None and world > 1) else (lambda t: None)
        self.passes = 0
        self._armed = True

    def _pass(self):
        e = self.eng
        e.shard_enqueue(0)
        self.all_reduce(self.buf)
        e.shard_enqueue(1)
        self.all_reduce(self.tail)
        e.shard_enqueue(2)
        self.all_reduce(self.tail)
        e.shard_enqueue(3)
        self.passes += 1

    def run(self, marg_flag=None):
        """The trust-region loop (and, with marg_flag, gauge fix + marginalization: one more all-reduce).  Returns
        (solution, (begin, end)) or (solution after the gauge fix, (begin, end), prior)."""
        import torch

        if not self._armed:
            self.eng.shard_restart()
        self._armed = False
        self.passes = 0
        limit = 4 * (self.win.max_num_iterations + 8)
        with torch.cuda.stream(self.stream):
            try:
                self._pass()
                while True:
                    self._pass()                  # one pass in flight ...
                    state = self.eng.shard_poll()  # ... behind the decision being read
                    if state == 2:
                        break
                    if self.passes > limit:
                        raise RuntimeError("sharded loop did not terminate")
            finally:
                # no record may stay outstanding, whatever ended the loop (the pass in flight behind a terminated loop was a
                # no-op; behind an error its record must still be consumed before the ring is reused)
                try:
                    while self.eng.shard_poll() is not None:
                        pass
                except RuntimeError:
                    pass
            if marg_flag is None:
                return self.eng.shard_finish(self.win.N), self.range
            if self.eng.shard_marginalize_linearize(marg_flag) == 1:
                self.all_reduce(self.buf)
            prior = self.eng.shard_marginalize_finish(marg_flag)
            return self.eng.shard_finish(self.win.N), self.range, prior


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
