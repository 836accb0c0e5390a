"""CPU tests of the host side of k_linb (lf-vio_amd/csrc/linb_plan.h: the groups of a large window): the header is plain C++, compiled
here with g++ into a small shared object and driven through ctypes.  Properties, on synthetic track-length distributions from a few
hundred to two million landmarks:
  * the groups partition the landmarks: every landmark in exactly one group, a group inside one start frame, strips consecutive;
  * a group has one to eight strips (<= 512 landmarks); only the last group of a start frame may end on a partial strip;
  * at most 500 groups wherever eight strips per group allow it (up to ~250 000 landmarks), more only where they cannot; the most
    expensive group first;
  * long tracks get few strips while the budget is small (windows up to 100 000 landmarks): a group whose longest track has more than
    eight steps holds at most two strips there — a window of millions packs eight strips whatever their length and runs in rounds."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include "linb_plan.h"
extern "C" int plan(const int *begin_s, int num_frames, const int *lm_cnt, int block, int max_groups, int cap, int *lm0, int *n, int *s, int *cost) {
  const std::vector<LinbGroup> g = linb_plan_groups(begin_s, num_frames, lm_cnt, block, max_groups);
  if ((int)g.size() > cap) return -1;
  for (size_t k = 0; k < g.size(); k++) lm0[k] = g[k].lm0, n[k] = g[k].n, s[k] = g[k].s, cost[k] = g[k].cost;
  return (int)g.size();
}
'''


@pytest.fixture(scope="module")
def planner():
    d = tempfile.mkdtemp(prefix="linb_plan_")
    src, so = os.path.join(d, "plan.cpp"), os.path.join(d, "libplan.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "lf-vio_amd", "csrc"), src, "-o", so])
    lib = C.CDLL(so)
    ip = C.POINTER(C.c_int)
    lib.plan.argtypes = [ip, C.c_int, ip, C.c_int, C.c_int, C.c_int, ip, ip, ip, ip]

    def run(start, cnt, max_groups=500):
        order = np.lexsort((cnt, start))  # the upload's order: start frame, then track length
        start, cnt = np.ascontiguousarray(start[order], np.int32), np.ascontiguousarray(cnt[order], np.int32)
        begin = np.searchsorted(start, np.arange(12)).astype(np.int32)
        cap = len(start) // 64 + 16
        out = [np.zeros(cap, np.int32) for _ in range(4)]
        ng = lib.plan(begin.ctypes.data_as(ip), 11, cnt.ctypes.data_as(ip), 64, max_groups, cap, *[o.ctypes.data_as(ip) for o in out])
        assert ng >= 0
        return start, cnt, begin, [o[:ng] for o in out]

    return run


def tracks(rng, n, long_share=0.3):
    start = rng.integers(0, 10, n)
    room = 11 - start
    cnt = np.minimum(room, 2 + rng.geometric(1.0 - long_share, n) - 1)
    return start.astype(np.int32), np.maximum(cnt, 2).astype(np.int32)


@pytest.mark.parametrize("n,long_share", [(321, 0.3), (5000, 0.3), (70000, 0.5), (100000, 0.7), (250000, 0.3), (2000000, 0.5)])
def test_the_groups_partition_the_window(planner, n, long_share):
    rng = np.random.default_rng(n)
    start, cnt, begin, (lm0, ln, s, cost) = planner(*tracks(rng, n, long_share))
    seen = np.zeros(n, np.int32)
    for a, k, f in zip(lm0, ln, s):
        assert 1 <= k <= 512 and begin[f] <= a and a + k <= begin[f + 1]  # inside one start frame
        assert (a - begin[f]) % 64 == 0                                    # whole strips from the start frame's first landmark on
        assert k % 64 == 0 or a + k == begin[f + 1]                        # only the last group of a start frame ends on a partial strip
        seen[a:a + k] += 1
    assert (seen == 1).all()
    assert (np.diff(cost) <= 0).all()  # the most expensive first
    strips = sum((begin[f + 1] - begin[f] + 63) // 64 for f in range(11))
    assert len(lm0) <= max(500, (strips + 7) // 8 + 11), (len(lm0), strips)
    # long tracks: few strips per group (where the budget is what ~500 groups of this window need, not what 8 strips cost)
    for a, k in zip(lm0, ln) if n <= 100000 else ():
        steps = cnt[a + k - 1] - 1
        if steps > 8:
            assert k <= 128, (a, k, steps)


def test_a_window_that_fits_500_groups_gets_at_most_500(planner):
    rng = np.random.default_rng(7)
    for n in (40960, 100000, 200000):
        _, _, _, (lm0, ln, s, cost) = planner(*tracks(rng, n))
        assert len(lm0) <= 500, (n, len(lm0))
        assert cost[0] <= 3 * max(cost[-1], 100), (n, cost[0], cost[-1])  # none much longer than the others


def test_every_landmark_anchored_at_one_frame(planner):
    n = 30000
    start, cnt = np.zeros(n, np.int32), np.full(n, 11, np.int32)  # the longest tracks possible, one start frame
    _, _, _, (lm0, ln, s, cost) = planner(start, cnt)
    assert (s == 0).all() and ln.sum() == n and (ln <= 64).all()  # ten steps per strip: one strip per group, its steps split four ways
