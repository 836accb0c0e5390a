// linb_plan.h — host side of k_linb (kernels_linw.h): the GROUPS of a large window.  Plain C++ (no HIP): lfvio_hip.hip builds
// the plan at upload, tests/test_linb_plan.py compiles this file alone and checks its properties on the CPU.
//
// The landmarks arrive in device order: ascending start frame, ascending track length inside a start frame (upload_window's
// bucket sort).  A strip is up to 64 consecutive landmarks of one start frame; a group is one to eight consecutive strips of one
// start frame, the work of one workgroup:
//   1 strip   four waves share its steps (steps 1, 5, 9 / 2, 6, 10 / ...)
//   2 strips  two waves per strip
//   3, 4      a strip per wave
//   5 .. 8    two strips per wave (wave w: strips w and w + 4)
// sized by a cost model of the kernel's phases (half thousands of cycles, measured with two workgroups per CU: 36 fixed — zeroing,
// the sums out —, 13 per strip a wave takes, 18 per step of it, 22 per block of the Schur phase, 6 where waves share a strip): the
// steps of a strip are a serial chain, so long tracks get few strips per group.  The budget is the smallest one that leaves at
// most max_groups groups — all of them resident at once (two workgroups per CU), none much longer than the others; where even
// eight strips per group leave more (millions of landmarks) the launch runs in rounds.  The most expensive groups first.
#pragma once
#include <algorithm>
#include <vector>

struct LinbGroup {
  int lm0, n, s, cost;  // first landmark, landmarks (<= 512), start frame, modelled cost
};

// begin_s[s] .. begin_s[s + 1]: the landmarks of start frame s (num_frames + 1 entries); lm_cnt: track lengths in device order
inline std::vector<LinbGroup> linb_plan_groups(const int *begin_s, int num_frames, const int *lm_cnt, int block, int max_groups) {
  std::vector<LinbGroup> groups;
  auto build = [&](int budget) {
    groups.clear();
    for (int s = 0; s < num_frames; s++) {
      const int b0 = begin_s[s], b1 = begin_s[s + 1], nst = (b1 - b0 + block - 1) / block;
      auto steps_of = [&](int strip) { return lm_cnt[std::min(b0 + (std::min(strip, nst - 1) + 1) * block, b1) - 1] - 1; };  // its longest track
      auto cost_of = [&](int i, int nstr) {
        const int split = nstr <= 1 ? 4 : nstr <= 2 ? 2 : 1;
        const int wave = nstr <= 4 ? 13 + 18 * ((steps_of(i + nstr - 1) + split - 1) / split) : 26 + 18 * (steps_of(i + 3) + steps_of(i + nstr - 1));
        return 36 + wave + 22 * nstr + (split > 1 ? 6 : 0);
      };
      for (int i = 0; i < nst;) {
        int nstr = 1;
        for (int cand : {8, 7, 6, 5, 4, 3, 2})
          if (cand <= nst - i && cost_of(i, cand) <= budget) {
            nstr = cand;
            break;
          }
        const int l0 = b0 + i * block, n = std::min(nstr * block, b1 - l0);
        groups.push_back(LinbGroup{l0, n, s, cost_of(i, nstr)});
        i += nstr;
      }
    }
  };
  for (int budget = 100; budget <= 2000; budget += 10) {
    build(budget);
    if ((int)groups.size() <= max_groups) break;
  }
  std::stable_sort(groups.begin(), groups.end(), [](const LinbGroup &a, const LinbGroup &b) { return a.cost > b.cost; });
  return groups;
}
