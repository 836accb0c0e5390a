// solve_plan_check.cpp — CPU check of lf-vio_amd/csrc/solve_plan.h (test infrastructure): the elimination plan of the
// reduced pose system — storage addresses, front layouts, update segments, sub-phase order — driven by plain loops.
// tests/test_solve_plan.py compares the result with a dense solve.  The kernel (kernels_solve2.h) uses the same header.
#include <cmath>
#include <cstring>
#include <vector>
#define PLAN_HD constexpr inline
#include "../../lf-vio_amd/csrc/solve_plan.h"

extern "C" int s2_reference_solve(const double *M, const double *rhs, double *y, double *max_unplaced) {
  std::vector<double> st(S2_STORE_LEN, 0.0);
  double unplaced = 0.0;
  // build: every entry of the lower triangle (and the rhs) at its storage address
  for (int i = 0; i < S2_KP; i++)
    for (int j = 0; j <= i; j++) {
      const double v = M[i * S2_KP + j];
      int mirror = -1;
      const int a = s2_store(i, j, &mirror);
      if (a < 0) {
        unplaced = std::fmax(unplaced, std::fabs(v));
        continue;
      }
      st[a] = v;
      if (mirror >= 0) st[mirror] = v;
    }
  for (int j = 0; j < S2_KP; j++) {
    const int a = s2_store(S2_KP, j, nullptr);
    if (a < 0) return -1;
    st[a] = rhs[j];
  }
  *max_unplaced = unplaced;
  auto factor = [&](int fi) -> int {
    double *X = st.data() + s2_front_base(fi);
    for (int k = 0; k < 9; k++) {
      const double d = X[k * S2_LDX + k];
      if (!(d > 0.0)) return -2;
      const double rs = 1.0 / std::sqrt(d);
      double m[9];
      for (int i = k + 1; i < 9; i++) m[i] = X[i * S2_LDX + k] * rs;
      for (int c = 0; c < S2_NCOL; c++) X[k * S2_LDX + c] *= rs;
      for (int i = k + 1; i < 9; i++)
        for (int c = 0; c < S2_NCOL; c++) X[i * S2_LDX + c] -= m[i] * X[k * S2_LDX + c];
    }
    return 0;
  };
  auto apply = [&](int fi) {
    const double *X = st.data() + s2_front_base(fi);
    S2Seg seg[8];
    const int ns = s2_segments(fi, seg);
    for (int s = 0; s < ns; s++) {
      const S2Seg &g = seg[s];
      for (int r = 0; r < g.rows; r++)
        for (int c = 0; c < g.cols; c++) {
          if (g.kind == 1 && g.tri && c > r) continue;
          double acc = 0.0;
          for (int k = 0; k < 9; k++) acc += X[k * S2_LDX + g.src_r + r] * X[k * S2_LDX + g.src_c + c];
          int a;
          if (g.kind == 0) a = g.base + r * g.sr + c * g.sc;
          else a = g.swap ? s2_lidx(g.i0 + c, g.j0 + r) : s2_lidx(g.i0 + r, g.j0 + c);
          st[a] -= acc;
        }
    }
    // camera x camera and rhs x camera: by tiles in the kernel
    const int c0 = s2_c0(fi), c1 = s2_c1(fi);
    for (int ci = c0; ci < c1; ci++) {
      for (int cj = c0; cj <= ci; cj++) {
        double acc = 0.0;
        for (int k = 0; k < 9; k++) acc += X[k * S2_LDX + S2_COL_CAM + ci] * X[k * S2_LDX + S2_COL_CAM + cj];
        st[s2_lidx(ci, cj)] -= acc;
      }
      double acc = 0.0;
      for (int k = 0; k < 9; k++) acc += X[k * S2_LDX + S2_COL_RHS] * X[k * S2_LDX + S2_COL_CAM + ci];
      st[s2_lidx(S2_NR, ci)] -= acc;
    }
  };
  const int round_begin[4] = {0, 4, 7, 9};
  int ph = 0;
  for (int r = 0; r < 3; r++) {
    for (int fi = round_begin[r]; fi < round_begin[r + 1]; fi++)
      if (int rc = factor(fi)) return rc;
    // the sub-phases of this round
    int done = 0;
    const int want = round_begin[r + 1] - round_begin[r];
    while (done < want) {
      for (int k = 0; k < 2; k++) {
        const int fi = s2_phase_front(ph, k);
        if (fi >= 0) {
          if (fi < round_begin[r] || fi >= round_begin[r + 1]) return -3;
          apply(fi), done++;
        }
      }
      ph++;
    }
  }
  if (ph != S2_NPHASE) return -4;
  // dense remainder: Cholesky of the 91 x 91 block with the rhs row riding along (row S2_NR)
  double *R = st.data();
  for (int k = 0; k < S2_NR; k++) {
    const double d = R[s2_lidx(k, k)];
    if (!(d > 0.0)) return -5;
    const double l = std::sqrt(d);
    R[s2_lidx(k, k)] = l;
    for (int i = k + 1; i <= S2_NR; i++) R[s2_lidx(i, k)] /= l;
    for (int i = k + 1; i <= S2_NR; i++)
      for (int j = k + 1; j <= i && j < S2_NR; j++) R[s2_lidx(i, j)] -= R[s2_lidx(i, k)] * R[s2_lidx(j, k)];
  }
  std::vector<double> yr(S2_NR);
  for (int k = S2_NR - 1; k >= 0; k--) {
    double t = R[s2_lidx(S2_NR, k)];
    for (int i = k + 1; i < S2_NR; i++) t -= R[s2_lidx(i, k)] * yr[i];
    yr[k] = t / R[s2_lidx(k, k)];
  }
  for (int c = 0; c < S2_KC; c++) y[c] = yr[c];
  for (int r = 0; r < 9; r++) y[S2_KC + 9 * 6 + r] = yr[S2_REM_SB6 + r], y[S2_KC + 9 * 8 + r] = yr[S2_REM_SB8 + r];
  // fronts, last eliminated first:  L^T y_f = z_f - X y_neighbours
  for (int fi = S2_NF - 1; fi >= 0; fi--) {
    const double *X = st.data() + s2_front_base(fi);
    const int f = s2_block(fi);
    double t[9];
    for (int k = 0; k < 9; k++) {
      double a = X[k * S2_LDX + S2_COL_RHS];
      for (int s = 0; s < 2; s++) {
        const int nb = s2_nb(fi, s);
        if (nb < 0) continue;
        for (int r = 0; r < 9; r++) a -= X[k * S2_LDX + (s ? S2_COL_B : S2_COL_A) + r] * y[S2_KC + 9 * nb + r];
      }
      for (int c = s2_c0(fi); c < s2_c1(fi); c++) a -= X[k * S2_LDX + S2_COL_CAM + c] * y[c];
      t[k] = a;
    }
    for (int k = 8; k >= 0; k--) {
      double a = t[k];
      for (int j = k + 1; j < 9; j++) a -= X[k * S2_LDX + j] * y[S2_KC + 9 * f + j];
      y[S2_KC + 9 * f + k] = a / X[k * S2_LDX + k];
    }
  }
  return 0;
}
