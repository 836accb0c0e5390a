// solve_plan.h — static elimination plan of the reduced pose system (k_solve), shared by the kernel and by the CPU check
// of the plan (tests/tools/solve_plan_check.cpp: the same tables and address functions drive a plain-loop solve that is
// compared with a dense one).
//
// The reduced system Ceres factors with one dense Cholesky (schur_complement_solver.cc, DENSE_SCHUR: 172 x 172 here) is not
// dense: after the landmarks are gone only the camera-side block [11 poses | ex | td] (73) is full.  The eleven speed/bias
// blocks (9 each) form a chain — IMU factor k couples (pose_k, sb_k, pose_k+1, sb_k+1), estimator.cpp:717-724 — and the
// prior reaches sb_0 only (marginalization keeps the speed/bias of the new frame 0: estimator.cpp:833-1005).  So nine of
// the eleven blocks are eliminated first, by cyclic reduction over the chain — independent blocks side by side, one wave
// each, 9 pivots deep per round instead of 99 in a row:
//     round 1: sb 1 3 5 7      round 2: sb 9 0 4      round 3: sb 2 10
// and the dense tiled Cholesky is left with [camera 73 | sb_6 | sb_8] = 91 unknowns (6 tiles of 16 instead of 11): 27 + 91
// pivots on the critical path instead of 172, a third of the flops.  Any elimination order of an SPD system is a valid
// Cholesky; the result differs from the natural order by rounding only.
//
// Storage (LDS, doubles):
//   remainder  lower 16 x 16 tiles of a 96 x 96 matrix, tile (a, b) at tile_id(a, b) * TSZ2, entry (r, k) at 17 r + k;
//              index: camera column c -> c, sb_6 -> 73 + r, sb_8 -> 82 + r, the right-hand side is ROW 91
//   front f    (one per eliminated block, in elimination order) 9 rows x S2_LDX: [own 9 | nbA 9 | nbB 9 | camera 73 | rhs]
//              — the block row of the matrix at the moment the block is eliminated; after its factorization the rows hold
//              [L^T | X = L^-1 A_fN | L^-1 b_f].  An entry that couples two blocks lives in the front of the one
//              eliminated FIRST (or in the remainder when neither is).
#pragma once

#ifndef PLAN_HD
#define PLAN_HD __host__ __device__ constexpr inline
#endif

constexpr int S2_NF = 9;            // eliminated speed/bias blocks
constexpr int S2_LDX = 104;         // row stride of a front
constexpr int S2_FSZ = 9 * S2_LDX;  // doubles per front
constexpr int S2_COL_A = 9, S2_COL_B = 18, S2_COL_CAM = 27, S2_COL_RHS = 100, S2_NCOL = 101;
constexpr int S2_NR = 91;           // unknowns of the dense remainder; the rhs is row S2_NR
constexpr int S2_NT = 6;            // its tiles per dimension
constexpr int S2_NTILES = S2_NT * (S2_NT + 1) / 2;
constexpr int S2_TLD = 17, S2_TSZ = 16 * S2_TLD;
constexpr int S2_REM_SB6 = 73, S2_REM_SB8 = 82;
constexpr int S2_KC = 73, S2_KP = 172;

// fronts in elimination order
PLAN_HD int s2_block(int fi) {  // speed/bias block of front fi
  constexpr int t[S2_NF] = {1, 3, 5, 7, 9, 0, 4, 2, 10};
  return t[fi];
}
PLAN_HD int s2_front_of(int f) {  // front of speed/bias block f, -1: the block stays in the remainder
  constexpr int t[11] = {5, 0, 7, 1, 6, 2, -1, 3, -1, 4, 8};
  return t[f];
}
PLAN_HD int s2_nb(int fi, int slot) {  // speed/bias neighbours of front fi at its elimination (slot 0: nbA, 1: nbB), -1: none
  constexpr int a[S2_NF] = {0, 2, 4, 6, 8, 2, 2, 6, 8};
  constexpr int b[S2_NF] = {2, 4, 6, 8, 10, -1, 6, -1, -1};
  return slot ? b[fi] : a[fi];
}
PLAN_HD int s2_c0(int fi) {  // camera columns [c0, c1) the front touches at its elimination
  constexpr int t[S2_NF] = {0, 12, 24, 36, 48, 0, 12, 0, 48};
  return t[fi];
}
PLAN_HD int s2_c1(int fi) {
  constexpr int t[S2_NF] = {18, 30, 42, 54, 66, 73, 42, 73, 66};
  return t[fi];
}
// rounds: fronts [0, 4), [4, 7), [7, 9); the updates of a round are applied in sub-phases of fronts that touch disjoint
// targets (two at a time where that holds): {0, 2} {1, 3} | {4, 6} {5} | {7} {8}
constexpr int S2_NPHASE = 6;
PLAN_HD int s2_phase_front(int ph, int k) {  // k-th front of update sub-phase ph, -1: none
  constexpr int a[S2_NPHASE] = {0, 1, 4, 5, 7, 8};
  constexpr int b[S2_NPHASE] = {2, 3, 6, -1, -1, -1};
  return k ? b[ph] : a[ph];
}

PLAN_HD int s2_tile_id(int a, int b) { return a * (a + 1) / 2 + b; }
// remainder entry (i, j), j <= i
PLAN_HD int s2_lidx(int i, int j) { return s2_tile_id(i >> 4, j >> 4) * S2_TSZ + (i & 15) * S2_TLD + (j & 15); }
PLAN_HD int s2_rem_of_sb(int f) { return f == 6 ? S2_REM_SB6 : S2_REM_SB8; }  // f in {6, 8}

// ---- where an entry of the (scaled, regularized) reduced system is stored -----------------------------------------------
// Variables: tangent column v in [0, 172) (camera side [0, 73), then sb_f at 73 + 9 f), v = 172 for the right-hand side.
// Returns the offset in doubles from the start of the solve storage [remainder tiles | fronts]; `diag_mirror` (may be
// null) receives a second offset (or -1) for entries inside the own block of a front, which is kept as a full square.
constexpr int S2_REM_LEN = S2_NTILES * S2_TSZ;  // 5712
PLAN_HD int s2_front_base(int fi) { return S2_REM_LEN + fi * S2_FSZ; }
constexpr int S2_STORE_LEN = S2_REM_LEN + S2_NF * S2_FSZ;  // 14136 doubles
PLAN_HD int s2_rem_index(int v) {  // remainder index of a variable that is not in a front
  if (v < S2_KC) return v;
  if (v == S2_KP) return S2_NR;
  const int f = (v - S2_KC) / 9, r = (v - S2_KC) % 9;
  return s2_rem_of_sb(f) + r;
}
PLAN_HD int s2_front_col(int fi, int v) {  // column of variable v in the layout of front fi, -1: not a neighbour
  if (v == S2_KP) return S2_COL_RHS;
  if (v < S2_KC) return (v >= s2_c0(fi) && v < s2_c1(fi)) ? S2_COL_CAM + v : -1;
  const int f = (v - S2_KC) / 9, r = (v - S2_KC) % 9;
  if (f == s2_block(fi)) return r;
  if (f == s2_nb(fi, 0)) return S2_COL_A + r;
  if (f == s2_nb(fi, 1)) return S2_COL_B + r;
  return -1;
}
PLAN_HD int s2_store(int u, int v, int *diag_mirror) {
  if (diag_mirror) *diag_mirror = -1;
  const int fu = (u >= S2_KC && u < S2_KP) ? s2_front_of((u - S2_KC) / 9) : -1;
  const int fv = (v >= S2_KC && v < S2_KP) ? s2_front_of((v - S2_KC) / 9) : -1;
  if (fu < 0 && fv < 0) {
    const int i = s2_rem_index(u), j = s2_rem_index(v);
    return i >= j ? s2_lidx(i, j) : s2_lidx(j, i);
  }
  // the front eliminated first owns the entry (fronts are numbered in elimination order)
  int first = fu, other = v, own = u;
  if (fu < 0 || (fv >= 0 && fv < fu)) first = fv, other = u, own = v;
  const int row = (own - S2_KC) % 9;
  const int col = s2_front_col(first, other);
  if (col < 0) return -1;  // not in the structure the plan assumes
  if (fu == fv && diag_mirror) *diag_mirror = s2_front_base(first) + ((other - S2_KC) % 9) * S2_LDX + row;
  return s2_front_base(first) + row * S2_LDX + col;
}

// ---- the update a factored front applies to what is left: target -= sum_k X[k][p] X[k][q] ------------------------------
// as rectangular segments of tasks with affine source columns; targets either affine (inside another front) or remainder
// entries.  The camera x camera part (and rhs x camera) goes by tiles in the kernel (s2_c0 / s2_c1), not through segments.
struct S2Seg {
  int rows, cols;      // tasks: (r, c), r < rows, c < cols
  int src_r, src_c;    // source columns in the front's layout: src_r + r, src_c + c
  int kind;            // 0: target in a front: base + r * sr + c * sc;  1: remainder entry (i0 + r * ir + c * ic_i, j0 + ...), see below
  int base, sr, sc;    // kind 0
  int i0, j0;          // kind 1: the entry is (i0 + c, j0 + r) if swap else (i0 + r, j0 + c); lower triangle only when tri
  int swap, tri;
};
struct S2SegList {
  S2Seg s[8];
  int n;
};
// segments of front fi (<= 8)
PLAN_HD S2SegList s2_segment_list(int fi) {
  S2SegList L = {};
  S2Seg *out = L.s;
  int n = 0;
  const int A = s2_nb(fi, 0), B = s2_nb(fi, 1), c0 = s2_c0(fi), nc = s2_c1(fi) - s2_c0(fi);
  const int nbs[2] = {A, B};
  const int nbcol[2] = {S2_COL_A, S2_COL_B};
  for (int s = 0; s < 2; s++) {
    const int X = nbs[s];
    if (X < 0) continue;
    const int fx = s2_front_of(X);
    // diagonal block of X
    {
      S2Seg g = {9, 9, nbcol[s], nbcol[s], 0, 0, 0, 0, 0, 0, 0, 0};
      if (fx >= 0) g.kind = 0, g.base = s2_front_base(fx), g.sr = S2_LDX, g.sc = 1;  // full square
      else g.kind = 1, g.i0 = s2_rem_of_sb(X), g.j0 = s2_rem_of_sb(X), g.tri = 1;
      out[n++] = g;
    }
    // X x camera
    if (nc > 0) {
      S2Seg g = {9, nc, nbcol[s], S2_COL_CAM + c0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (fx >= 0) g.kind = 0, g.base = s2_front_base(fx) + S2_COL_CAM + c0, g.sr = S2_LDX, g.sc = 1;
      else g.kind = 1, g.i0 = s2_rem_of_sb(X), g.j0 = c0, g.swap = 0;  // entry (sb index, camera column): sb rows are below the camera rows
      out[n++] = g;
    }
    // X x rhs
    {
      S2Seg g = {9, 1, nbcol[s], S2_COL_RHS, 0, 0, 0, 0, 0, 0, 0, 0};
      if (fx >= 0) g.kind = 0, g.base = s2_front_base(fx) + S2_COL_RHS, g.sr = S2_LDX, g.sc = 1;
      else g.kind = 1, g.i0 = S2_NR, g.j0 = s2_rem_of_sb(X), g.swap = 1;  // entry (rhs row, sb index): (i0 + c, j0 + r)
      out[n++] = g;
    }
  }
  if (A >= 0 && B >= 0) {  // coupling of the two neighbours: rows = A index, cols = B index
    S2Seg g = {9, 9, S2_COL_A, S2_COL_B, 0, 0, 0, 0, 0, 0, 0, 0};
    const int fa = s2_front_of(A), fb = s2_front_of(B);
    if (fa < 0 && fb < 0) {
      // both stay: remainder entry (larger index, smaller index)
      const int ia = s2_rem_of_sb(A), ib = s2_rem_of_sb(B);
      g.kind = 1;
      if (ia > ib) g.i0 = ia, g.j0 = ib, g.swap = 0;
      else g.i0 = ib, g.j0 = ia, g.swap = 1;
    } else if (fb < 0 || (fa >= 0 && fa < fb)) {  // A's front owns it: row = A index, column = slot of B
      const int col = s2_front_col(fa, S2_KC + 9 * B);
      g.kind = 0, g.base = s2_front_base(fa) + col, g.sr = S2_LDX, g.sc = 1;
    } else {  // B's front owns it: row = B index, column = slot of A
      const int col = s2_front_col(fb, S2_KC + 9 * A);
      g.kind = 0, g.base = s2_front_base(fb) + col, g.sr = 1, g.sc = S2_LDX;
    }
    out[n++] = g;
  }
  L.n = n;
  return L;
}
PLAN_HD int s2_segments(int fi, S2Seg *out) {
  const S2SegList L = s2_segment_list(fi);
  for (int k = 0; k < L.n; k++) out[k] = L.s[k];
  return L.n;
}
