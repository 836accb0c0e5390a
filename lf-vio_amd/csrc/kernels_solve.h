// kernels_solve.h — reduced pose system (assembly, Jacobi scaling, Cauchy point, Schur
// complement, dense Cholesky, back-substitution), landmark back-substitution, dogleg,
// candidate cost sweep and the Ceres accept/reject bookkeeping.
//
// Semantics follow Ceres 1.12 (trust_region_minimizer.cc, dogleg_strategy.cc,
// schur_complement_solver.cc); see oracle/oracle_solver.cpp for the step-by-step CPU
// statement these kernels are checked against.
#pragma once
#include "kernels_lin.h"
#include "tr_decide.h"

constexpr int SOLVE_THREADS = 256;
constexpr int NROW = KP + 1;                       // + rhs row (forward substitution fused)
constexpr int LPACK = NROW * (NROW + 1) / 2;       // 15051
// The reduced system lives in LDS as the lower 16 x 16 tiles of a 176 x 176 matrix (row 172 = rhs row, rows 173..175
// padding): tile (a, b), a >= b, at tile_id(a, b) * TSZ; inside a tile entry (r, k) sits at 17 r + k.  The row stride of
// 17 keeps every access pattern of the factorization free of bank conflicts with plain linear addressing (so the
// offsets fold into the instructions): an MFMA operand load (lane (c, g) reads row c, column g + 4 s), a D-layout
// store (rows g + 4 r, column c) and a row per lane.
constexpr int NTL = 11, NTILES = NTL * (NTL + 1) / 2;  // 66
constexpr int TLD = 17, TSZ = 16 * TLD;                 // 272 doubles per tile
constexpr int TPACK = NTILES * TSZ;                     // 17952 doubles
constexpr size_t SOLVE_LDS = (size_t)(TPACK + 256 + 8 * KP + 256 + 64) * sizeof(double);
__host__ __device__ constexpr int tile_id(int a, int b) { return a * (a + 1) / 2 + b; }
__host__ __device__ constexpr int tile_a(int t) {
  int a = 0;
  while ((a + 1) * (a + 2) / 2 <= t) a++;
  return a;
}
__host__ __device__ constexpr int tile_b(int t) { return t - tile_id(tile_a(t), 0); }
DEV int tsw(int r, int k) { return r * TLD + k; }
DEV int lidx(int i, int j) { return tile_id(i >> 4, j >> 4) * TSZ + tsw(i & 15, j & 15); }  // entry (i, j), j <= i
typedef double solve_d4 __attribute__((ext_vector_type(4)));
// every lane <- the lane with the same row (lane & 15) in quarter T (lanes 16 T .. 16 T + 15): two gfx950 lane swaps per dword
template <int T>
DEV int quarter_bcast32(int v) {
  auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);  // [q0 q0 q2 q2] | [q1 q1 q3 q3]
  const int w = (T & 1) ? r[1] : r[0];
  auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);  // [lo lo] | [hi hi]
  return (T & 2) ? q[1] : q[0];
}
DEV double quarter_bcast(double x, int t) {  // t: compile-time after unrolling
  const int hi = __double2hiint(x), lo = __double2loint(x);
  switch (t) {
    case 0: return __hiloint2double(quarter_bcast32<0>(hi), quarter_bcast32<0>(lo));
    case 1: return __hiloint2double(quarter_bcast32<1>(hi), quarter_bcast32<1>(lo));
    case 2: return __hiloint2double(quarter_bcast32<2>(hi), quarter_bcast32<2>(lo));
    default: return __hiloint2double(quarter_bcast32<3>(hi), quarter_bcast32<3>(lo));
  }
}
#define SOLVE_KEEP(v) asm volatile("" ::"v"(v))
#define STAMP(S, k) do { if (threadIdx.x == 0) (S)->dbg[k] = (long long)__builtin_readcyclecounter(); } while (0)

DEV double block_sum(double v, double *scratch, int tid) {
  v = wave_sum(v);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  double s = 0;
  for (int w = 0; w < SOLVE_THREADS / 64; w++) s += scratch[w];
  return s;
}
// several sums at once: one pair of barriers for all of them (same association as block_sum: butterfly inside the wave,
// then the four wave totals in order)
template <int NV>
DEV void block_sum_n(double (&v)[NV], double *scratch, int tid) {
#pragma unroll
  for (int k = 0; k < NV; k++) v[k] = wave_sum(v[k]);
  __syncthreads();
  if ((tid & 63) == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) scratch[(tid >> 6) * NV + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; k++) {
    double s = 0;
    for (int w = 0; w < SOLVE_THREADS / 64; w++) s += scratch[w * NV + k];
    v[k] = s;
  }
}
DEV double block_max(double v, double *scratch, int tid) {
  v = wave_max(v);
  __syncthreads();
  if ((tid & 63) == 0) scratch[tid >> 6] = v;
  __syncthreads();
  double s = 0;
  for (int w = 0; w < SOLVE_THREADS / 64; w++) s = fmax(s, scratch[w]);
  return s;
}

// ---------------------------------------------------------------------------
// Blocked Cholesky and back-substitution of a symmetric system held in LDS as the lower 16 x 16 tiles of a 16 TN x 16 TN matrix
// (tile layout above), TK pivots, the rhs as row TK (so the forward substitution rides along).  Shared by the dense solve of the
// whole reduced system (TN = 11, TK = 172: solve_body) and by the camera block that is left once the speed/bias chain is
// eliminated along its block structure (TN = 5, TK = 76: kernels_solveb.h).  Whole workgroup of SOLVE_THREADS threads; both end
// on a barrier.  tile_cholesky returns this thread's view of "a pivot was not positive" (or-ed into `bad`).
// ---------------------------------------------------------------------------
template <int TN, int TK>
DEV bool tile_cholesky(double *Hs, double *invd, int tid, bool bad) {
  // ---- blocked right-looking Cholesky, TN block columns of 16.  Per block column:
  //   F  wave 0 factors the diagonal tile, one ROW per lane, fully unrolled: pivot and column entries travel by
  //      v_readlane, the update uses the raw column (a_ik a_jk / d_k: the reciprocal runs beside the broadcasts), the
  //      columns are scaled by 1/sqrt(d_k) once at the end;
  //   P  the rows below solve  x L_kk^T = a  — one thread per row, L_kk read from a plain copy with uniform addresses;
  //   U  the trailing tiles take  A_ij -= L_ik L_jk^T  on the FP64 matrix pipe (4 v_mfma_f64_16x16x4_f64 per tile),
  //      tiles dealt round-robin to the four waves; operands and accumulators go straight between LDS and the MFMA
  //      register layouts.
  // The rhs row rides along as row TK - 16 (TN - 1) of the last block row (never a pivot), which is the forward substitution.
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;  // wave in an SGPR: scalar loop control below
  // F: factor diagonal tile kb in wave 0.  Lane (row c, quarter g) owns columns g, g + 4, g + 8, g + 12 of its row of the FULL
  // symmetric tile (a[i] = A[c][g + 4 i]).  The four pivots of a panel are eliminated on the vector pipe inside the panel only
  // (pivot by v_readlane, the pivot column for every quarter by v_permlane16/32_swap, the pivot row by DPP row_newbcast — all
  // requested before the reciprocal they run beside); the columns behind the panel take the rank-4 update
  // C -= P diag(1/d) P^T in ONE v_mfma_f64_16x16x4_f64 whose A, B and C operands are the registers as they stand: the
  // accumulator layout D[g + 4 r][c] is the transpose of the ownership, and the tile is symmetric.  30 vector instructions
  // per pivot instead of 47 (tools/micro/f_mfma.hip: 4 900 -> 3 700 cycles per tile); the raw columns are scaled by
  // 1/sqrt(d_k) once at the end.
  // last_term: the tile still lacks the term of block column kb - 1 (its panel tile (kb, kb - 1) has just been solved): it
  // is taken here, on the registers the factorization starts from — the accumulator layout of  A - P P^T  is this very
  // ownership (the tile is symmetric) — instead of a trip of the tile through LDS in between.
  auto factor = [&](int kb, bool last_term) {
    const int nb = kb < TN - 1 ? 16 : TK - 16 * (TN - 1);  // pivots in this block column (12 in the last)
    double *Td = Hs + tile_id(kb, kb) * TSZ;
    const int c = lane & 15, gq = lane >> 4;
    solve_d4 a;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int col = gq + 4 * i;
      a[i] = col <= c ? Td[tsw(c, col)] : Td[tsw(col, c)];
    }
    if (last_term) {
      const double *Tp = Hs + tile_id(kb, kb - 1) * TSZ + c * TLD + gq;
      double pv[4];
#pragma unroll
      for (int q = 0; q < 4; q++) pv[q] = Tp[4 * q];
#pragma unroll
      for (int q = 0; q < 4; q++) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-pv[q], pv[q], a, 0, 0, 0);
    }
    double dsave[4] = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
    for (int p = 0; p < 4; p++) {
      double bop = 0.0;  // this lane's B operand: its panel column times 1/d of that column's pivot
      // The four pivots of a panel without a cross-lane step between them.  Taken one at a time (the loop below), every
      // pivot is readlane -> reciprocal -> multiply -> multiply-add with a quarter and a row broadcast feeding it: ~250
      // cycles, most of them the hand-offs.  Instead everything the panel needs from other lanes is fetched FIRST, all of it
      // independent — the ten entries of its 4 x 4 pivot block (uniform, v_readlane) and this row's four panel entries (one
      // per quarter) — then every lane runs the block's LDL^T for itself (four reciprocals in a row, the only chain left) and
      // solves its own row against it: y_k = x_k - sum_{t<k} m_kt y_t, m_kt = Y_kt / d_t.  A panel is whole or absent (nb is
      // 16 or 12).
      if (4 * p < nb) {
        const int l0 = 4 * p;
        const double B00 = readlane_f64(a[p], l0), B10 = readlane_f64(a[p], l0 + 1), B20 = readlane_f64(a[p], l0 + 2), B30 = readlane_f64(a[p], l0 + 3);
        const double B11 = readlane_f64(a[p], 16 + l0 + 1), B21 = readlane_f64(a[p], 16 + l0 + 2), B31 = readlane_f64(a[p], 16 + l0 + 3);
        const double B22 = readlane_f64(a[p], 32 + l0 + 2), B32 = readlane_f64(a[p], 32 + l0 + 3), B33 = readlane_f64(a[p], 48 + l0 + 3);
        const double x0 = quarter_bcast(a[p], 0), x1 = quarter_bcast(a[p], 1), x2 = quarter_bcast(a[p], 2), x3 = quarter_bcast(a[p], 3);
        const double d0 = B00, r0 = fast_rcp(d0);
        const double m10 = B10 * r0, m20 = B20 * r0, m30 = B30 * r0;
        const double d1 = fma(-m10, B10, B11), Y21 = fma(-m10, B20, B21), Y31 = fma(-m10, B30, B31);
        const double r1 = fast_rcp(d1);
        const double m21 = Y21 * r1, m31 = Y31 * r1;
        const double d2 = fma(-m21, Y21, fma(-m20, B20, B22)), Y32 = fma(-m21, Y31, fma(-m20, B30, B32));
        const double r2 = fast_rcp(d2);
        const double m32 = Y32 * r2;
        const double d3 = fma(-m32, Y32, fma(-m31, Y31, fma(-m30, B30, B33)));
        const double r3 = fast_rcp(d3);
        if (!(d0 > 0.0) || !(d1 > 0.0) || !(d2 > 0.0) || !(d3 > 0.0)) bad = true;
        const double y1 = fma(-m10, x0, x1), y2 = fma(-m21, y1, fma(-m20, x0, x2)), y3 = fma(-m32, y2, fma(-m31, y1, fma(-m30, x0, x3)));
        const double yk = gq == 0 ? x0 : gq == 1 ? y1 : gq == 2 ? y2 : y3;
        const double rk = gq == 0 ? r0 : gq == 1 ? r1 : gq == 2 ? r2 : r3;
        dsave[p] = gq == 0 ? d0 : gq == 1 ? d1 : gq == 2 ? d2 : d3;
        a[p] = yk;
        bop = yk * rk;
      }
      if (p < 3 && 4 * p < nb) {
        solve_d4 cv = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[p], bop, a, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (i > p) a[i] = cv[i];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int col = gq + 4 * i;
      const double rs = fast_rsqrt(dsave[i]);
      double v = 0.0;
      if (col < nb) v = c > col ? a[i] * rs : (c == col ? dsave[i] * rs : 0.0);
      Td[tsw(c, col)] = v;
      if (c == col && col < nb) invd[16 * kb + col] = rs;
    }
  };
  // Two tiles (tA, tj) and (tB, tj) of ONE block column (tB >= TN: only the first) take the terms k0 .. k1 - 1 of
  // A_ij -= sum_k L_ik L_jk^T  with their accumulators in registers across the terms: the two share the B operand L_jk
  // (12 operand loads per term for two tiles instead of 16), their MFMA chains are independent, and the operands of term
  // k + 1 are requested before the MFMAs of term k.  Per tile and term the same four MFMAs in the same order as `update`
  // issues them, so a tile's value does not depend on how its terms are grouped into calls.
  // (TWO as a type: the body is straight-line code per variant, and inside the loop nothing is conditional — the operands of
  // the next term are fetched whether or not there is one (the tile behind the last operand tile exists: k1 <= tj) — so that
  // the wait in front of a term's MFMAs counts the loads of the NEXT term as still outstanding instead of waiting for them.)
  auto accumulate_impl = [&](auto two_c, int tA, int tB, int tj, int k0, int k1) {
    constexpr bool TWO = decltype(two_c)::value;
    const int c = lane & 15, gq = lane >> 4, offA = c * TLD + gq, offC = gq * TLD + c;
    double *TcA = Hs + tile_id(tA, tj) * TSZ + offC, *TcB = Hs + tile_id(TWO ? tB : tA, tj) * TSZ + offC;
    // the operand tiles of consecutive terms are consecutive tiles of a block row (tile_id(t, k + 1) = tile_id(t, k) + 1)
    const double *pA = Hs + tile_id(tA, k0) * TSZ + offA, *pB = Hs + tile_id(TWO ? tB : tA, k0) * TSZ + offA, *pJ = Hs + tile_id(tj, k0) * TSZ + offA;
    solve_d4 cA, cB = {0.0, 0.0, 0.0, 0.0};
    double xA[4], xB[4] = {0, 0, 0, 0}, xJ[4], yA[4], yB[4] = {0, 0, 0, 0}, yJ[4];  // two operand sets: no copies between terms
    auto fetch = [&](double (&a)[4], double (&b)[4], double (&jv)[4]) {
#pragma unroll
      for (int q = 0; q < 4; q++) a[q] = pA[4 * q], jv[q] = pJ[4 * q];
      if (TWO) {
#pragma unroll
        for (int q = 0; q < 4; q++) b[q] = pB[4 * q];
      }
      pA += TSZ, pB += TSZ, pJ += TSZ;
    };
    auto term = [&](const double (&a)[4], const double (&b)[4], const double (&jv)[4]) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        cA = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[q], jv[q], cA, 0, 0, 0);
        if (TWO) cB = __builtin_amdgcn_mfma_f64_16x16x4f64(-b[q], jv[q], cB, 0, 0, 0);
      }
    };
#pragma unroll
    for (int q = 0; q < 4; q++) cA[q] = TcA[4 * TLD * q];
    if (TWO) {
#pragma unroll
      for (int q = 0; q < 4; q++) cB[q] = TcB[4 * TLD * q];
    }
    fetch(xA, xB, xJ);
    int k = k0;
    for (; k + 1 < k1; k += 2) {
      fetch(yA, yB, yJ);
      term(xA, xB, xJ);
      fetch(xA, xB, xJ);
      term(yA, yB, yJ);
    }
    if (k < k1) term(xA, xB, xJ);
#pragma unroll
    for (int r = 0; r < 4; r++) TcA[4 * TLD * r] = cA[r];
    if (TWO) {
#pragma unroll
      for (int r = 0; r < 4; r++) TcB[4 * TLD * r] = cB[r];
    }
  };
  auto accumulate = [&](int tA, int tB, int tj, int k0, int k1) {
    if (tB < TN) accumulate_impl(std::true_type{}, tA, tB, tj, k0, k1);  // wave-uniform
    else accumulate_impl(std::false_type{}, tA, tB, tj, k0, k1);
  };
  // a wave's tiles t0, t0 + 3, t0 + 6 (those that exist) of block column tj
  auto accumulate_column = [&](int t0, int tj, int k0, int k1) {
    if (t0 < TN) accumulate(t0, t0 + 3, tj, k0, k1);
    if (t0 + 6 < TN) accumulate(t0 + 6, TN, tj, k0, k1);
  };
  if (wave == 0) factor(0, false);
  __syncthreads();
  for (int kb = 0; kb < TN - 1; kb++) {
    {  // P: the rows below solve x L_kk^T = a, one thread per row
      const int ta = kb + 1 + (tid >> 4), r = tid & 15;
      if (ta < TN) {
        double *Tp = Hs + tile_id(ta, kb) * TSZ;
        const double *Tk = Hs + tile_id(kb, kb) * TSZ;
        double x[16], dv[16];
#pragma unroll
        for (int j = 0; j < 16; j++) x[j] = Tp[tsw(r, j)], dv[j] = invd[16 * kb + j];
        // column j of L_kk (uniform addresses) is requested one step ahead: LDS reads complete in order, so a read issued
        // inside the step that uses it would cost that step a round trip
        double lc[16], ln[16];
#pragma unroll
        for (int t = 1; t < 16; t++) lc[t] = Tk[tsw(t, 0)];
#pragma unroll
        for (int j = 0; j < 16; j++) {
#pragma unroll
          for (int t = j + 2; t < 16; t++) ln[t] = Tk[tsw(t, j + 1)];
          x[j] *= dv[j];
#pragma unroll
          for (int t = j + 1; t < 16; t++) x[t] = fma(-x[j], lc[t], x[t]);
#pragma unroll
          for (int t = j + 2; t < 16; t++) lc[t] = ln[t];
        }
#pragma unroll
        for (int j = 0; j < 16; j++) Tp[tsw(r, j)] = x[j];
      }
    }
    __syncthreads();
    {  // U, left-looking with look-ahead.  A right-looking step would now update ALL trailing tiles with block column kb (55,
      // 45, 36 ... of them in the first columns, every one read and written through LDS: those columns were bound by that
      // traffic, 34k of the factorization's 100k cycles).  Only block column kb + 1 is needed next: it takes its LAST term
      // here — wave 0 the diagonal tile, which it then factors; waves 1..3 the tiles below it, which the next panel solve
      // reads — and while wave 0 factors, waves 1..3 give block column kb + 2 every term but its last one (the operand
      // panels 0 .. kb are final), accumulators in registers across the terms, two tiles at a time sharing their B operand:
      // a third of the LDS traffic per term, and never more than 30 tile-terms between two barriers.  Each tile still receives
      // its terms in the order 0, 1, 2, ...: same bits.
      const int j1 = kb + 1, j2 = kb + 2;
      if (wave == 0) {
        factor(j1, true);
      } else {
        accumulate_column(j1 + wave, j1, kb, kb + 1);
        accumulate_column(j2 + (3 - wave), j2, 0, kb + 1);  // (the wave with the most tiles above starts furthest down here)
      }
    }
    __syncthreads();
  }
  return bad;
}

// L^T y = z in place: z is the rhs row of the factored tiles, the result is left in yv[0, TK).
template <int TN, int TK>
DEV void tile_backsub(double *Hs, double *yv, const double *invd, int tid) {
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // ---- blocked back-substitution  L^T y = z  (TN diagonal blocks of 16), right-looking, from the last block to the
  //      first.  The diagonal tiles are inverted first, ALL of them at once — four tiles per wave, one column per lane,
  //      X = L_kk^-1 by forward substitution with the tile's entries as LDS broadcasts (120 fma per lane, no cross-lane
  //      traffic) — so that a block of the solution is a 16 x 16 product  y_k = X^T z_k  (16 independent fma per lane)
  //      instead of a chain of sixteen dependent broadcast steps; then every earlier block takes z_j -= L_kj^T y_k at
  //      once (thread (j, c): one column of one tile).  29 k -> 9 k cycles.
  if (tid < TK) yv[tid] = Hs[lidx(TK, tid)];  // z = the rhs row (it sits in the last diagonal tile: copied out before that tile is overwritten)
  __syncthreads();
  {
    const int tsel = 4 * wave + (lane >> 4), c = lane & 15;
    if (tsel < TN) {
      const int nb = tsel < TN - 1 ? 16 : TK - 16 * (TN - 1);
      double *Tk = Hs + tile_id(tsel, tsel) * TSZ;
      double x[16], dv[16], lr[16], ln[16];
#pragma unroll
      for (int r = 0; r < 16; r++) dv[r] = invd[16 * tsel + (r < nb ? r : 0)];
      lr[0] = Tk[tsw(1, 0)];
      x[0] = (0 >= c && 0 < nb) ? dv[0] : 0.0;
#pragma unroll
      for (int r = 1; r < 16; r++) {
        // row r + 1 of L is requested while row r is in work (an LDS read issued inside the chain costs it a round trip)
        if (r + 1 < 16) {
#pragma unroll
          for (int j = 0; j <= r; j++) ln[j] = Tk[tsw(r + 1, j)];
        }
        double acc = r == c ? 1.0 : 0.0, acc1 = 0.0;  // two accumulators: half the dependent chain
#pragma unroll
        for (int j = 0; j + 1 < r; j += 2) acc = fma(-lr[j], x[j], acc), acc1 = fma(-lr[j + 1], x[j + 1], acc1);  // x[j] = 0 above the diagonal
        if (r & 1) acc = fma(-lr[r - 1], x[r - 1], acc);
        x[r] = (r >= c && r < nb) ? (acc + acc1) * dv[r] : 0.0;
        if (r + 1 < 16) {
#pragma unroll
          for (int j = 0; j <= r; j++) lr[j] = ln[j];
        }
      }
      // the LDS operations of one wave complete in program order: every read above precedes these writes
#pragma unroll
      for (int r = 0; r < 16; r++) Tk[tsw(r, c)] = x[r];
    }
  }
  __syncthreads();
  auto tile_solve = [&](int blk) {  // 16 lanes of wave 0: y_blk = X^T z_blk, in place
    const int o = 16 * blk, nb = (TK - o) < 16 ? (TK - o) : 16, c = lane & 15;
    const double *Tk = Hs + tile_id(blk, blk) * TSZ;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r & 3] = fma(Tk[tsw(r, c)], r < nb ? yv[o + r] : 0.0, acc[r & 3]);
    if (c < nb) yv[o + c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  };
  auto apply = [&](int blk, int j, int c) {  // z_j[c] -= (L_kj^T y_k)[c]
    const int o = 16 * blk, nb = (TK - o) < 16 ? (TK - o) : 16;
    const double *Tj = Hs + tile_id(blk, j) * TSZ;
    double acc[4] = {yv[16 * j + c], 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < 16; r++)
      if (r < nb) acc[r & 3] = fma(-Tj[tsw(r, c)], yv[o + r], acc[r & 3]);
    yv[16 * j + c] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  };
  if (wave == 0 && lane < 16) tile_solve(TN - 1);
  __syncthreads();
  for (int blk = TN - 1; blk >= 1; blk--) {
    if (wave == 0) {  // look-ahead: the next block to be solved gets its update first
      if (lane < 16) {
        apply(blk, blk - 1, lane);
        tile_solve(blk - 1);
      }
    } else {
      const int j = (tid - 64) >> 4;
      if (j < blk - 1) apply(blk, j, tid & 15);
    }
    __syncthreads();
  }
}

// What thread 0 of the dense solve leaves in the header once the Gauss-Newton step is there: the pose-side sums,
// and at iteration 0 IterationZero + the first
// FinalizeIterationAndCheckIfMinimizerCanContinue of trust_region_minimizer.cc.
DEV void solve_epilogue(Slot *S, TRState *tr, const double *ls, double gn2, double ggn, double gG, double gN, double qgn, double qnn) {
  tr->q[Q_GN_SQ] = gn2;
  tr->q[Q_GRAD_GN] = ggn;
  tr->q[Q_gG] = gG;
  tr->q[Q_gN] = gN;
  tr->q[Q_GN] = qgn;
  tr->q[Q_NN] = qnn;
  tr->chol_fail = 0;
  // gradient_max_norm of the pose side is filled in by k_dogleg (off the critical path)
  if (S->sharded) tr->lm_bmax = ls[XS_BMAX];
  if (tr->iteration == 0) {
    {
      const FrameState *x0 = &S->x[tr->cur];
      double xn = ls[3];
      for (int f = 0; f < LFVIO_NUM_FRAMES; f++) {
        for (int k = 0; k < 7; k++) xn += x0->pose[f][k] * x0->pose[f][k];
        for (int k = 0; k < 9; k++) xn += x0->sb[f][k] * x0->sb[f][k];
      }
      if (S->est_ex)
        for (int k = 0; k < 7; k++) xn += x0->ex[k] * x0->ex[k];
      if (S->est_td) xn += x0->td * x0->td;
      tr->x_norm = sqrt(xn);
    }
    LfvioIterationSummary it;
    it.cost = tr->x_cost;
    it.cost_change = 0;
    it.gradient_max_norm = 0;
    it.step_norm = 0;
    it.relative_decrease = 0;
    it.trust_region_radius = tr->radius;
    it.step_is_valid = 0;
    it.step_is_successful = 0;
    tr->trace[0] = it;
    tr->trace_len = 1;
    tr->num_unsucc = 1;
    tr->initial_cost = tr->x_cost;
    tr->iteration = 1;
    tr->scaled = 1;
    tr->new_point = 1;
    if (S->max_iter <= 0) tr->done = 1;
    if (!isfinite(tr->x_cost)) {
      tr->done = 1;
      tr->error = LFVIO_ERR_NONFINITE;
    }
  } else if (tr->do_lin && tr->trace_len > 0) {
    tr->trace[tr->trace_len - 1].cost = tr->x_cost;  // HandleSuccessfulStep: cost re-evaluated at x
    tr->new_point = 1;
  }
}

// ---------------------------------------------------------------------------
// k_solve_dense: grid (1, batch) x 256, dynamic LDS = SOLVE_LDS.
// Input: H_pp (packed) and g_p assembled by k_sum, Schur sums, landmark scalars.
// ---------------------------------------------------------------------------
// xch_off / imu_off: byte offsets of the exchange buffer (H_pp | g_p | Schur sums | scalars) and of the IMU factor outputs
// inside a slot blob — passed by value so that the whole input of the kernel is requested in ONE round of loads, together
// with the loop flags and before the first branch (a GP<> member would have to be fetched first: one more round trip).
// ASSEMBLE (a resident batch linearized by k_linw, kernels_linw.h): the exchange buffer holds only the VISUAL terms of the
// camera part of H_pp (packed rows 0 .. KC - 1); the (at most two) IMU factors and the prior of every entry are added here,
// on load, in k_sum's order — the 119 KB packed matrix is never written or read back.  g_p arrives complete.
// Static scatter table of the assembling form (built once per context, lfvio_hip.hip build_asm_table): where the packed visual
// entry e goes in the LDS tiles (ASM_VIS ints), then for the IMU factors — even ones first, then odd ones: the factors of one
// parity share no column — every entry of the lower triangle of their 30 x 30 block as (index into imu_out) | (tile address << 16).
constexpr int ASM_VIS = KC * (KC + 1) / 2, ASM_IMU_F = 30 * 31 / 2, ASM_IMU_HALF = 5 * ASM_IMU_F, ASM_LEN = ASM_VIS + 2 * ASM_IMU_HALF;
__host__ __device__ constexpr int asm_lidx(int i, int j) { return tile_id(i >> 4, j >> 4) * TSZ + (i & 15) * TLD + (j & 15); }
template <bool ASSEMBLE>
DEV void solve_body(Slot *S, double *smem, long long xch_off, long long imu_off, long long prior_A_off, const int *asm_tab) {
  TRState *tr = &S->tr;
  const int tid = threadIdx.x;
  const int er = tid >> 4, ek = tid & 15, esw = tsw(er, ek);  // this thread's entry of every tile
  // This thread's entry (er, ek) of every lower tile: H_pp stays in these registers for the whole kernel — the reduced
  // system is built from them and the quadratic forms G^T H G, G^T H N, N^T H N are taken from them, so H_pp is read once.
  double hreg[NTILES];
  // the Schur sums this thread will subtract: entry (er, ek) of the 15 pose-side tiles, the rhs column, z2
  double sreg[15], srhs[5], scross = 0.0, gval = 0.0, cp = 0.0;
  const double *xch = (const double *)((const char *)S + xch_off);
  if (ASSEMBLE) {
    // (the assembly below is real work behind barriers: a slot with nothing to solve in this pass leaves first)
    const TRFlags f0 = tr_flags_decided(S);
    if ((f0.done | !f0.do_schur) && !S->dec_pending) return;
  }
  {
    const double *Hg = xch + XOFF_H, *Sg = xch + XOFF_S;
#pragma unroll
    for (int t = 0; t < NTILES; t++) {
      const int i = 16 * tile_a(t) + er, j = 16 * tile_b(t) + ek;
      if (!ASSEMBLE) hreg[t] = (i < KP && j <= i) ? Hg[i * (i + 1) / 2 + j] : 0.0;
    }
    if (ASSEMBLE) {
      // H_pp is assembled in the LDS tiles the reduced system will be built in, from three contiguous sources read with
      // coalesced loads — the prior's J0^T J0 (n x n, constant over the call), the visual terms of the camera part (packed,
      // k_linw), the ten 30 x 30 blocks of the IMU factors — each scattered to its entries by a phase of its own (within a
      // phase an entry has one writer: the prior's column map is injective, IMU factors f and f + 2 share no column), and
      // every thread then takes its entries of the tiles into the registers the rest of the kernel works from.
      double *Ht = smem;
      const double *imu_out = (const double *)((const char *)S + imu_off), *prior_A = (const double *)((const char *)S + prior_A_off);
      const int prior_ok = S->prior_valid && (!S->sharded || S->pose_side), prior_n = prior_ok ? S->prior_n : 0;
      const bool ex_on = S->est_ex != 0, td_on = S->est_td != 0;
      const int lane64 = tid & 63, wv4 = tid >> 6;
      // Requests first, in batches with fixed trip counts (a loop that loads, waits and scatters one entry at a time is a
      // memory round trip per entry): the prior's rows of this thread's two columns and its visual entries, then its IMU
      // entries; the tiles are cleared while the first batch is in flight.
      STAMP(S, 16);
      constexpr int PR_ROWS = 19, VIS_E = (ASM_VIS + SOLVE_THREADS - 1) / SOLVE_THREADS, IMU_E = (ASM_IMU_HALF + SOLVE_THREADS - 1) / SOLVE_THREADS;
      const int pc0 = lane64, pc1 = lane64 + 64;
      const int gj0 = pc0 < prior_n ? S->prior_cmap[pc0] : 0, gj1 = pc1 < prior_n ? S->prior_cmap[pc1] : 0;
      int gi_r[PR_ROWS];
      double pa0[PR_ROWS], pa1[PR_ROWS], vis[VIS_E];
#pragma unroll
      for (int k = 0; k < PR_ROWS; k++) {
        const int pr = wv4 + 4 * k, prc = pr < prior_n ? pr : 0;
        gi_r[k] = prior_n > 0 ? S->prior_cmap[prc] : 0;
        pa0[k] = prior_n > 0 ? prior_A[prc * prior_n + (pc0 < prior_n ? pc0 : 0)] : 0.0;
        pa1[k] = prior_n > 0 ? prior_A[prc * prior_n + (pc1 < prior_n ? pc1 : 0)] : 0.0;
      }
      int vdst[VIS_E];
#pragma unroll
      for (int k = 0; k < VIS_E; k++) {
        const int e = tid + SOLVE_THREADS * k, ec = e < ASM_VIS ? e : 0;
        vis[k] = Hg[ec], vdst[k] = asm_tab[ec];
      }
      STAMP(S, 17);
      {  // (16-byte stores: half as many of them)
        static_assert(TPACK % 2 == 0, "pairs");
        double2 *Ht2 = (double2 *)Ht;
#pragma unroll 5
        for (int e = tid; e < TPACK / 2; e += SOLVE_THREADS) Ht2[e] = make_double2(0.0, 0.0);
      }
      __syncthreads();
      STAMP(S, 18);
#pragma unroll
      for (int k = 0; k < PR_ROWS; k++) {
        const int pr = wv4 + 4 * k;
        if (pr < prior_n && pc0 < prior_n && gi_r[k] >= gj0) Ht[lidx(gi_r[k], gj0)] = pa0[k];
        if (pr < prior_n && pc1 < prior_n && gi_r[k] >= gj1) Ht[lidx(gi_r[k], gj1)] = pa1[k];
      }
      // (priors beyond 76 rows or 128 columns — not what the reference's marginalization produces, but legal input: plain loops)
      for (int pr = wv4 + 4 * PR_ROWS; pr < prior_n; pr += SOLVE_THREADS / 64) {
        const int gi = S->prior_cmap[pr];
        if (pc0 < prior_n && gi >= gj0) Ht[lidx(gi, gj0)] = prior_A[pr * prior_n + pc0];
        if (pc1 < prior_n && gi >= gj1) Ht[lidx(gi, gj1)] = prior_A[pr * prior_n + pc1];
      }
      for (int pc = lane64 + 128; pc < prior_n; pc += 64) {
        const int gj = S->prior_cmap[pc];
        for (int pr = wv4; pr < prior_n; pr += SOLVE_THREADS / 64) {
          const int gi = S->prior_cmap[pr];
          if (gi >= gj) Ht[lidx(gi, gj)] = prior_A[pr * prior_n + pc];
        }
      }
      STAMP(S, 19);
      double im[2][IMU_E];
      int idst[2][IMU_E];
#pragma unroll
      for (int par = 0; par < 2; par++)
#pragma unroll
        for (int k = 0; k < IMU_E; k++) {
          const int e = tid + SOLVE_THREADS * k;
          const int d = asm_tab[ASM_VIS + par * ASM_IMU_HALF + (e < ASM_IMU_HALF ? e : 0)];
          im[par][k] = imu_out[d & 0xffff], idst[par][k] = d >> 16;
        }
      __syncthreads();
      STAMP(S, 20);
#pragma unroll
      for (int k = 0; k < VIS_E; k++)
        if (tid + SOLVE_THREADS * k < ASM_VIS) Ht[vdst[k]] += vis[k];
      __syncthreads();
      STAMP(S, 21);
#pragma unroll
      for (int par = 0; par < 2; par++) {
#pragma unroll
        for (int k = 0; k < IMU_E; k++)
          if (tid + SOLVE_THREADS * k < ASM_IMU_HALF) Ht[idst[par][k]] += im[par][k];
        __syncthreads();
      }
#pragma unroll
      for (int t = 0; t < NTILES; t++) {
        const int i = 16 * tile_a(t) + er, j = 16 * tile_b(t) + ek;
        const bool act_i = (ex_on || i < off_ex() || i >= off_ex() + 6) && (td_on || i != off_td());
        const bool act_j = (ex_on || j < off_ex() || j >= off_ex() + 6) && (td_on || j != off_td());
        const double h = Ht[t * TSZ + esw];
        hreg[t] = (i < KP && j <= i && act_i && act_j) ? h : 0.0;
      }
      __syncthreads();  // (the tiles are rebuilt below, by other threads' schedules too: everyone has its copy first)
      STAMP(S, 22);
    }
    if (tid < KP) gval = xch[XOFF_G + tid];
    int n15 = 0;
#pragma unroll
    for (int t = 0; t < NTILES; t++)
      if (tile_a(t) <= 4) {
        const int i = 16 * tile_a(t) + er, j = 16 * tile_b(t) + ek;
        sreg[n15++] = Sg[schur_index(min(i, j), max(i, j))];
      }
#pragma unroll
    for (int b = 0; b < 5; b++) srhs[b] = Sg[schur_index(16 * b + ek, COL_B)];
    if (tid < KC) scross = Sg[schur_index(tid, COL_K)];
  }
  const int sharded = S->sharded;
  // landmark-side scalars: local sums, or the all-reduced totals of the sharded mode
  const double *ls = sharded ? xch + XOFF_C : S->lm_sum;
  // the pieces of the cost at x, one per thread (summed by thread 0 in the fixed order below)
  if (tid < 12) {
    if (tid == 0) cp = ls[0];
    else if (!sharded) cp = tid == 1 ? S->prior_g[KP] : ((const double *)((const char *)S + imu_off))[(size_t)(tid - 2) * IMU_OUT + 930];
  }
  const bool est_ex = S->est_ex != 0, est_td = S->est_td != 0;
  // A decision made in the prologue of this pass's k_lin is still parked in S->dec: this kernel is the one workgroup of
  // the slot, so it moves it into the header (the other threads read the header after the barriers below).
  const TRFlags fl = tr_flags_decided(S);
  const int dec_pending = S->dec_pending;
  const double mu_decided = S->dec.mu;  // (the other threads do not rely on seeing thread 0's header stores)
  const double mu_header = tr->mu;
#pragma unroll
  for (int t = 0; t < NTILES; t++) SOLVE_KEEP(hreg[t]);
#pragma unroll
  for (int t = 0; t < 15; t++) SOLVE_KEEP(sreg[t]);
#pragma unroll
  for (int t = 0; t < 5; t++) SOLVE_KEEP(srhs[t]);
  SOLVE_KEEP(scross);
  SOLVE_KEEP(gval);
  SOLVE_KEEP(cp);
  SOLVE_KEEP(mu_header);
  if (threadIdx.x == 0 && dec_pending) {
    decision_to_header(tr, S->dec);
    S->dec_pending = 0;
  }
  if (fl.done | !fl.do_schur) return;
  double *Hs = smem;                 // TPACK: H_pp, then S, then L as 16 x 16 tiles (rhs row = row 172)
  double *Ld = Hs + TPACK;           // 16 x 16: the diagonal block just factored, plain, TRANSPOSED (panel solve); later rhs
  double *g = Ld + 256;              // KP
  double *sc = g + KP;               // scale
  double *dg = sc + KP;              // diagonal_
  double *gr = dg + KP;              // gradient_
  double *Gd = gr + KP;              // unscaled Cauchy direction  S gr / dg
  double *yv = Gd + KP;              // y, then N direction
  double *hv = yv + KP;              // gauss_newton_step_
  double *invd = hv + KP;            // 1 / L_ii
  double *scratch = invd + KP;       // 256 (+64 pad)
  auto active = [&](int c) { return (est_ex || c < off_ex() || c >= off_ex() + 6) && (est_td || c != off_td()); };
  STAMP(S, 0);
#pragma unroll
  for (int a = 0; a < NTL; a++)
    if (er == ek && 16 * a + er < KP) hv[16 * a + er] = hreg[tile_id(a, a)];  // the diagonal, for the scaling below
  if (tid < KP) g[tid] = gval;
  if (tid < 12) scratch[tid] = cp;
  __syncthreads();
  if (fl.do_lin && tid == 0) {
    double cost = scratch[0];
    if (!sharded) {
      cost += scratch[1];
      for (int f = 0; f < LFVIO_WINDOW_SIZE; f++) cost += scratch[2 + f];
    }
    tr->x_cost = cost;
  }
  __syncthreads();  // scratch is reused below
  STAMP(S, 1);
  // ---- Jacobi scaling (iteration 0 only), diagonal_, gradient_  (dogleg_strategy.cc ComputeStep)
  const double mu = dec_pending ? mu_decided : mu_header;
  if (tid < KP) {
    const int i = tid;
    const double hii = hv[i];
    double s;
    if (!tr->scaled) {
      s = 1.0 / (1.0 + sqrt(hii));
      S->scale_p[i] = s;
    } else {
      s = S->scale_p[i];
    }
    const double d = sqrt(fmin(fmax(s * s * hii, 1e-6), 1e32));
    const double gi = active(i) ? s * g[i] / d : 0.0;
    sc[i] = s, dg[i] = d, gr[i] = gi;
    Gd[i] = s * gi / d;
    S->diag_p[i] = d;
    S->grad_p[i] = gi;
  }
  __syncthreads();
  STAMP(S, 2);
  // ---- reduced system, in place:  S = S_p (H_pp - Schur) S_p + mu D^2  and the rhs row; the Cauchy-point quadratic
  //      form G^T H G is accumulated from the same entries on the way.
  double qgg_part = 0, grhs_part = 0;  // grhs: G . (g - z1) over the active columns = gamma^T rhs of the scaled system (the forms below)
  {
    int n15 = 0;
    // per-thread slices of the vectors: row index 16 a + er, column index 16 b + ek
    double Gi[NTL], Gj[NTL], si[NTL], sj[NTL];
    bool ai[NTL], aj[NTL];
#pragma unroll
    for (int a = 0; a < NTL; a++) {
      const int i = 16 * a + er, j = 16 * a + ek;
      Gi[a] = i < KP ? Gd[i] : 0.0, si[a] = i < KP ? sc[i] : 0.0, ai[a] = i < KP && active(i);
      Gj[a] = j < KP ? Gd[j] : 0.0, sj[a] = j < KP ? sc[j] : 0.0, aj[a] = j < KP && active(j);
    }
#pragma unroll
    for (int t = 0; t < NTILES; t++) {
      const int a = tile_a(t), b = tile_b(t);
      const int i = 16 * a + er, j = 16 * b + ek;
      double v = 0.0;
      if (i < KP && j <= i) {
        double h = hreg[t];
        qgg_part = fma(h * Gi[a], (i == j) ? Gj[b] : 2.0 * Gj[b], qgg_part);
        if (ai[a] && aj[b]) {
          if (a <= 4 && i < KC) h -= sreg[a <= 4 ? n15 : 0];  // j <= i < 73
          v = si[a] * sj[b] * h;
          if (i == j) v += mu * dg[i] * dg[i];
        } else {
          v = (i == j) ? 1.0 : 0.0;
        }
      } else if (a == NTL - 1 && i == KP && j < KP) {
        if (aj[b]) {
          double r = g[j];
          if (j < KC) r -= srhs[b < 5 ? b : 0];  // z1
          v = sj[b] * r;
          grhs_part = fma(Gj[b], r, grhs_part);
        }
      }
      Hs[t * TSZ + esw] = v;
      if (a <= 4) n15++;
    }
  }
  // ---- Cauchy point: alpha = ||gradient_||^2 / ||J (gradient_/diagonal_)||^2
  {
    double gs = 0, cross = 0;
    if (tid < KP) {
      gs = gr[tid] * gr[tid];
      if (tid < KC) cross = scross * Gd[tid];  // z2 . G_c
    }
    double sums[3] = {qgg_part, gs, cross};
    block_sum_n(sums, scratch, tid);
    const double q_gg = sums[0], gsq = sums[1], cr = sums[2];
    if (tid == 0) {
      const double Jg2 = q_gg + 2.0 * cr + ls[2];
      const double gtot = gsq + ls[1];
      tr->alpha = gtot / Jg2;
      tr->grad_sq_total = gtot;
      tr->q[Q_GG] = q_gg;
      tr->q[Q_GRAD_SQ] = gsq;
    }
  }
  __syncthreads();
  STAMP(S, 3);

  // ---- blocked Cholesky of the 172 x 172 system (tile_cholesky above); ComputeGaussNewtonStep: `while (mu_ < max_mu_)` — no attempt at mu >= 1
  bool bad = tile_cholesky<NTL, KP>(Hs, invd, tid, !(mu < 1.0));
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  (void)wave, (void)lane;
  STAMP(S, 4);
  {
    double f = bad ? 1.0 : 0.0;
    f = block_max(f, scratch, tid);
    bad = f > 0.0;
  }
  STAMP(S, 5);
  const double zt = tid < KP ? Hs[lidx(KP, tid)] : 0.0;  // z = L^-1 rhs (the rhs row of the factor): y^T rhs = |z|^2, for N^T H N below
  tile_backsub<NTL, KP>(Hs, yv, invd, tid);
  STAMP(S, 6);
  {
    double f = 0.0;
    if (tid < KP && !isfinite(yv[tid])) f = 1.0;
    f = block_max(f, scratch, tid);
    if (f > 0.0) bad = true;
  }
  if (bad) {
    // LINEAR_SOLVER_FAILURE inside ComputeGaussNewtonStep: mu *= 10 and retry (same Jacobian)
    if (tid == 0) {
      tr->chol_fail = 1;
      if (mu < 1.0) tr->mu = mu * 10.0;
    }
    return;
  }
  // ---- Gauss-Newton step, directions and pose-side quadratic forms
  if (tid < KP) {
    const double y = yv[tid];
    const double gn = -dg[tid] * y;  // gauss_newton_step_ = -diagonal_ * y
    S->gn_p[tid] = gn;
    const double Nd = -sc[tid] * y;  // unscaled GN direction
    yv[tid] = Nd;
    hv[tid] = gn;
    if (tid < KC) {
      S->uc_grad[tid] = Gd[tid];
      S->uc_gn[tid] = Nd;
    }
  }
  if (tid >= KC && tid < WLD) S->uc_grad[tid] = S->uc_gn[tid] = 0.0;
  __syncthreads();
  {
    // G^T H N and N^T H N WITHOUT another pass over H_pp (round 5; it was a second walk over this thread's 66 tile entries).  With the
    // scaled system  S = s (H - C) s + mu D^2  (C: the landmarks' Schur sums, camera block), S y = rhs, N = -s y, G = s gamma:
    //   N^T H N = y^T rhs - mu |D y|^2 + N_c^T C N_c = |z|^2 - mu |gauss_newton|^2 + N_c^T C N_c         (z = L^-1 rhs: L^T y = z)
    //   G^T H N = -gamma^T rhs - mu gradient_ . gauss_newton + G_c^T C N_c                                (gamma^T rhs = G . (g - z1))
    // — sums over 172 entries and over the 15 camera tiles of C, every term of them already here.  No cancellation: |z|^2 and the C form
    // are non-negative and the mu terms are 1e-4 of them or less.
    double gn2 = 0, ggn = 0, gG = 0, gN = 0;
    if (tid < KP) {
      const double gn = hv[tid];
      gn2 = gn * gn;
      ggn = gr[tid] * gn;
      gG = g[tid] * Gd[tid];
      gN = g[tid] * yv[tid];
    }
    // N_c^T C N_c and G_c^T C N_c from the reduced Schur sums this thread has held since the start (the first is also the landmark
    // part of |gauss_newton|^2 an lfvio_group rank forms for itself: k_lm_cb2 above has the identities, with (sum c b w) . N_c)
    double nsn = 0, gcn = 0, z1n = 0;
    {
      int n15 = 0;
#pragma unroll
      for (int t = 0; t < NTILES; t++)
        if (tile_a(t) <= 4) {
          const int i = 16 * tile_a(t) + er, j = 16 * tile_b(t) + ek;
          if (i < KC && j <= i) {
            nsn = fma(sreg[n15], (i == j ? 1.0 : 2.0) * yv[i] * yv[j], nsn);
            gcn = fma(sreg[n15], i == j ? Gd[i] * yv[j] : Gd[i] * yv[j] + yv[i] * Gd[j], gcn);
          }
          n15++;
        }
    }
    if (sharded == 2) {
      if (er == 0) {
#pragma unroll
        for (int b = 0; b < 5; b++)
          if (16 * b + ek < KC) z1n = fma(srhs[b], yv[16 * b + ek], z1n);
      }
    }
    double pcb = 0, pnc = 0;  // (k_lm_cb2's partials, reduced over the ranks: a pair per thread)
    if (sharded == 2 && tid < XP_WGS) pcb = xch[XOFF_P + 2 * tid], pnc = xch[XOFF_P + 2 * tid + 1];
    double sums[11] = {gn2, ggn, gG, gN, zt * zt, grhs_part, nsn, z1n, pcb, pnc, gcn};
    block_sum_n(sums, scratch, tid);
    gn2 = sums[0], ggn = sums[1], gG = sums[2], gN = sums[3];
    const double qnn = sums[4] - mu * gn2 + sums[6], qgn = -sums[5] - mu * ggn + sums[10];
    STAMP(S, 7);
    if (tid == 0) {
      solve_epilogue(S, tr, ls, gn2, ggn, gG, gN, qgn, qnn);
      if (sharded == 2) {
        const double cb2 = sums[8], ncl = sums[9];
        tr->q[Q_LGN] = (cb2 + 2.0 * sums[7] + sums[6]) / (1.0 + mu);
        tr->q[Q_LGG] = -(cb2 + sums[7]);
        tr->q[Q_LEX] = ncl == 0.0 ? 1.0 : 0.0;  // (a landmark on the clamp: the first identity does not hold, the dogleg falls back)
      } else {
        tr->q[Q_LEX] = 0.0;
      }
    }
  }
}
template <bool ASSEMBLE>
__global__ __launch_bounds__(SOLVE_THREADS) void k_solve_dense(char *base, size_t stride, long long xch_off, long long imu_off, long long prior_A_off, const int *asm_tab) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  solve_body<ASSEMBLE>(SLOT(base, stride), smem, xch_off, imu_off, prior_A_off, asm_tab);
}

// ---------------------------------------------------------------------------
// k_backsub: grid (nLmBlocks, batch) x 64 — landmark part of the Gauss-Newton step
//   y_l = (s_l b_l - s_l w_l . (S_c y_c)) / e_l ;  gauss_newton_l = -diagonal_l y_l
// plus the two dot products w_l . G_c, w_l . N_c every dogleg interpolant needs.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_backsub(char *base, size_t stride) {
  Slot *S = SLOT(base, stride);
  const TRState *tr = &S->tr;
  {
    const TRFlags fl = tr_flags(tr);
    if (fl.done | !fl.do_schur | fl.chol_fail) return;
  }
  __shared__ double ug[WLD], un[WLD];
  const int lane = threadIdx.x;
  for (int c = lane; c < WLD; c += 64) ug[c] = S->uc_grad[c], un[c] = S->uc_gn[c];
  __syncthreads();
  const int l = blockIdx.x * LM_BLOCK + lane;
  double gn2 = 0, ggn = 0;
  if (l < S->N) {
    const double *w = S->W + S->lm_woff[l];  // stored over its non-zero span: [frames of the track | ex | td | ..]
    const int lo = 6 * S->lm_start[l], n6 = 6 * S->lm_cnt[l];
    double d1 = 0, d2 = 0;
    for (int c = 0; c < n6; c++) {
      const double wc = w[c];
      d1 = fma(wc, ug[lo + c], d1);
      d2 = fma(wc, un[lo + c], d2);
    }
#pragma unroll
    for (int c = 66; c < KC; c++) {
      const double wc = w[n6 + c - 66];
      d1 = fma(wc, ug[c], d1);
      d2 = fma(wc, un[c], d2);
    }
    const double s = S->scale_l[l];
    // s_l w_l . (S_c y_c) = -s_l d2   (N_c = -S_c y_c)
    const double y = (s * S->b[l] + s * d2) * S->einv_l[l];
    const double gn = -S->diag_l[l] * y;
    S->gn_l[l] = gn;
    S->d1[l] = d1;
    S->d2[l] = d2;
    gn2 = gn * gn;
    ggn = S->grad_l[l] * gn;
  }
  gn2 = wave_sum(gn2);
  ggn = wave_sum(ggn);
  if (lane == 0) {
    double *p = S->lm_part + (size_t)blockIdx.x * LMS;
    p[8] = gn2, p[9] = ggn;
  }
}

// ---------------------------------------------------------------------------
// k_dogleg<INLINE, WT>: grid (spec, batch) x 128 (INLINE: x DOGLEG_INLINE_THREADS) — ComputeTraditionalDoglegStep, candidate
// pose-side state, per-pair table of the candidate.  The body (dogleg_body) is also the prologue of k_step.
// ---------------------------------------------------------------------------
// k_backsub_wt: the same from the TRANSPOSED rows (Slot::Wt, column pairs wt_ld apart) — what k_linb leaves of a large window
// instead of the compact rows: a lane's row is at most WT_PAIRS coalesced 16-byte loads, taken over its own span only (frames
// start .. start + count - 1, then extrinsic and td: nothing outside it is defined).  Same products in the same order.
__global__ __launch_bounds__(64) void k_backsub_wt(char *base, size_t stride, int wt_ld) {
  Slot *S = SLOT(base, stride);
  const TRState *tr = &S->tr;
  {
    const TRFlags fl = tr_flags(tr);
    if (fl.done | !fl.do_schur | fl.chol_fail) return;
  }
  __shared__ double ug[WLD], un[WLD];
  const int lane = threadIdx.x;
  for (int c = lane; c < WLD; c += 64) ug[c] = S->uc_grad[c], un[c] = S->uc_gn[c];
  __syncthreads();
  const int l = blockIdx.x * LM_BLOCK + lane;
  double gn2 = 0, ggn = 0;
  if (l < S->N) {
    const int lo = 3 * S->lm_start[l], hi = lo + 3 * S->lm_cnt[l];
    const double s = S->scale_l[l], bl = S->b[l], einv = S->einv_l[l], dgl = S->diag_l[l], grl = S->grad_l[l];
    const double2 *wt = (const double2 *)(const double *)S->Wt + l;
    double2 wv[WT_PAIRS];
#pragma unroll
    for (int cp = 0; cp < WT_PAIRS; cp++) wv[cp] = ((cp >= lo && cp < hi) || cp >= 33) ? wt[(size_t)cp * wt_ld] : make_double2(0.0, 0.0);
    double d1 = 0, d2 = 0;
#pragma unroll
    for (int c = 0; c < KC; c++) {
      const double w = (c & 1) ? wv[c >> 1].y : wv[c >> 1].x;
      d1 = fma(w, ug[c], d1), d2 = fma(w, un[c], d2);
    }
    // s_l w_l . (S_c y_c) = -s_l d2   (N_c = -S_c y_c)
    const double y = (s * bl + s * d2) * einv;
    const double gn = -dgl * y;
    S->gn_l[l] = gn;
    S->d1[l] = d1;
    S->d2[l] = d2;
    gn2 = gn * gn;
    ggn = grl * gn;
  }
  gn2 = wave_sum(gn2);
  ggn = wave_sum(ggn);
  if (lane == 0) {
    double *p = S->lm_part + (size_t)blockIdx.x * LMS;
    p[8] = gn2, p[9] = ggn;
  }
}

// INLINE (windows of at most DOGLEG_INLINE_BLOCKS landmark blocks): the landmark part of the Gauss-Newton step (k_backsub)
// is formed here, one landmark per thread, and that launch is left out of the pass — at this size a kernel is a few
// microseconds of launch and first-load latency whatever it does.
constexpr int DOGLEG_INLINE_BLOCKS = 5, DOGLEG_INLINE_THREADS = 64 * DOGLEG_INLINE_BLOCKS;
// ComputeTraditionalDoglegStep for one radius: step = cg * gradient_ + cn * gauss_newton_, ||step|| = sn
DEV void dogleg_coeffs(double grad_sq_total, double gn_sq_total, double grad_gn_total, double alpha, double radius, double &cg, double &cn,
                       double &sn) {
  const double gradient_norm = sqrt(grad_sq_total), gauss_newton_norm = sqrt(gn_sq_total);
  if (gauss_newton_norm <= radius) {  // Case 1
    cg = 0.0, cn = 1.0, sn = gauss_newton_norm;
  } else if (gradient_norm * alpha >= radius) {  // Case 2
    cg = -(radius / gradient_norm), cn = 0.0, sn = radius;
  } else {  // Case 3
    const double b_dot_a = -alpha * grad_gn_total;
    const double a_squared_norm = (alpha * gradient_norm) * (alpha * gradient_norm);
    const double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + gauss_newton_norm * gauss_newton_norm;
    const double c = b_dot_a - a_squared_norm;
    const double d = sqrt(c * c + b_minus_a_squared_norm * (radius * radius - a_squared_norm));
    const double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + c);
    cg = -alpha * (1.0 - beta), cn = beta;
    // ||cg g + cn n||
    sn = sqrt(cg * cg * grad_sq_total + 2.0 * cg * cn * grad_gn_total + cn * cn * gn_sq_total);
  }
}
// spec: number of candidates to prepare (1, or 1 + SPEC_EXTRA for small windows): candidate z is the step for radius / 2^z.
// Launched with grid.x = spec: workgroup z prepares candidate z (each one repeats the short common part — the kernel is
// a latency chain, three of them side by side cost what one costs); workgroup 0 alone writes what is shared — except the
// landmark part of the step in the INLINE form, which every workgroup writes (the same values).
#ifdef LFVIO_DOGLEG_PROFILE
#define GSTAMP(k) do { if (blockIdx.x == 0) STAMP(S, k); } while (0)
#else
#define GSTAMP(k) do { } while (0)
#endif
// FUSED: the body runs as the prologue of k_step in EVERY workgroup of the candidate's cost evaluation; what the cost body
// reads afterwards (the landmark part of the step, the candidate state and its pair table) is then written by each of
// them — the same values from the same inputs — and read back behind a workgroup barrier.  sh2: [cg, cn] per candidate.
// Returns false when the slot takes no step in this pass.
// FUSED: the body runs as the prologue of k_step in EVERY workgroup of the candidate's cost evaluation; what the cost body
// reads afterwards (the landmark part of the step, the candidate state and its pair table) is then written by each of
// them — the same values from the same inputs — and read back behind a workgroup barrier.  sh2: [cg, cn] per candidate.
// Returns false when the slot takes no step in this pass.
// WT: the inline back-substitution reads Slot::Wt (launches whose k_lin has all roles in one grid and writes it); a resident
// batch, whose landmark role is a launch of its own and whose sweeps are throughput, keeps the compact rows of W.
template <bool FUSED, bool inline_backsub, bool WT, int WT_PARTS = 1>
DEV bool dogleg_body(Slot *S, int z_lo, int z_hi, bool first, int spec, double *sh2) {
  TRState *tr = &S->tr;
  const int tid = threadIdx.x, nthr = blockDim.x, nwv = nthr >> 6;
  GSTAMP(7);
  // The kernel is one chain of small dependent steps; a load issued behind a branch or a barrier costs a full memory
  // round trip (1 - 1.5 us) of its own, so everything that is read — the header, this thread's piece of the state, of the
  // gradient and of the step vectors, the landmark partials, the landmark rows — is requested here in one batch, before the first use.
  const TRHead t = *reinterpret_cast<const TRHead *>(tr);
  const int sharded = S->sharded, nLmBlocks = S->nLmBlocks, est_ex = S->est_ex, est_td = S->est_td, ex_off = S->ex_fixed_off;
  const int cur = t.cur, do_schur = t.do_schur;
  const FrameState *x = &S->x[cur];
  FrameState *xc = &S->x[cur ^ 1];
  const double lm4 = sharded ? 0.0 : S->lm_sum[4];
  const double xg0 = sharded ? S->xch[XOFF_C + XS_GN2] : 0.0, xg1 = sharded ? S->xch[XOFF_C + XS_GGN] : 0.0;
  double xb[7] = {0, 0, 0, 0, 0, 0, 0}, gpv[6] = {0, 0, 0, 0, 0, 0};
  const int role = tid < 12 ? 0 : (tid >= 16 && tid < 16 + 99) ? 1 : tid == 120 ? 2 : 3;
  const int po = tid < 11 ? off_pose(tid) : off_ex();
  if (role == 0) {
    const double *src = tid < 11 ? x->pose[tid] : x->ex;
#pragma unroll
    for (int k = 0; k < 7; k++) xb[k] = src[k];
#pragma unroll
    for (int k = 0; k < 6; k++) gpv[k] = S->gp[po + k];
  } else if (role == 1) {
    const int e = tid - 16;
    xb[0] = x->sb[e / 9][e % 9], gpv[0] = S->gp[off_sb(0) + e];
  } else if (role == 2) {
    xb[0] = x->td, gpv[0] = S->gp[off_td()];
  }
  double vgr[2], vgn[2], vdg[2], vsc[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int i = tid + nthr * q;
    const bool in = i < KP;
    vgr[q] = in ? S->grad_p[i] : 0.0, vgn[q] = in ? S->gn_p[i] : 0.0, vdg[q] = in ? S->diag_p[i] : 1.0, vsc[q] = in ? S->scale_p[i] : 0.0;
  }
  double a = 0, b = 0;
  if (do_schur && !inline_backsub && !sharded)
#pragma unroll 4
    for (int k = tid; k < nLmBlocks; k += nthr) {
      a += S->lm_part[(size_t)k * LMS + 8];
      b += S->lm_part[(size_t)k * LMS + 9];
    }
  const int last_ok = t.trace_len > 0 ? tr->trace[t.trace_len - 1].step_is_successful : 0;
  if (t.done || t.chol_fail) return false;
  GSTAMP(8);
  __shared__ double sh[2 * DOGLEG_INLINE_BLOCKS];
  __shared__ double delta[KP];
  __shared__ double cand[84 + TAB_SCRATCH];  // candidate poses (pose[0..10], ex) for build_tab, and its scratch
  if (do_schur && inline_backsub) {
    // k_backsub, one landmark per thread:  y_l = (s_l b_l - s_l w_l . (S_c y_c)) / e_l,  gauss_newton_l = -diagonal_l y_l,
    // and the two dot products w_l . G_c, w_l . N_c every dogleg interpolant needs
    static_assert(DOGLEG_INLINE_THREADS == SPEC_MAX_LM, "one landmark per thread of k_dogleg / k_step");
    double *ug = cand, *un = cand + WLD;
    for (int c = tid; c < WLD; c += nthr) ug[c] = S->uc_grad[c], un[c] = S->uc_gn[c];
    __syncthreads();
    // (one landmark per thread with the 320 threads of k_dogleg / k_step / k_stepw)
    // (written as two guarded trips, not a loop: with one trip known the loads above stay in one batch with these)
    auto row = [&](const int l) {
      const double s = S->scale_l[l], bl = S->b[l], einv = S->einv_l[l], dgl = S->diag_l[l], grl = S->grad_l[l];
      double d1 = 0, d2 = 0;
      if (WT) {
        // The landmark's row from the transposed copy (Slot::Wt, written by k_lin): WT_PAIRS 16-byte loads that the lanes of
        // a wave share line by line, all in flight before the first product (a wave holds 63 loads in flight at most; from
        // the compact rows of W it is KC loads of 64 lines each: 10 500 cycles for this block, 5 900 like this).  The entries
        // outside the track's span are zeros, which leave the sums as they are: same products, same order.
        // (WT_PARTS > 1 — k_stepw, whose workgroups are throughput: the row in that many batches of loads, so that the 148
        // registers of a whole row do not decide how many workgroups share a CU; same products, same order)
        const double2 *wt = (const double2 *)(const double *)S->Wt + l;
        constexpr int PER = (WT_PAIRS + WT_PARTS - 1) / WT_PARTS;
#pragma unroll
        for (int part = 0; part < WT_PARTS; part++) {
          double2 wv[PER];
#pragma unroll
          for (int k = 0; k < PER; k++) {
            const int cp = part * PER + k;
            wv[k] = cp < WT_PAIRS ? wt[(size_t)cp * SPEC_MAX_LM] : make_double2(0.0, 0.0);
          }
#pragma unroll
          for (int k = 0; k < PER; k++) {
            const int c0 = 2 * (part * PER + k);
            if (c0 < KC) d1 = fma(wv[k].x, ug[c0], d1), d2 = fma(wv[k].x, un[c0], d2);
            if (c0 + 1 < KC) d1 = fma(wv[k].y, ug[c0 + 1], d1), d2 = fma(wv[k].y, un[c0 + 1], d2);
          }
        }
      } else {
        // the compact row: all eleven frame slots of it at once (a shorter track reads on into the rows behind it, which are
        // dropped), the dot products then run on registers
        const int woff = S->lm_woff[l], st = S->lm_start[l], cnt = S->lm_cnt[l];
        const double *w = S->W + woff;
        const int lo = 6 * st, n6 = 6 * cnt;
        double wv[66], wt[KC - 66];
#pragma unroll
        for (int c = 0; c < 66; c++) wv[c] = w[c];
#pragma unroll
        for (int c = 66; c < KC; c++) wt[c - 66] = w[n6 + c - 66];
#pragma unroll
        for (int f = 0; f < 11; f++)
          if (f < cnt) {
#pragma unroll
            for (int e = 0; e < 6; e++) {
              d1 = fma(wv[6 * f + e], ug[lo + 6 * f + e], d1);
              d2 = fma(wv[6 * f + e], un[lo + 6 * f + e], d2);
            }
          }
#pragma unroll
        for (int c = 66; c < KC; c++) {
          d1 = fma(wt[c - 66], ug[c], d1);
          d2 = fma(wt[c - 66], un[c], d2);
        }
      }
      const double y = (s * bl + s * d2) * einv;
      const double gn = -dgl * y;
      {  // (every workgroup of the launch, candidate z's or — k_step — cost block's: the same values)
        S->gn_l[l] = gn;
        S->d1[l] = d1;
        S->d2[l] = d2;
      }
      a += gn * gn;
      b += grl * gn;
    };
    if (tid < S->N) row(tid);
    if (nthr < SPEC_MAX_LM && tid + nthr < S->N) row(tid + nthr);
    __syncthreads();  // ug / un alias cand
  }
  GSTAMP(9);
  // total norms: pose side (k_solve) + landmark partials (k_backsub)
  a = wave_sum(a), b = wave_sum(b);
  if ((tid & 63) == 0) sh[(tid >> 6) * 2] = a, sh[(tid >> 6) * 2 + 1] = b;
  __syncthreads();
  if (tid == 0) {
    double gn_sq_total = t.gn_sq_total, grad_gn_total = t.grad_gn_total;
    const double grad_sq_total = t.grad_sq_total, radius = t.radius, alpha = t.alpha;
    if (do_schur) {
      double lgn = 0, lgg = 0;
      for (int w = 0; w < nwv; w++) lgn += sh[2 * w], lgg += sh[2 * w + 1];
      if (sharded) lgn = xg0, lgg = xg1;  // all-reduced (k_xpack 2)
      gn_sq_total = t.q[Q_GN_SQ] + lgn;
      grad_gn_total = t.q[Q_GRAD_GN] + lgg;
      if (first) {
        tr->gn_sq_total = gn_sq_total;
        tr->grad_gn_total = grad_gn_total;
      }
    }
    // A rank of an lfvio_group (Slot::sharded 2) does not know the landmark part of ||gauss_newton||^2 here — it travels with the
    // candidate's cost, in ONE all-reduce behind this kernel: the candidate of a pass with a fresh solve is the Gauss-Newton
    // step itself (what the dogleg takes whenever that step fits the radius), and k_decide, which sees the reduced norms, either
    // confirms it or voids the pass (decide_body); a pass without a fresh solve has the totals in the header and takes them.
    // ... unless the solve has already formed the landmark parts from the reduced Schur sums (k_lm_cb2: exact while no landmark
    // of the window sits on the diagonal clamp), which is the usual case: then this IS the dogleg step and nothing is left to confirm.
    bool spec_gn = sharded == 2 && do_schur;
    if (spec_gn && t.q[Q_LEX] == 1.0) {
      gn_sq_total = t.q[Q_GN_SQ] + t.q[Q_LGN];
      grad_gn_total = t.q[Q_GRAD_GN] + t.q[Q_LGG];
      if (first) tr->gn_sq_total = gn_sq_total, tr->grad_gn_total = grad_gn_total;
      spec_gn = false;
    }
    if (first && sharded == 2) tr->gn_unconfirmed = spec_gn ? 1 : 0;
    for (int z = z_lo; z < z_hi; z++) {
      double cg, cn, sn;
      dogleg_coeffs(grad_sq_total, gn_sq_total, grad_gn_total, alpha, ldexp(radius, -z), cg, cn, sn);
      if (spec_gn) cg = 0.0, cn = 1.0, sn = 0.0;
      if (z == 0) tr->cg = cg, tr->cn = cn, tr->dogleg_step_norm = sn;
      else tr->cgE[z - 1] = cg, tr->cnE[z - 1] = cn, tr->snE[z - 1] = sn;
      sh2[2 * z] = cg, sh2[2 * z + 1] = cn;
    }
    if (first) tr->spec_n = spec;
  }
  __syncthreads();
  if (t.new_point && first) {
    // gradient_max_norm = max |x - Plus(x, -g)| (EvaluateGradientAndJacobian), pose side + landmarks
    double mx = 0;
    if (role == 0) {
      if (tid < 11 || est_ex) {
        double d[6], xo[7];
        for (int k = 0; k < 6; k++) d[k] = -gpv[k];
        pose_plus(xb, d, xo);
        for (int k = 0; k < 7; k++) mx = fmax(mx, fabs(xb[k] - xo[k]));
      }
    } else if (role == 1) {
      mx = fabs(gpv[0]);
    } else if (role == 2 && est_td) {
      mx = fabs(gpv[0]);
    }
    mx = wave_max(mx);
    __syncthreads();
    if ((tid & 63) == 0) sh[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
      double gm = sharded ? t.lm_bmax : lm4;
      for (int w = 0; w < nwv; w++) gm = fmax(gm, sh[w]);
      tr->gmax_pose = gm;
      if (t.trace_len > 0) {
        tr->trace[t.trace_len - 1].gradient_max_norm = gm;
        if (last_ok && gm <= 1e-10) {  // GradientToleranceReached
          tr->termination = LFVIO_CONVERGENCE;
          tr->done = 1;
        }
      }
      tr->new_point = 0;
    }
    __syncthreads();
  }
  for (int z = z_lo; z < z_hi; z++) {
    const double cg = sh2[2 * z], cn = sh2[2 * z + 1];
    FrameState *xz = z == 0 ? xc : &S->xE[z - 1];
    GSTAMP(18);
    // delta = (step / diagonal_) * scale, step = cg gradient_ + cn gauss_newton_
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int i = tid + nthr * q;
      if (i < KP) {
        double d = (cg * vgr[q] + cn * vgn[q]) / vdg[q] * vsc[q];
        const bool act = (est_ex || i < off_ex() || i >= off_ex() + 6) && (est_td || i != off_td());
        if (!act) d = 0.0;
        delta[i] = d;
        if (z == 0) S->step_p[i] = d;
      }
    }
    __syncthreads();
    // candidate = Plus(x, delta); ambient step norm and candidate norm, pose side
    double dn = 0, xn = 0;
    if (role == 0) {
      double *xo = tid < 11 ? xz->pose[tid] : xz->ex;
      double xv[7];
      if (tid < 11 || est_ex) {
        pose_plus(xb, delta + po, xv);
        for (int k = 0; k < 7; k++) dn += (xb[k] - xv[k]) * (xb[k] - xv[k]), xn += xv[k] * xv[k];
      } else {
        for (int k = 0; k < 7; k++) xv[k] = xb[k];
      }
      for (int k = 0; k < 7; k++) xo[k] = xv[k], cand[7 * tid + k] = xv[k];
    } else if (role == 1) {
      const int e = tid - 16, f = e / 9, k = e % 9;
      const double v = xb[0] + delta[off_sb(f) + k];
      xz->sb[f][k] = v;
      dn = (v - xb[0]) * (v - xb[0]);
      xn = v * v;
    } else if (role == 2) {
      double v = xb[0];
      if (est_td) {
        v = xb[0] + delta[off_td()];
        dn = (v - xb[0]) * (v - xb[0]);
        xn = v * v;
      }
      xz->td = v;
    }
    dn = wave_sum(dn), xn = wave_sum(xn);
    if ((tid & 63) == 0) sh[(tid >> 6) * 2] = dn, sh[(tid >> 6) * 2 + 1] = xn;
    __syncthreads();
    if (tid == 0) {
      double sdn = 0, sxn = 0;
      for (int w = 0; w < nwv; w++) sdn += sh[2 * w], sxn += sh[2 * w + 1];
      if (z == 0) tr->step_sq_pose = sdn, tr->xn2_pose_cand = sxn;
      else tr->step_sqE[z - 1] = sdn, tr->xn2E[z - 1] = sxn;
    }
    GSTAMP(19);
    build_tab<false>(cand, z == 0 ? &S->tab[cur ^ 1] : &S->tabE[z - 1], tid, cand + 84, ex_off);  // ends on a barrier: delta / sh / cand are free again
    GSTAMP(3);
  }
  return true;
}
template <bool INLINE, bool WT = false>
__global__ __launch_bounds__(INLINE ? DOGLEG_INLINE_THREADS : 128) void k_dogleg(char *base, size_t stride, int spec) {
  __shared__ double sh2[2 * (1 + SPEC_EXTRA)];
  // (every launch has grid.x = spec: workgroup z prepares candidate z)
  dogleg_body<false, INLINE, WT>(SLOT(base, stride), (int)blockIdx.x, (int)blockIdx.x + 1, blockIdx.x == 0, spec, sh2);
}

// ---------------------------------------------------------------------------
// k_cost<LPT>: grid (nLmBlocks + 10 + 1, batch) x 64 LPT — candidate point:
//   landmark blocks: lambda_c = lambda + delta_l, cost of every observation at the candidate, landmark part of the model
//                    cost change and of the norms
//   IMU / prior blocks: residual-only evaluation at the candidate
// LPT = lanes per track (and per row of J0 in the prior block).  4: a track of 11 observations is three deep instead of
// ten — the latency of a single window; 1: a quarter of the waves for the same work — the throughput of a resident batch.
// ---------------------------------------------------------------------------
// IMU_ROLE false: the grid has no workgroups for the IMU factors (k_cost_imu evaluates them one lane per factor: resident
// batches, where a workgroup per factor with its residual on one lane is mostly issue slots spent on idle lanes) and the
// instantiation carries none of that code.
// THREADS > 64 LPT (the body behind the dogleg prologue of k_step, whose workgroups are wider): the spare threads go
// through the barriers with nothing to add.
template <int LPT, bool IMU_ROLE, int THREADS>
DEV void cost_body(Slot *S, int cur, int z, int b, int gLm, double cg, double cn) {
  constexpr int NW = LPT;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool act = THREADS == 64 * LPT || tid < 64 * LPT;
  const int nxt = cur ^ 1;
  const Tab *T = z == 0 ? &S->tab[nxt] : &S->tabE[z - 1];
  const FrameState *x = z == 0 ? &S->x[nxt] : &S->xE[z - 1];
  double *lam_out = z == 0 ? (double *)S->lam[nxt] : (double *)S->lamE[z - 1];
  double *cost_part = z == 0 ? (double *)S->cost_part : (double *)S->cost_partE + (size_t)(z - 1) * (SPEC_MAX_LM / 64) * LMS;
  double *pose_cost = z == 0 ? S->pose_cost : S->pose_costE[z - 1];
  __shared__ double red[4 * 8];
  if (b < gLm) {
    if (b >= S->nLmBlocks) return;
    const double td = x->td;
    const int lml = tid / LPT, q = tid % LPT;
    const int l = b * LM_BLOCK + lml;
    double cost = 0, mlin = 0, mquad = 0, dn = 0, xn = 0;
    if (act && l < S->N) {
      const double s = S->scale_l[l];
      const double dl = (cg * S->grad_l[l] + cn * S->gn_l[l]) / S->diag_l[l] * s;
      const double lam = S->lam[cur][l];
      const double lc = lam + dl;
      if (q == 0) {
        lam_out[l] = lc;
        dn = dl * dl;
        xn = lc * lc;
        // model: -(delta.g) - 1/2 delta^T H delta, landmark rows/cols
        const double wd = cg * S->d1[l] + cn * S->d2[l];  // w_l . delta_c
        mlin = dl * S->b[l];
        mquad = 2.0 * dl * wd + S->a[l] * dl * dl;
      }
      const int i = S->lm_start[l], k = S->lm_cnt[l], o0 = S->lm_obs0[l];
      ObsPair ob;
      load_obs(S, o0, ob.pi, ob.vi, ob.tdi, ob.rowi);
      if (LPT == 4) {
        // at most three observations per lane (tracks are <= 11 long): all of them and their pair tables are requested
        // before the first residual is formed
        ObsPair obq[3];
        m33 Tq[3];
        d3 cq[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
          const int o = 1 + q + 4 * u;
          if (o < k) {
            const int pair = i * 11 + i + o;
            load_obs(S, o0 + o, obq[u].pj, obq[u].vj, obq[u].tdj, obq[u].rowj);
            Tq[u] = ldm(T->T[pair]), cq[u] = ld3(T->c[pair]);
          }
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
          const int o = 1 + q + 4 * u;
          if (o < k) {
            obq[u].pi = ob.pi, obq[u].vi = ob.vi, obq[u].tdi = ob.tdi, obq[u].rowi = ob.rowi;
            cost += 0.5 * visual_cost(obq[u], lc, td, S->est_td, S->tr_over_row, S->half_row, S->sqrt_info, Tq[u], cq[u]);
          }
        }
      } else {
        for (int o = 1 + q; o < k; o += LPT) {
          const int pair = i * 11 + i + o;
          load_obs(S, o0 + o, ob.pj, ob.vj, ob.tdj, ob.rowj);
          cost += 0.5 * visual_cost(ob, lc, td, S->est_td, S->tr_over_row, S->half_row, S->sqrt_info, ldm(T->T[pair]),
                                    ld3(T->c[pair]));
        }
      }
    }
    cost = wave_sum(cost), mlin = wave_sum(mlin), mquad = wave_sum(mquad), dn = wave_sum(dn), xn = wave_sum(xn);
    if (lane == 0 && act) red[wv * 8] = cost, red[wv * 8 + 1] = mlin, red[wv * 8 + 2] = mquad, red[wv * 8 + 3] = dn, red[wv * 8 + 4] = xn;
    __syncthreads();
    if (tid < 5) {
      double v = red[tid];
      for (int w = 1; w < NW; w++) v += red[8 * w + tid];
      cost_part[(size_t)b * LMS + tid] = v;
    }
    return;
  }
  b -= gLm;
  if (IMU_ROLE && b < LFVIO_WINDOW_SIZE) {
    __shared__ double rr[15];
    double c = 0.0;
    if (S->imu_active[b]) {
      if (tid == 0) imu_raw_residual(&S->imu[b], S->g, x->pose[b], x->sb[b], x->pose[b + 1], x->sb[b + 1], rr);
      __syncthreads();
      if (wv == 0) {
        double v = 0;
        if (lane < 15) {
          const double *Sq = S->imu_sqrt[b];
          for (int k = lane; k < 15; k++) v = fma(Sq[lane * 15 + k], rr[k], v);
          v = v * v;
        }
        c = 0.5 * wave_sum(v);
      }
    }
    if (tid == 0) pose_cost[b] = c;
    return;
  }
  {
    __shared__ double dx[KP];
    double c = 0.0;
    if (S->prior_valid) {
      const int n = S->prior_n;
      const double *J = S->prior_J;
      if (LPT == 4) {
        // r = r0 + J0 dx, n <= 76 rows: four lanes per row with 19 columns each, rows tid / 4 and that + 64.  The entries of J0
        // a lane needs are requested in one batch (a loop that loads an entry and uses it is a memory round trip per entry:
        // 16 000 cycles for this block, the longest of the cost evaluation), before dx is even formed.
        const int q = tid % LPT, c0 = q * 19, r0 = act ? tid / LPT : n;
        double jv[2][19], pr0 = 0.0, pr1 = 0.0;
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
          const int row = r0 + 64 * rr;
          const double prv = (q == 0 && row < n) ? S->prior_r[row] : 0.0;
          if (rr == 0) pr0 = prv;
          else pr1 = prv;
#pragma unroll
          for (int k = 0; k < 19; k++) jv[rr][k] = (row < n && c0 + k < n) ? J[row * n + c0 + k] : 0.0;
        }
        if (tid < S->prior_nb) prior_block_dx(S, x, tid, dx);
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; rr++) {
          const int row = r0 + 64 * rr;
          if (row < n) {
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 19; k++)
              if (c0 + k < n) s = fma(jv[rr][k], dx[c0 + k], s);
            s = quad_sum(s);
            if (q == 0) {
              s += rr == 0 ? pr0 : pr1;
              c += s * s;
            }
          }
        }
      } else {
        if (tid < S->prior_nb) prior_block_dx(S, x, tid, dx);
        __syncthreads();
        for (int row = tid; row < n; row += 64) {
          double s = 0.0;
          for (int cc = 0; cc < n; cc++) s = fma(J[row * n + cc], dx[cc], s);
          s += S->prior_r[row];
          c += s * s;
        }
      }
      c = wave_sum(c);
      if (lane == 0 && act) red[wv] = c;
      __syncthreads();
      c = red[0];
      for (int w = 1; w < NW; w++) c += red[w];
      c *= 0.5;
    }
    if (tid == 0) pose_cost[10] = c;
  }
}

template <int LPT, bool IMU_ROLE = true>
__global__ __launch_bounds__(64 * LPT) void k_cost(char *base, size_t stride, int gLm, int spec) {
  Slot *S = SLOT(base, stride);
  const TRState *tr = &S->tr;
  // header fields in one batch of loads, before the first branch (a load behind a branch is a round trip of its own)
  const TRFlags fl = tr_flags(tr);
  // candidate z of the pass: blocks [z * nb, (z + 1) * nb) of the grid, nb = gLm + 10 + 1 (spec = 1: the one candidate)
  const int nb = gLm + (IMU_ROLE ? LFVIO_WINDOW_SIZE : 0) + 1;
  const int z = spec > 1 ? (int)blockIdx.x / nb : 0;
  const double cg = z == 0 ? tr->cg : tr->cgE[z > 0 ? z - 1 : 0], cn = z == 0 ? tr->cn : tr->cnE[z > 0 ? z - 1 : 0];
  if (fl.done | fl.chol_fail) return;
  cost_body<LPT, IMU_ROLE, 64 * LPT>(S, fl.cur, z, (int)blockIdx.x - z * nb, gLm, cg, cn);
}

// ---------------------------------------------------------------------------
// k_step: grid (spec (gLm + 10 + 1), batch) x 320 — k_dogleg (with the landmark back-substitution inline) and k_cost<4> as ONE
// launch, for the few small windows whose passes are latency.  Workgroup (z, b) — candidate z, cost block b — first forms
// the step like k_dogleg's workgroup z does (every one of them: the dogleg is a chain of a few microseconds that needs the
// whole window's norms, and sixteen workgroups repeating it side by side take no longer than one), then evaluates its own
// block of the candidate like k_cost.  No workgroup waits for another; what several write they write identically.
// One launch and one round of first loads less per pass.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(DOGLEG_INLINE_THREADS) void k_step(char *base, size_t stride, int gLm, int spec) {
  Slot *S = SLOT(base, stride);
  __shared__ double sh2[2 * (1 + SPEC_EXTRA)];
  const int nb = gLm + LFVIO_WINDOW_SIZE + 1;
  const int z = spec > 1 ? (int)blockIdx.x / nb : 0;
  const int cur = tr_flags(&S->tr).cur;
  // (what one workgroup writes for all — the totals, the gradient norm of a new point — falls to the one with the shortest
  // cost block: the first IMU factor of candidate 0)
  const int b = (int)blockIdx.x - z * nb;
  if (!dogleg_body<true, true, true>(S, z, z + 1, (int)blockIdx.x == gLm, spec, sh2)) return;
#ifdef LFVIO_DOGLEG_PROFILE
  const long long t_role = (long long)__builtin_readcyclecounter();
#endif
  // (dogleg_body ends on a workgroup barrier: this workgroup's writes of the candidate are visible to all of its waves)
  cost_body<4, true, DOGLEG_INLINE_THREADS>(S, cur, z, b, gLm, sh2[2 * z], sh2[2 * z + 1]);
  GSTAMP(20);
#ifdef LFVIO_DOGLEG_PROFILE
  if ((int)blockIdx.x == gLm && threadIdx.x == 0) S->dbg[21] = (long long)__builtin_readcyclecounter() - t_role;     // IMU factor 0 of candidate 0
  if ((int)blockIdx.x == nb - 1 && threadIdx.x == 0) S->dbg[22] = (long long)__builtin_readcyclecounter() - t_role;  // prior block of candidate 0
#endif
}

// ---------------------------------------------------------------------------
// k_xpack: grid (1, batch) x 256 — sharded mode only: local scalar partials into the exchange scalars.
//   which & 7:  2  phase B (after k_backsub): landmark parts of ||gauss_newton||^2 and gradient . gauss_newton
//               3  phase C (after k_cost): the candidate's cost and model terms
//               6  both B and C (lfvio_group: one all-reduce carries them)
//   which & 8:  the scalars of another phase are in the tail already — nothing is zeroed; otherwise everything else is
//               zeroed so that the caller can sum-all-reduce the 16-scalar tail blindly.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_xpack(char *base, size_t stride, int which_bits) {
  const int which = which_bits & 7;
  Slot *S = SLOT(base, stride);
  const TRState *tr = &S->tr;
  if (!S->sharded) return;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  double *sc = S->xch + XOFF_C;
  __shared__ double red[4][7];
  double v[7] = {0, 0, 0, 0, 0, 0, 0};
  const TRFlags fl = tr_flags(tr);
  const bool want_b = which == 2 || which == 6, want_c = which == 3 || which == 6;
  if (!fl.done && !fl.chol_fail) {
    // a rank of a large window has thousands of block partials: 256 threads, the loads of four blocks in flight per thread
    const int nb = S->nLmBlocks;
    const double *srcb = (const double *)S->lm_part + 8, *srcc = (const double *)S->cost_part;
    for (int k0 = tid; k0 < nb; k0 += 4 * 256) {
      double t[4][7];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = k0 + 256 * u;
#pragma unroll
        for (int q = 0; q < 2; q++) t[u][q] = (k < nb && want_b) ? srcb[(size_t)k * LMS + q] : 0.0;
#pragma unroll
        for (int q = 0; q < 5; q++) t[u][2 + q] = (k < nb && want_c) ? srcc[(size_t)k * LMS + q] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int q = 0; q < 7; q++) v[q] += t[u][q];
    }
    if (want_c && S->pose_side == 1 && tid < 11) v[2] += S->pose_cost[tid];
  }
#pragma unroll
  for (int q = 0; q < 7; q++) v[q] = wave_sum(v[q]);
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 7; q++) red[wv][q] = v[q];
  }
  if (tid < 16 && !(which_bits & 8)) sc[tid] = 0.0;
  __syncthreads();
  if (tid == 0) {
    double s7[7];
#pragma unroll
    for (int q = 0; q < 7; q++) s7[q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
    if (want_b) sc[XS_GN2] = s7[0], sc[XS_GGN] = s7[1];
    if (want_c) sc[XS_CCOST] = s7[2], sc[XS_MLIN] = s7[3], sc[XS_MQUAD] = s7[4], sc[XS_DN] = s7[5], sc[XS_XN] = s7[6];
  }
}

// ---------------------------------------------------------------------------
// k_lm_cb2: grid (XP_WGS, batch) x 256 — lfvio_group, behind the sweep of a rank's landmarks: per workgroup (a fixed stride of landmarks)
//   sum_l c_l b_l^2   (c_l = s_l^2 / e_l, the weight of landmark l in the Schur complement)   and
//   the number of landmarks whose diagonal_ entry sits on Ceres' min / max_lm_diagonal clamp,
// into the pair of the workgroup behind the exchange scalars (XOFF_P), which rides in the all-reduce of the reduced system.  With the sums — and the Schur sums every rank holds after the all-reduce —
// the landmark parts of ||gauss_newton_step_||^2 and gradient_ . gauss_newton_step_ are quadratic forms in the camera part N_c of
// the Gauss-Newton direction that EVERY rank evaluates for itself right after the solve (solve_body):
//   y_l = s_l (b_l + w_l . N_c) / e_l,  gauss_newton_l = -d_l y_l,  d_l^2 = s_l^2 a_l,  e_l = s_l^2 a_l (1 + mu)   (no clamp)
//   sum_l gauss_newton_l^2        = ( sum c b^2 + 2 (sum c b w) . N_c + N_c^T (sum c w w^T) N_c ) / (1 + mu)
//   sum_l gradient_l gauss_newton_l = -( sum c b^2 + (sum c b w) . N_c )
// so the dogleg needs no second all-reduce between the solve and the candidate.  A clamped landmark breaks the first identity:
// the pass then falls back to the unconfirmed Gauss-Newton candidate (dogleg_body / decide_body).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lm_cb2(char *base, size_t stride) {
  Slot *S = SLOT(base, stride);
  if (tr_flags(&S->tr).done) return;
  const int tid = threadIdx.x, N = S->N;
  const double *sl = S->scale_l, *av = S->a, *bv = S->b, *ei = S->einv_l;
  double cb2 = 0.0, ncl = 0.0;
  for (int l0 = blockIdx.x * 256 + tid; l0 < N; l0 += 4 * XP_WGS * 256) {  // (the loads of four landmarks in flight)
    double s[4], b[4], a[4], e[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int l = l0 + u * XP_WGS * 256, lc = l < N ? l : l0;
      s[u] = sl[lc], b[u] = l < N ? bv[lc] : 0.0, a[u] = av[lc], e[u] = ei[lc];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const double s2a = s[u] * s[u] * a[u];
      cb2 = fma(s[u] * s[u] * e[u], b[u] * b[u], cb2);
      ncl += (l0 + u * XP_WGS * 256 < N && (s2a < 1e-6 || s2a > 1e32)) ? 1.0 : 0.0;
    }
  }
  __shared__ double red[4][2];
  cb2 = wave_sum(cb2), ncl = wave_sum(ncl);
  if ((tid & 63) == 0) red[tid >> 6][0] = cb2, red[tid >> 6][1] = ncl;
  __syncthreads();
  if (tid < 2) S->xch[XOFF_P + 2 * blockIdx.x + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// k_cost_imu: grid ceil(10 * batch / 64) x 64 — the candidate's IMU factor costs of a resident batch, one lane per factor
// (slot = index / 10): residual, sqrt_info weighting and the squared norm are a thousand serial instructions, which 64
// factors share per wave here.  Candidate 0 only (resident batches do not speculate).
__global__ __launch_bounds__(64) void k_cost_imu(char *base, size_t stride, int count) {
  const int g = blockIdx.x * 64 + threadIdx.x;
  if (g >= count * LFVIO_WINDOW_SIZE) return;
  const int slot = g / LFVIO_WINDOW_SIZE, f = g - slot * LFVIO_WINDOW_SIZE;
  Slot *S = reinterpret_cast<Slot *>(base + stride * (size_t)slot);
  const TRFlags fl = tr_flags(&S->tr);
  if (fl.done | fl.chol_fail) return;
  double c = 0.0;
  if (S->imu_active[f]) {
    const FrameState *x = &S->x[fl.cur ^ 1];
    double rr[15];
    imu_raw_residual(&S->imu[f], S->g, x->pose[f], x->sb[f], x->pose[f + 1], x->sb[f + 1], rr);
    const double *Sq = S->imu_sqrt[f];
#pragma unroll
    for (int r = 0; r < 15; r++) {
      double v = 0;
#pragma unroll
      for (int k = r; k < 15; k++) v = fma(Sq[r * 15 + k], rr[k], v);
      c += v * v;
    }
    c *= 0.5;
  }
  S->pose_cost[f] = c;
}

// ---------------------------------------------------------------------------
// k_decide: grid (1, batch) x 64 — TrustRegionMinimizer bookkeeping.  One iteration per evaluated candidate: the pass holds
// tr->spec_n of them (the steps for radius, radius / 2, radius / 4), and they are taken in that order exactly as Ceres would
// meet them — a rejected step halves the radius and the next candidate is the step for that radius; the walk stops at the
// first accepted, invalid or terminating one.  An accepted candidate beyond the first is copied into the regular slot.
// ---------------------------------------------------------------------------
#ifdef LFVIO_DECIDE_PROFILE
#define DSTAMP(k) STAMP(S, k)
#else
#define DSTAMP(k) do { } while (0)
#endif
// (threads past the first wave — k_decide_gauge — only pass the barrier)
DEV void decide_body(Slot *S) {
  TRState *tr = &S->tr;
  const int lane = threadIdx.x;
  // Everything the kernel needs from the slot header in ONE batch of loads, before the first branch: a load issued
  // behind a branch waits a full memory round trip of its own, and this kernel is nothing but such a chain.
  DSTAMP(20);
  TRHead t = *reinterpret_cast<const TRHead *>(tr);
  const int sharded = S->sharded, max_iter = S->max_iter, nLmBlocks = S->nLmBlocks, nlm = S->N;
  __shared__ int acc_sh;
  if (t.done) return;
  if (sharded && S->xch[XOFF_C + XS_ERR] != 0.0) {
    // a peer of the group failed and said so in the collective behind this pass (group.inc): every rank ends the loop here, in
    // the same pass, instead of waiting in a collective the failed rank would never enter
    if (lane == 0) tr->done = 1, tr->error = LFVIO_ERR_DEVICE;
    return;
  }
  if (sharded == 2 && t.do_schur && !t.chol_fail && t.gn_unconfirmed) {
    // The candidate of this pass was the Gauss-Newton step, formed before its norm was known (dogleg_body: a landmark on the clamp).  With the reduced
    // landmark parts in: the dogleg's own case analysis — if it says "Gauss-Newton step" (it fits the radius), the candidate IS
    // the dogleg step and the pass is decided as always; if not, the pass is void: the totals go into the header, nothing is
    // re-linearized or solved, and the next pass forms the interpolated step from them (the path of a rejected step).
    const double *sc = S->xch + XOFF_C;
    const double gn_sq = t.q[Q_GN_SQ] + sc[XS_GN2], ggn = t.q[Q_GRAD_GN] + sc[XS_GGN];
    double cg, cn, sn;
    dogleg_coeffs(t.grad_sq_total, gn_sq, ggn, t.alpha, t.radius, cg, cn, sn);
    t.gn_sq_total = gn_sq, t.grad_gn_total = ggn;
    if (lane == 0) tr->gn_sq_total = gn_sq, tr->grad_gn_total = ggn;
    if (cg == 0.0 && cn == 1.0) {
      t.cg = 0.0, t.cn = 1.0, t.dogleg_step_norm = sn;
      if (lane == 0) tr->dogleg_step_norm = sn;
    } else {
      if (lane == 0) tr->do_lin = 0, tr->do_schur = 0;
      return;
    }
  }
  if (lane < 64) {
    const int K = decide_candidates(t);
    DecideSums sm;
    decide_sums(S, t, K, sharded, nLmBlocks, lane, sm);
    DSTAMP(21);
    if (lane == 0) {
      const int az = decide_walk(t, sm, K, sharded, max_iter, tr);
      DSTAMP(22);
      TRDecision d;
      decision_from(d, t, az);
      decision_to_header(tr, d);
      acc_sh = az | (t.cur << 8);
    }
  }
  __syncthreads();
  DSTAMP(23);
  const int az = acc_sh & 255;
  if (az > 0 && lane < 64) copy_accepted(S, az, acc_sh >> 8, nlm, lane, 64);
}
__global__ __launch_bounds__(64) void k_decide(char *base, size_t stride) {
  Slot *S = SLOT(base, stride);
  decide_body(S);
  // the loop is closed and the gated gauge fix is the next kernel: no worker may claim this state from here on (kernels_spec.h)
  if (S->spec_on && threadIdx.x == 0 && S->tr.done && S->tail_state == 0) spec_closing(S);
}
