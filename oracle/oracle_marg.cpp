// oracle_marg.cpp — CPU restatement of the marginalization half of
// Estimator::optimization() (vins_estimator/src/estimator.cpp:833-1005) and of
// MarginalizationInfo (factor/marginalization_factor.cpp:3-319).
//
// TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (see oracle_solver.cpp header).
// The reference orders parameter blocks by iterating an
// std::unordered_map<long,...> keyed by ADDRESS (marginalization_factor.cpp:177-192,
// 306-315), which is implementation-defined; the canonical order here is
//   dropped:  Pose[0], SpeedBias[0], landmarks anchored at frame 0 (feature order)
//             (MARGIN_SECOND_NEW: Pose[WINDOW_SIZE-1])
//   kept:     poses by frame, speed/bias by frame, ex pose, td
// A = J^T J and b = J^T r are order-independent up to that permutation.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <thread>

#include "oracle_solver.h"

namespace orc {

namespace {

struct Factor {                 // ResidualBlockInfo after Evaluate()
  std::vector<int> cols;        // global column index (into A) of every local Jacobian column
  int nres;
  std::vector<double> J;        // nres x cols.size()
  std::vector<double> r;        // nres
};

inline int key_of(LfvioBlockId id) { return id.kind * 100 + id.frame; }
inline int local_size(int kind) {
  return (kind == LFVIO_BLOCK_POSE || kind == LFVIO_BLOCK_EX_POSE) ? 6 : (kind == LFVIO_BLOCK_SPEEDBIAS ? 9 : 1);
}
inline int global_size(int kind) {
  return (kind == LFVIO_BLOCK_POSE || kind == LFVIO_BLOCK_EX_POSE) ? 7 : (kind == LFVIO_BLOCK_SPEEDBIAS ? 9 : 1);
}
inline const double *state_block(const State &x, LfvioBlockId id) {
  switch (id.kind) {
    case LFVIO_BLOCK_POSE: return x.pose[id.frame];
    case LFVIO_BLOCK_SPEEDBIAS: return x.sb[id.frame];
    case LFVIO_BLOCK_EX_POSE: return x.ex;
    default: return &x.td;
  }
}

}  // namespace

int g_marg_threads = 1;  // NUM_THREADS of marginalization_factor.h:13 when set to 4 (timing variant of the CPU baseline)

int marginalize(const LfvioWindow &w, int flag, LfvioPrior *out, std::vector<double> *A_out, std::vector<double> *b_out) {
  Problem pb(w);
  State x = pb.initial_state();
  const bool has_prior = pb.has_prior;
  const LfvioPrior *pr = has_prior ? w.prior : nullptr;
  const double eps = 1e-8;  // marginalization_factor.h:70

  // ---- which parameter blocks take part, and which are dropped
  std::map<int, LfvioBlockId> present;  // canonical order comes from the sort below
  std::map<int, bool> dropped;
  auto touch = [&](LfvioBlockId id, bool drop) {
    int k = key_of(id);
    present[k] = id;
    if (drop) dropped[k] = true;
  };
  std::vector<int> lm_drop;  // landmarks anchored at frame 0 (MARGIN_OLD)

  if (flag == LFVIO_MARGIN_OLD) {
    if (pr)
      for (int i = 0; i < pr->num_blocks; i++) {
        LfvioBlockId id = pr->blocks[i];
        bool drop = (id.kind == LFVIO_BLOCK_POSE && id.frame == 0) || (id.kind == LFVIO_BLOCK_SPEEDBIAS && id.frame == 0);
        touch(id, drop);  // estimator.cpp:840-846
      }
    if (w.imu[0].sum_dt < 10.0) {  // estimator.cpp:858
      touch({LFVIO_BLOCK_POSE, 0}, true);
      touch({LFVIO_BLOCK_SPEEDBIAS, 0}, true);
      touch({LFVIO_BLOCK_POSE, 1}, false);
      touch({LFVIO_BLOCK_SPEEDBIAS, 1}, false);
    }
    for (int l = 0; l < pb.N; l++) {
      if (w.start_frame[l] != 0) continue;  // estimator.cpp:878-880
      lm_drop.push_back(l);
      int k = w.obs_offset[l + 1] - w.obs_offset[l];
      touch({LFVIO_BLOCK_POSE, 0}, true);
      for (int j = 1; j < k; j++) touch({LFVIO_BLOCK_POSE, j}, false);
      touch({LFVIO_BLOCK_EX_POSE, 0}, false);
      if (pb.est_td) touch({LFVIO_BLOCK_TD, 0}, false);
    }
  } else {
    // MARGIN_SECOND_NEW: only if the prior touches Pose[WINDOW_SIZE-1] (estimator.cpp:942-943)
    bool touches = false;
    if (pr)
      for (int i = 0; i < pr->num_blocks; i++)
        if (pr->blocks[i].kind == LFVIO_BLOCK_POSE && pr->blocks[i].frame == LFVIO_WINDOW_SIZE - 1) touches = true;
    if (!touches) {
      if (pr)
        *out = *pr;
      else
        out->valid = 0;
      return LFVIO_OK;
    }
    for (int i = 0; i < pr->num_blocks; i++) {
      LfvioBlockId id = pr->blocks[i];
      if (id.kind == LFVIO_BLOCK_SPEEDBIAS && id.frame == LFVIO_WINDOW_SIZE - 1) return LFVIO_ERR_ARG;  // ROS_ASSERT :953
      touch(id, id.kind == LFVIO_BLOCK_POSE && id.frame == LFVIO_WINDOW_SIZE - 1);
    }
  }

  // ---- marginalize(): column layout (marginalization_factor.cpp:176-194)
  std::map<int, int> idx;  // block key -> first column
  int pos = 0;
  for (auto &kv : present)
    if (dropped.count(kv.first)) {
      idx[kv.first] = pos;
      pos += local_size(kv.second.kind);
    }
  std::vector<int> lm_col(pb.N, -1);
  for (int l : lm_drop) lm_col[l] = pos++;
  const int m = pos;
  std::vector<LfvioBlockId> kept;
  for (auto &kv : present)
    if (!dropped.count(kv.first)) {
      idx[kv.first] = pos;
      pos += local_size(kv.second.kind);
      kept.push_back(kv.second);
    }
  const int n = pos - m;
  if (pos > 6000) return LFVIO_ERR_ARG;  // dense restatement only
  if (n > LFVIO_MAX_PRIOR_DIM || (int)kept.size() > LFVIO_MAX_PRIOR_BLOCKS) return LFVIO_ERR_ARG;

  // ---- preMarginalize(): evaluate every factor (marginalization_factor.cpp:110-129, 3-69)
  std::vector<Factor> factors;
  if (pr) {  // MarginalizationFactor, no loss
    Factor f;
    f.nres = pr->n;
    const double *params[LFVIO_MAX_PRIOR_BLOCKS];
    for (int i = 0; i < pr->num_blocks; i++) params[i] = state_block(x, pr->blocks[i]);
    f.r.resize(pr->n);
    prior_residual(*pr, params, f.r.data(), nullptr);
    f.cols.assign(pr->n, -1);
    for (int i = 0; i < pr->num_blocks; i++) {
      int c0 = idx[key_of(pr->blocks[i])];
      for (int k = 0; k < local_size(pr->blocks[i].kind); k++) f.cols[pr->block_idx[i] + k] = c0 + k;
    }
    f.J.assign(pr->linearized_jacobians, pr->linearized_jacobians + (size_t)pr->n * pr->n);
    factors.push_back(std::move(f));
  }
  if (flag == LFVIO_MARGIN_OLD) {
    if (w.imu[0].sum_dt < 10.0 && pb.imu_active[0]) {
      Factor f;
      f.nres = 15;
      double Jpi[15 * 7], Jsi[15 * 9], Jpj[15 * 7], Jsj[15 * 9];
      f.r.resize(15);
      imu_evaluate(w.imu[0], pb.imu_sqi[0], w.g, x.pose[0], x.sb[0], x.pose[1], x.sb[1], f.r.data(), Jpi, Jsi, Jpj, Jsj);
      f.J.resize(15 * 30);
      f.cols.resize(30);
      for (int rr = 0; rr < 15; rr++) {
        for (int c = 0; c < 6; c++) f.J[rr * 30 + c] = Jpi[rr * 7 + c];
        for (int c = 0; c < 9; c++) f.J[rr * 30 + 6 + c] = Jsi[rr * 9 + c];
        for (int c = 0; c < 6; c++) f.J[rr * 30 + 15 + c] = Jpj[rr * 7 + c];
        for (int c = 0; c < 9; c++) f.J[rr * 30 + 21 + c] = Jsj[rr * 9 + c];
      }
      int p0 = idx[key_of({LFVIO_BLOCK_POSE, 0})], s0 = idx[key_of({LFVIO_BLOCK_SPEEDBIAS, 0})];
      int p1 = idx[key_of({LFVIO_BLOCK_POSE, 1})], s1 = idx[key_of({LFVIO_BLOCK_SPEEDBIAS, 1})];
      for (int c = 0; c < 6; c++) f.cols[c] = p0 + c, f.cols[15 + c] = p1 + c;
      for (int c = 0; c < 9; c++) f.cols[6 + c] = s0 + c, f.cols[21 + c] = s1 + c;
      factors.push_back(std::move(f));
    }
    const int nvf = (int)pb.vf.size();
    for (int k = 0; k < nvf; k++) {
      int l = pb.vf_lm[k];
      if (pb.vf_i[k] != 0) continue;
      int fj = pb.vf_j[k];
      Factor f;
      f.nres = 2;
      f.r.resize(2);
      double Ji[14], Jj[14], Jex[14], Jf[2], Jtd[2] = {0, 0};
      // ResidualBlockInfo::Evaluate asks for every Jacobian (ex pose included even when it is constant in the solve)
      visual_evaluate(pb.vf[k], pb.est_td, w.tr, w.row, w.sqrt_info, x.pose[0], x.pose[fj], x.ex, x.lam[l], x.td, f.r.data(),
                      Ji, Jj, Jex, Jf, pb.est_td ? Jtd : nullptr);
      int nc = pb.est_td ? 20 : 19;
      f.J.resize(2 * nc);
      f.cols.resize(nc);
      int p0 = idx[key_of({LFVIO_BLOCK_POSE, 0})], pj = idx[key_of({LFVIO_BLOCK_POSE, fj})];
      int pe = idx[key_of({LFVIO_BLOCK_EX_POSE, 0})];
      for (int rr = 0; rr < 2; rr++) {
        for (int c = 0; c < 6; c++) {
          f.J[rr * nc + c] = Ji[rr * 7 + c];
          f.J[rr * nc + 6 + c] = Jj[rr * 7 + c];
          f.J[rr * nc + 12 + c] = Jex[rr * 7 + c];
        }
        f.J[rr * nc + 18] = Jf[rr];
        if (pb.est_td) f.J[rr * nc + 19] = Jtd[rr];
      }
      for (int c = 0; c < 6; c++) f.cols[c] = p0 + c, f.cols[6 + c] = pj + c, f.cols[12 + c] = pe + c;
      f.cols[18] = lm_col[l];
      if (pb.est_td) f.cols[19] = idx[key_of({LFVIO_BLOCK_TD, 0})];
      corrector_apply(f.r.data(), 2, f.J.data(), nc);  // loss_function = CauchyLoss(1.0), estimator.cpp:896
      factors.push_back(std::move(f));
    }
  }

  // ---- ThreadsConstructA: factors dealt round-robin to NUM_THREADS = 4 accumulators,
  //      summed thread 3 -> 0 (marginalization_factor.cpp:141-172, 232-261)
  const int NT = 4;
  std::vector<std::vector<double>> At(NT, std::vector<double>((size_t)pos * pos, 0.0));
  std::vector<std::vector<double>> bt(NT, std::vector<double>(pos, 0.0));
  // accumulator t takes factors t, t+4, ... in order, so running the four on their own threads (the reference's
  // pthreads; g_marg_threads = 4) gives bit-identical sums to running them one after the other (= 1, the default)
  auto construct = [&](int t) {
    std::vector<double> &Aa = At[t];
    std::vector<double> &ba = bt[t];
    for (size_t fi = t; fi < factors.size(); fi += NT) {
      const Factor &f = factors[fi];
      const int nc = (int)f.cols.size();
      for (int c1 = 0; c1 < nc; c1++) {
        if (f.cols[c1] < 0) continue;
        double g = 0;
        for (int rr = 0; rr < f.nres; rr++) g += f.J[(size_t)rr * nc + c1] * f.r[rr];
        ba[f.cols[c1]] += g;
        for (int c2 = 0; c2 < nc; c2++) {
          if (f.cols[c2] < 0) continue;
          double s = 0;
          for (int rr = 0; rr < f.nres; rr++) s += f.J[(size_t)rr * nc + c1] * f.J[(size_t)rr * nc + c2];
          Aa[(size_t)f.cols[c1] * pos + f.cols[c2]] += s;
        }
      }
    }
  };
  if (g_marg_threads >= NT) {
    std::thread th[NT];
    for (int t = 0; t < NT; t++) th[t] = std::thread(construct, t);
    for (int t = 0; t < NT; t++) th[t].join();
  } else {
    for (int t = 0; t < NT; t++) construct(t);
  }
  std::vector<double> A((size_t)pos * pos, 0.0), b(pos, 0.0);
  for (int t = NT - 1; t >= 0; t--) {
    for (size_t i = 0; i < A.size(); i++) A[i] += At[t][i];
    for (int i = 0; i < pos; i++) b[i] += bt[t][i];
  }
  At.clear();

  // ---- Schur complement with eigen-decomposition pseudo-inverse (:267-281)
  std::vector<double> Amm((size_t)m * m), dm(m), Vm((size_t)m * m), Amm_inv((size_t)m * m, 0.0);
  for (int i = 0; i < m; i++)
    for (int j = 0; j < m; j++) Amm[(size_t)i * m + j] = 0.5 * (A[(size_t)i * pos + j] + A[(size_t)j * pos + i]);
  if (m > 0) sym_eig(Amm.data(), m, dm.data(), Vm.data());
  for (int k = 0; k < m; k++) {
    if (!(dm[k] > eps)) continue;
    double inv = 1.0 / dm[k];
    for (int i = 0; i < m; i++) {
      double vi = Vm[(size_t)i * m + k] * inv;
      if (vi == 0.0) continue;
      for (int j = 0; j < m; j++) Amm_inv[(size_t)i * m + j] += vi * Vm[(size_t)j * m + k];
    }
  }
  // A = Arr - Arm Amm_inv Amr ; b = brr - Arm Amm_inv bmm
  std::vector<double> T((size_t)n * m, 0.0);  // Arm * Amm_inv
  for (int i = 0; i < n; i++)
    for (int k = 0; k < m; k++) {
      double a = A[(size_t)(m + i) * pos + k];
      if (a == 0.0) continue;
      for (int j = 0; j < m; j++) T[(size_t)i * m + j] += a * Amm_inv[(size_t)k * m + j];
    }
  std::vector<double> Ar((size_t)n * n), br(n);
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < n; j++) {
      double s = 0;
      for (int k = 0; k < m; k++) s += T[(size_t)i * m + k] * A[(size_t)k * pos + (m + j)];
      Ar[(size_t)i * n + j] = A[(size_t)(m + i) * pos + (m + j)] - s;
    }
    double s = 0;
    for (int k = 0; k < m; k++) s += T[(size_t)i * m + k] * b[k];
    br[i] = b[m + i] - s;
  }
  if (A_out) *A_out = Ar;
  if (b_out) *b_out = br;

  // ---- second eigen-decomposition -> linearized_jacobians / residuals (:283-291)
  std::vector<double> S(n), V2((size_t)n * n);
  sym_eig(Ar.data(), n, S.data(), V2.data());
  std::memset(out, 0, sizeof *out);
  out->valid = 1;
  out->m = m;
  out->n = n;
  for (int k = 0; k < n; k++) {
    double s = S[k] > eps ? S[k] : 0.0;
    double sinv = S[k] > eps ? 1.0 / S[k] : 0.0;
    double s_sqrt = std::sqrt(s), sinv_sqrt = std::sqrt(sinv);
    double vb = 0;
    for (int i = 0; i < n; i++) {
      out->linearized_jacobians[(size_t)k * n + i] = s_sqrt * V2[(size_t)i * n + k];
      vb += V2[(size_t)i * n + k] * br[i];
    }
    out->linearized_residuals[k] = sinv_sqrt * vb;
  }

  // ---- getParameterBlocks with addr_shift (:299-319; estimator.cpp:921-933, 969-993)
  out->num_blocks = (int)kept.size();
  for (size_t i = 0; i < kept.size(); i++) {
    LfvioBlockId id = kept[i];
    const double *data = state_block(x, id);  // parameter_block_data snapshot (:121-126)
    for (int k = 0; k < global_size(id.kind); k++) out->block_x0[i][k] = data[k];
    out->block_idx[i] = idx[key_of(id)] - m;
    LfvioBlockId shifted = id;
    if (flag == LFVIO_MARGIN_OLD) {
      if (id.kind == LFVIO_BLOCK_POSE || id.kind == LFVIO_BLOCK_SPEEDBIAS) shifted.frame = id.frame - 1;
    } else {
      if ((id.kind == LFVIO_BLOCK_POSE || id.kind == LFVIO_BLOCK_SPEEDBIAS) && id.frame == LFVIO_WINDOW_SIZE)
        shifted.frame = LFVIO_WINDOW_SIZE - 1;
    }
    out->blocks[i] = shifted;
  }
  return LFVIO_OK;
}

}  // namespace orc
