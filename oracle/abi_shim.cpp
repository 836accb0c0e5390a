// abi_shim.cpp — TEST INFRASTRUCTURE ONLY.  The entry points of include/lfvio.h implemented over the CPU oracle, so that
// the host mirror (lf-vio_amd/host) can be linked against the oracle instead of the HIP library and run its whole loop —
// processIMU / processImage / triangulate / optimization / slideWindow — on the CPU.  tests/ compare the trajectory the
// product stack (host mirror + liblfvio_hip.so) writes for a recording with the one this stack writes.  Nothing under
// lf-vio_amd/ loads this library; the product fails without liblfvio_hip.so.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../include/lfvio.h"

extern "C" {
int oracle_solve(const LfvioWindow *w, LfvioSolution *out);
int oracle_marginalize(const LfvioWindow *w, int flag, LfvioPrior *out, double *A_out, double *b_out);
int oracle_optimize(const LfvioWindow *w, int flag, LfvioSolution *sol, LfvioPrior *prior_out, double *seconds);
int oracle_triangulate(const LfvioTriangulateIn *in, double *depth);
int oracle_shift_depth(int n, const double *uv_i, const double *marg_R, const double *marg_P, const double *new_R, const double *new_P,
                       double init_depth, double *depth);
int oracle_preintegrate(const double *acc0, const double *gyr0, const double *ba, const double *bg, int n, const double *dt,
                        const double *acc, const double *gyr, const double *noise, LfvioPreintegration *out);
}

namespace {
struct SlotCopy {  // a window with its arrays owned
  LfvioWindow w;
  std::vector<int> start_frame, obs_offset;
  std::vector<double> inv_depth, point, velocity, cur_td, uv_y, lam_out;
  LfvioPrior prior_in, prior_out;
  LfvioSolution sol;
  bool has_prior_out = false;
  void set(const LfvioWindow *in) {
    w = *in;
    const int N = in->num_landmarks, M = in->num_observations;
    start_frame.assign(in->start_frame, in->start_frame + N), obs_offset.assign(in->obs_offset, in->obs_offset + N + 1);
    inv_depth.assign(in->inv_depth, in->inv_depth + N);
    point.assign(in->obs_point, in->obs_point + 3 * (size_t)M), velocity.assign(in->obs_velocity, in->obs_velocity + 3 * (size_t)M);
    cur_td.assign(in->obs_cur_td, in->obs_cur_td + M), uv_y.assign(in->obs_uv_y, in->obs_uv_y + M);
    w.start_frame = start_frame.data(), w.obs_offset = obs_offset.data(), w.inv_depth = inv_depth.data();
    w.obs_point = point.data(), w.obs_velocity = velocity.data(), w.obs_cur_td = cur_td.data(), w.obs_uv_y = uv_y.data();
    if (in->prior && in->prior->valid) {
      prior_in = *in->prior;
      w.prior = &prior_in;
    } else
      w.prior = nullptr;
    lam_out.assign(N > 0 ? N : 1, 0.0);
  }
};
}  // namespace

struct lfvio_ctx {
  std::string err;
  std::vector<SlotCopy> slots;
  bool pending = false;  // lfvio_batch_optimize_begin without its _finish yet
};

extern "C" {

lfvio_ctx *lfvio_create(int) { return new lfvio_ctx(); }
void lfvio_destroy(lfvio_ctx *c) { delete c; }
const char *lfvio_last_error(const lfvio_ctx *c) { return c ? c->err.c_str() : "null context"; }
const char *lfvio_version(void) { return "lfvio C-ABI over the CPU oracle (test infrastructure)"; }

int lfvio_solve(lfvio_ctx *c, const LfvioWindow *in, LfvioSolution *out) {
  if (!c || !in || !out) return LFVIO_ERR_ARG;
  return oracle_solve(in, out);
}
int lfvio_marginalize(lfvio_ctx *c, const LfvioWindow *in, int flag, LfvioPrior *out) {
  if (!c || !in || !out) return LFVIO_ERR_ARG;
  return oracle_marginalize(in, flag, out, nullptr, nullptr);
}
int lfvio_batch_reserve(lfvio_ctx *c, int batch, int, int) {
  if (!c || batch < 1) return LFVIO_ERR_ARG;
  if ((int)c->slots.size() < batch) c->slots.resize(batch);
  return LFVIO_OK;
}
int lfvio_batch_upload(lfvio_ctx *c, int slot, const LfvioWindow *in) {
  if (!c || !in || slot < 0 || slot >= (int)c->slots.size()) return LFVIO_ERR_ARG;
  c->slots[slot].set(in);
  return LFVIO_OK;
}
int lfvio_batch_optimize(lfvio_ctx *c, int count, int marg_flag) {
  if (!c || count < 0 || count > (int)c->slots.size()) return LFVIO_ERR_ARG;
  for (int s = 0; s < count; s++) {
    SlotCopy &S = c->slots[s];
    S.sol.inv_depth = S.lam_out.data();
    int rc = oracle_optimize(&S.w, marg_flag, &S.sol, &S.prior_out, nullptr);
    if (rc != LFVIO_OK) return rc;
    S.has_prior_out = true;
  }
  return LFVIO_OK;
}
int lfvio_batch_optimize_async(lfvio_ctx *c, int count, int marg_flag) { return lfvio_batch_optimize(c, count, marg_flag); }
int lfvio_batch_sync(lfvio_ctx *c) { return c ? LFVIO_OK : LFVIO_ERR_ARG; }
int lfvio_batch_download(lfvio_ctx *c, int slot, LfvioSolution *sol, LfvioPrior *prior) {
  if (!c || slot < 0 || slot >= (int)c->slots.size()) return LFVIO_ERR_ARG;
  SlotCopy &S = c->slots[slot];
  if (sol) {
    double *lam = sol->inv_depth;
    *sol = S.sol;
    sol->inv_depth = lam;
    if (lam) std::memcpy(lam, S.lam_out.data(), sizeof(double) * S.w.num_landmarks);
  }
  if (prior && S.has_prior_out) *prior = S.prior_out;
  return LFVIO_OK;
}
// chained upload: the prior a begin() left uncollected (the shim's "in flight") or the one given
int lfvio_batch_upload_chained(lfvio_ctx *c, int slot, const LfvioWindow *in, LfvioPrior *prior_io) {
  if (!c || !in || !prior_io || slot < 0 || slot >= (int)c->slots.size()) return LFVIO_ERR_ARG;
  if (c->pending && slot == 0) {
    int rc = lfvio_batch_optimize_finish(c, prior_io);
    if (rc != LFVIO_OK) return rc;
  }
  LfvioWindow w = *in;
  w.prior = prior_io->valid ? prior_io : nullptr;
  return lfvio_batch_upload(c, slot, &w);
}
// ... and the form in which the prior never reaches the caller (the product keeps it on the device): here simply the prior of
// the call "in flight", handed from one slot copy to the next
int lfvio_batch_upload_chained_device(lfvio_ctx *c, int slot, const LfvioWindow *in) {
  if (!c || !in || slot != 0 || c->slots.empty() || !c->pending) return LFVIO_ERR_ARG;
  std::unique_ptr<LfvioPrior> p(new LfvioPrior);
  p->valid = 0;
  int rc = lfvio_batch_optimize_finish(c, p.get());
  if (rc != LFVIO_OK) return rc;
  LfvioWindow w = *in;
  w.prior = p->valid ? p.get() : nullptr;
  return lfvio_batch_upload(c, slot, &w);
}
// the split form: the oracle has nothing to overlap — begin does everything, finish hands the prior over
int lfvio_batch_optimize_begin(lfvio_ctx *c, int marg_flag, LfvioSolution *sol) {
  if (!c || !sol || c->slots.empty()) return LFVIO_ERR_ARG;
  int rc = lfvio_batch_optimize(c, 1, marg_flag);
  if (rc != LFVIO_OK) return rc;
  c->pending = true;
  return lfvio_batch_download(c, 0, sol, nullptr);
}
int lfvio_batch_optimize_finish(lfvio_ctx *c, LfvioPrior *prior) {
  if (!c) return LFVIO_ERR_ARG;
  c->pending = false;
  return prior ? lfvio_batch_download(c, 0, nullptr, prior) : LFVIO_OK;
}
int lfvio_batch_optimize_pending(const lfvio_ctx *c) { return c && c->pending ? 1 : 0; }
void *lfvio_stream(lfvio_ctx *) { return nullptr; }

int lfvio_triangulate(lfvio_ctx *c, const LfvioTriangulateIn *in, double *estimated_depth) {
  if (!c || !in) return LFVIO_ERR_ARG;
  if (in->num_landmarks == 0) return LFVIO_OK;
  return oracle_triangulate(in, estimated_depth);
}
int lfvio_shift_depth(lfvio_ctx *c, int n, const double *uv_i, const double marg_R[9], const double marg_P[3], const double new_R[9],
                      const double new_P[3], double init_depth, double *estimated_depth) {
  if (!c || n < 0) return LFVIO_ERR_ARG;
  if (n == 0) return LFVIO_OK;
  return oracle_shift_depth(n, uv_i, marg_R, marg_P, new_R, new_P, init_depth, estimated_depth);
}
int lfvio_preintegrate(lfvio_ctx *c, int num_intervals, const LfvioImuInterval *in, const double noise[4], LfvioPreintegration *out) {
  if (!c || num_intervals < 0) return LFVIO_ERR_ARG;
  for (int k = 0; k < num_intervals; k++) {
    int rc = oracle_preintegrate(in[k].acc_0, in[k].gyr_0, in[k].linearized_ba, in[k].linearized_bg, in[k].num_samples, in[k].dt, in[k].acc,
                                 in[k].gyr, noise, &out[k]);
    if (rc != 0) return rc;
  }
  return LFVIO_OK;
}
// lfvio_group over the oracle: the "devices" are one CPU — whatever the mask, ONE oracle context answers (the host mirror
// references these entry points; the oracle stack of the tests runs with the default single-device mask)
struct lfvio_group {
  lfvio_ctx ctx;
  std::string err;
};
lfvio_group *lfvio_group_create(unsigned device_mask) { return device_mask ? new lfvio_group() : nullptr; }
lfvio_group *lfvio_group_create_local(int, int shards) { return shards > 0 ? new lfvio_group() : nullptr; }
void lfvio_group_destroy(lfvio_group *g) { delete g; }
lfvio_ctx *lfvio_group_ctx(lfvio_group *g, int i) { return (g && i == 0) ? &g->ctx : nullptr; }
const char *lfvio_group_last_error(const lfvio_group *g) { return g ? g->err.c_str() : "null group"; }
int lfvio_group_solve(lfvio_group *g, const LfvioWindow *in, int marg_flag, LfvioSolution *sol, LfvioPrior *prior) {
  if (!g || !in || !sol) return LFVIO_ERR_ARG;
  if (marg_flag < 0) return oracle_solve(in, sol);
  return oracle_optimize(in, marg_flag, sol, prior, nullptr);
}
}
