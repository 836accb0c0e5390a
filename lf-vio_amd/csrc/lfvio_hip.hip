// lfvio_hip.hip — C-ABI implementation (include/lfvio.h) on HIP for gfx950 (MI355X).
//
// Host side of the hot path of LF-VIO's Estimator::optimization()
// (vins_estimator/src/estimator.cpp:676-1009): packs one LfvioWindow into the device
// layout of dev_types.h, launches the kernel pipeline and unpacks the result.
// There is NO CPU fallback: without a usable HIP device lfvio_create() fails.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
// The few RCCL types group.inc needs, declared here with the values of the public ABI (rccl.h: ncclSuccess = 0, ncclSum = 0,
// ncclDouble = 8, a 128-byte unique id): librccl is dlopen()ed by group.inc, not linked, and the single-GPU library builds on
// a ROCm install without the RCCL development headers.
typedef struct ncclComm *ncclComm_t;
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct {
  char internal[NCCL_UNIQUE_ID_BYTES];
} ncclUniqueId;
typedef int ncclResult_t;
typedef int ncclDataType_t;
typedef int ncclRedOp_t;
constexpr ncclResult_t ncclSuccess = 0;
constexpr ncclRedOp_t ncclSum = 0;
constexpr ncclDataType_t ncclDouble = 8;

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/lfvio.h"
#include "../../include/lfvio_debug.h"
#include "kernels_marg.h"
#include "kernels_solve.h"
#include "kernels_feat.h"
#include "kernels_linw.h"
#include "linb_plan.h"
#include "kernels_stepw.h"

#define HIPCHK(ctx, call)                                                                      \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                          \
      return LFVIO_ERR_DEVICE;                                                                 \
    }                                                                                          \
  } while (0)

namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Layout {  // byte offsets inside one slot blob, by capacity
  int maxN = 0, maxM = 0;
  int capLmBlocks = 0, capChunks = 0, capSchurParts = 0;
  size_t in_begin = 0, in_end = 0, total = 0;
  size_t anc[8], pmo[8], pm_pair, linb_lm0, linb_ns, linw_begin = 0, linw_end = 0;  // k_linw's copies of the observations (behind the regular inputs: uploaded on their own, resident batches only)
  size_t lm_start, lm_cnt, lm_obs0, lm_perm, lm_woff, lam0, obs[8], pm_obs, pm_lm, chunk_pair, chunk_begin, chunk_end, sum_off, sum_end_marg, sum_items, prior_J,
      prior_r;
  size_t lam[2], lamE[SPEC_EXTRA], cost_partE, prior_A, a, b, W, Wt, scale_l, grad_l, gn_l, diag_l, einv_l, d1, d2, gram_part, pairG, schur_part,
      xch, lm_part, cost_part, imu_out, imu_raw, mscr, eig_aux;
};

Layout make_layout(int maxN, int maxM) {
  Layout L;
  L.maxN = maxN;
  L.maxM = maxM;
  L.capLmBlocks = std::max(1, (maxN + LM_BLOCK - 1) / LM_BLOCK);
  L.capChunks = 64 + maxM / CHUNK_MAX;
  // (one part per landmark workgroup of k_lin: 64 landmarks each, or 32 for windows of at most SPEC_MAX_LM landmarks — Slot::lm_half)
  L.capSchurParts = std::max(L.capLmBlocks + 1, 2 * ((std::min(maxN, SPEC_MAX_LM) + LM_BLOCK - 1) / LM_BLOCK));
  static_assert(LINB_LEN >= SCHUR_LEN, "the Schur partials of k_lin share the array of k_linb's group partials, which are the larger");
  size_t o = align_up(sizeof(Slot), 256);
  auto take = [&](size_t bytes) {
    size_t r = o;
    o = align_up(o + bytes, 256);
    return r;
  };
  const size_t N = std::max(maxN, 1), M = std::max(maxM, 1), LB = (size_t)L.capLmBlocks * LM_BLOCK;
  L.in_begin = o;
  L.lm_start = take(N * 4), L.lm_cnt = take(N * 4), L.lm_obs0 = take(N * 4), L.lm_perm = take(N * 4);
  L.lm_woff = take((N + 1) * 4);
  L.lam0 = take(N * 8);
  for (int k = 0; k < 8; k++) L.obs[k] = take(M * 8);
  L.pm_obs = take(M * 4), L.pm_lm = take(M * 4);
  L.chunk_pair = take((size_t)L.capChunks * 4), L.chunk_begin = take((size_t)L.capChunks * 4),
  L.chunk_end = take((size_t)L.capChunks * 4);
  // (the two arrays whose used part varies most come last, so that an upload copies [in_begin, used end of sum_items)
  // and the n x n the prior really has — 0.2 MB instead of 0.7 MB at 300 landmarks)
  L.prior_r = take(LFVIO_MAX_PRIOR_DIM * 8);
  L.sum_off = take((size_t)(SUM_VIS + 1) * 4), L.sum_end_marg = take((size_t)SUM_VIS * 4);
  L.sum_items = take((size_t)SUM_ITEMS_CAP * 4);
  L.prior_J = take((size_t)LFVIO_MAX_PRIOR_DIM * LFVIO_MAX_PRIOR_DIM * 8);
  L.in_end = o;
  L.linw_begin = o;
  for (int k = 0; k < 8; k++) L.anc[k] = take(N * 8);
  for (int k = 0; k < 8; k++) L.pmo[k] = take(M * 8);
  L.pm_pair = take(M);
  L.linb_lm0 = take(((size_t)L.capLmBlocks + LFVIO_NUM_FRAMES + 1) * 4), L.linb_ns = take(((size_t)L.capLmBlocks + LFVIO_NUM_FRAMES + 1) * 4);  // (at most a group per strip)
  L.linw_end = o;
  L.lam[0] = take(LB * 8), L.lam[1] = take(LB * 8);
  for (int k = 0; k < SPEC_EXTRA; k++) L.lamE[k] = take((size_t)SPEC_MAX_LM * 8);
  L.cost_partE = take((size_t)SPEC_EXTRA * (SPEC_MAX_LM / 64) * LMS * 8);
  L.prior_A = take((size_t)LFVIO_MAX_PRIOR_DIM * LFVIO_MAX_PRIOR_DIM * 8);
  L.a = take(LB * 8), L.b = take(LB * 8), L.W = take(LB * WLD * 8);
  L.Wt = take((size_t)WT_PAIRS * std::max((size_t)SPEC_MAX_LM, LB) * 16);  // (a large window's rows are LB apart: k_linb)
  L.scale_l = take(LB * 8), L.grad_l = take(LB * 8), L.gn_l = take(LB * 8), L.diag_l = take(LB * 8);
  L.einv_l = take(LB * 8), L.d1 = take(LB * 8), L.d2 = take(LB * 8);
  L.gram_part = take((size_t)L.capChunks * NGP * 8);
  L.pairG = take((size_t)NPAIR * NGP * 8);
  L.schur_part = take((size_t)L.capSchurParts * LINB_LEN * 8);  // (k_linb's partials live there too: LINB_LEN > SCHUR_LEN doubles per group, at most a group per strip)
  L.xch = take((size_t)XCH_ALLOC * 8);
  L.lm_part = take((size_t)L.capLmBlocks * LMS * 8);
  L.cost_part = take((size_t)L.capLmBlocks * LMS * 8);
  L.imu_out = take((size_t)LFVIO_WINDOW_SIZE * IMU_OUT * 8);
  L.imu_raw = take((size_t)LFVIO_WINDOW_SIZE * IMU_RAW * 8);
  L.mscr = take((size_t)HPP_CAP * 8);
  L.eig_aux = take(4096);
  L.total = align_up(o, 4096);
  return L;
}

size_t down_bytes(int maxN) { return sizeof(FrameState) * 2 + sizeof(TRState) + 2 * ((size_t)maxN * 8 + 128) + sizeof(LfvioPrior) + 4096; }

struct SlotHostInfo {
  bool spec_on = false;  // Slot::spec_on of the resident window
  // The uploaded state holds a quaternion off the unit sphere (struct Tab, dev_types.h: |q|^2 further than TAB_OFF_SPHERE from 1): the
  // sweep that linearizes at that state — the first pass of a call — and, where the off-sphere quaternion is an extrinsic that is not
  // estimated, every sweep of the window run the instantiations that know the reference's two back-rotations
  bool offs_first = false, offs_all = false;
  int N = 0, M = 0, gLm = 0, gLw = 0, gCh = 0, gSc = 0;  // gLm: landmark blocks of 64; gLw: landmark workgroups of k_lin
  int marg_n = 0;            // the largest prior (tangent rows) a marginalization of this window can produce
  int prior_n = 0;           // rows of the uploaded window's input prior (0: none)
  int max_iter = 0;          // LfvioWindow::max_num_iterations of the uploaded window
  double max_seconds = -1.0;  // LfvioWindow::max_solver_time_in_seconds (<= 0: no cap)
  std::vector<int> perm;  // device order -> caller order
  bool linw_ok = false;    // the window carries a LinwPlan and its arrays (kernels_linw.h)
  bool linb_ok = false;    // ... as a group list: a large single window (k_linb)
  int linb_ng = 0;
  bool uploaded = false;   // the slot's work-array pointers are on the device (until the next reserve())
  bool resident = false;   // a window is resident: N, perm, grid sizes below describe what the device holds.  Cleared while an upload
                           // rewrites them and set again when its copies are enqueued, so a refused upload leaves a slot that every
                           // later call refuses instead of one described by the wrong sizes
  LfvioPrior in_prior;    // kept for the "prior passes through" case of MARGIN_SECOND_NEW
  bool has_in_prior = false;
  bool in_prior_device = false;  // ... whose values never came to the host (lfvio_batch_upload_chained_device): in_prior holds the structure only
  // what a marginalization of this window leaves as its prior, as planned at the upload [MARGIN_OLD, MARGIN_SECOND_NEW]: the
  // structure of the NEXT window's input prior when that window takes it over on the device
  struct MargOut {
    int valid = 0, n = 0, m = 0, nb = 0;
    int kind[LFVIO_MAX_PRIOR_BLOCKS], frame[LFVIO_MAX_PRIOR_BLOCKS], idx[LFVIO_MAX_PRIOR_BLOCKS];
  } marg_out[2];
  int mail_seq = 0;  // Slot::mail_seq of the resident window: what its mailbox flags are set to
  // k_sum's gather lists on the device are a function of the chunks per frame pair alone: kept from one upload of the slot
  // to the next while that table stays the same (consecutive windows of one estimator: nearly always)
  std::vector<int> list_key;  // pair_chunk0[0 .. NPAIR], pre_gram
  int list_items = -1;
};

}  // namespace

struct lfvio_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  char *d_feat = nullptr;  // scratch of lfvio_triangulate / lfvio_shift_depth (grow-only)
  size_t feat_bytes = 0;
  char *h_feat = nullptr;  // pinned mirror of d_feat: the inputs of a feature step go up as ONE copy, its outputs come down as one
  std::string err;
  int batch = 0;
  Layout L;
  char *d_base = nullptr;
  char *h_stage = nullptr;  // pinned, one slot's header + inputs
  char *h_down = nullptr;   // pinned download buffer
  size_t h_down_bytes = 0;
  std::vector<SlotHostInfo> info;
  std::vector<int> perm_build;  // upload_window builds the next permutation here and swaps it in at its commit point
  // cached graph of the solve loop
  hipGraphExec_t graph = nullptr;
  int g_batch = 0, g_lm = 0, g_ch = 0, g_sc = 0, g_iters = 0, g_linw = 0, g_offs = 0;
  // cached graph of one chunk of passes (synchronous entry points: the loop is launched chunk by chunk)
  // [publish]: the variants whose gated gauge fix / marginalization also push state and prior into the caller's mailbox
  // (lfvio_batch_optimize_begin) — a kernel argument, so the plain call pays nothing for the split one
  bool publish = false;
  hipGraphExec_t chunk[2] = {}, tail[2][3] = {};  // chunk[speculation variant]
  hipGraphExec_t first[2][2][4][13] = {};  // [speculation variant: 0 three candidates (or none), 1 four][publish][0: solve only, 1 + flag: with the gated tail][passes in the first graph]
  int fixed_passes = 0;             // debug (lfvio_debug_configure "first_passes"): > 0 sizes every first graph with this many passes
  int recent_passes[4] = {0, 0, 0, 0}, recent_head = 0;  // passes of the last four synchronous calls (predict())
  bool predicted_early = false;     // lfvio_batch_optimize_begin has fed this call's pass count to predict() already (the join / finish that follows must not again)
  int predict_passes = 4;           // passes the first graph of the next call carries: the most any of the last four calls needed (predict()); tail[flag]: force-done + gated gauge fix + marginalization
  double pass_seconds = 2e-4;       // measured duration of one pass of a continuation chunk (sizes the first graph of a call with a wall-clock cap)
  int k_batch = 0, k_lm = 0, k_ch = 0, k_sc = 0, k_spec = 0, k_linw = 0, k_offs = 0;
  int k_setup = -1;  // bits of k_setup's launch the graphs were captured with (setup_launch)  // (k_offs: slots_offs of the captured launches)
  int *d_pending = nullptr, *h_pending = nullptr;  // number of slots whose trust-region loop is not done
  // lfvio_batch_optimize_begin / _finish: the solution of slot 0 arrives in host memory the gated gauge fix writes directly
  // (Slot::mail, dev_types.h MAIL_*) while the marginalization of the same graph is still running; `inflight` from the moment
  // begin has returned on that flag until the stream has been synchronized again (join_inflight: every entry point that
  // touches the slots or the stream starts with it).  The landmark-parallel steps either side of optimization()
  // (triangulate / shift_depth / preintegrate) have their own stream and scratch and do not wait for the tail.
  char *h_mail = nullptr, *d_mail = nullptr;
  hipStream_t fstream = nullptr;
  // The marginalization run ahead of the loop's end (kernels_spec.h): a SHADOW slot behind the last one (contexts of one small
  // window), workers on a stream of their own (lowest priority: a third pool of hardware queues), one graph of SIDE_ROUNDS rounds
  // per marginalization flag and hand-over variant.  mail[6]: the ticket of the call in flight, echoed into mail[7] by the worker
  // that delivers its prior.
  static constexpr int WORKERS = 2;  // two, so that the newest accepted state never waits for the round of an older one to notice it is stale
  hipStream_t sstream[WORKERS] = {};
  bool shadow = false;          // the blob has batch + WORKERS slots, the last ones the shadows of slot 0
  bool marg_ahead = true;       // debug (lfvio_debug_configure): off = every call ends with the serial tail
  hipGraphExec_t side[WORKERS][2][2] = {};  // [worker][marg_flag][publish]
  int side_ticket = 0;          // last ticket handed out
  bool side_launched = false;   // workers were started for the call in flight (or just finished)
  bool side_known = false;      // ... and h_pending[2] holds its tail_state (the first graph carried the tail)
  long long stat_ahead_calls = 0, stat_ahead_hits = 0;  // calls with workers | priors a worker delivered
  bool inflight = false;
  bool unsynced = false;        // finish() took the prior from the mailbox and left the last microseconds of the graph to the next join
  // a prior collected on the caller's behalf because the slots had to be re-allocated while its call was in flight
  // (reserve): lfvio_batch_optimize_finish / lfvio_batch_upload_chained hand it over
  // the copies out of h_stage are awaited by the NEXT user of the staging block, not by the upload that enqueued them
  hipEvent_t stage_event = nullptr;
  bool stage_busy = false;
  double up_us[4] = {0, 0, 0, 0};  // last upload: host packing | collecting the chained prior | prior + copies enqueued | final synchronization (lfvio_debug_upload_times)
  std::unique_ptr<LfvioPrior> held;
  bool has_held = false;
  int inflight_flag = 0;        // marg_flag of the call in flight
  bool pipelined = false;       // the window resident in slot 0 was uploaded behind a marginalization still running (lfvio_batch_upload_chained_device):
                                // the stream holds work nobody has waited for; the next lfvio_batch_optimize_begin goes out behind it without a wait
  int mail_seq = 0;             // sequence number of the last upload (Slot::mail_seq)
  std::unique_ptr<LfvioPrior> chain_struct;  // the structure-only prior of a device-chained upload
  bool chain_err_told = false;     // the call in flight ran without the prior it was promised and its begin() has said so
  bool debug_break_chain = false;  // lfvio_debug_break_next_chain: the next device-chained upload promises a prior of another size than the device will find
  bool inflight_first = false;  // the flag came out of the first graph: {tail_state, passes_used} land in h_pending[2..3] when it ends
  bool use_graph = true;
  int stat_chunks = 0;  // graph launches of the last synchronous solve loop (debug)
  int last_passes = 0;  // passes of the trust-region loop the last synchronous call used (slowest slot)
  int last_iters = 0;   // ... and, for one window, the iterations they covered: a window whose steps are mostly rejected gets more speculative candidates per pass
  bool fixed_spec = false;  // lfvio_debug_configure "spec_count": a fixed number of candidates (measurements)
  int spec_count = 3;   // candidates per pass of a speculating launch: radius, radius / 2, radius / 4 (, radius / 8)
  int *d_lwt = nullptr;  // static table of k_linw's phase 3 (kernels_linw.h LWT_*)
  int *d_asm = nullptr;  // static scatter table of k_solve_dense<true> (kernels_solve.h ASM_*)
  bool shard_group = false;  // the resident shard is driven by an lfvio_group (two collectives per pass: shard.inc phase 4)
  int shard_kmax0 = -1;  // lfvio_shard_begin: the longest track among the WHOLE window's frame-0 landmarks (0: none), for the marginalization's plan
  int lm_half = 1;       // windows of at most SPEC_MAX_LM landmarks: 8 lanes per track in k_lin's landmark role (lfvio_debug_configure "lm_half" 0: the 4-lane form)
  int linw_mode = 1;     // 1: resident batches linearize with k_linw when every slot of the launch carries a plan; 0: never (k_lin roles + k_sum);
                         // 2: any launch of planned windows, however few (tests).  lfvio_debug_configure "linw"
  double init_radius = 1e4;  // initial_trust_region_radius of the windows uploaded from now on (lfvio_debug_configure "initial_radius")
  double fn_tol = 1e-6;  // function_tolerance of the windows uploaded from now on (lfvio_debug_configure "function_tolerance")
  bool force_eig = false;  // debug: k_marg_solve takes the eigen-decomposition path for the dropped block even when the Cholesky path applies
  // landmark-sharded mode (multi-GPU)
  bool shard_active = false;
  int shard_begin = 0, shard_end = 0, shard_state = 0;
  std::vector<int> sh_start, sh_off;
  // stream-ordered sharded driver: ring of pinned flag records, one per enqueued decision (shard.inc)
  static constexpr int FLAG_RING = 4;
  int *h_flags = nullptr;
  hipEvent_t flag_event[FLAG_RING] = {};
  long long flag_head = 0, flag_tail = 0;
};

namespace {

// keep_side: the workers' graphs (kernels_spec.h) stay — they are captured for the slot's CAPACITY, not for the resident window's launch
// dimensions, and a stream of windows changes those every few frames
void destroy_graph(lfvio_ctx *c, bool keep_side = false) {
  // (a graph whose tail is still running behind an early state is not destroyed under it)
  if (c->stream && (c->inflight || c->unsynced || c->pipelined)) {
    (void)hipStreamSynchronize(c->stream);
    c->unsynced = false, c->pipelined = false;
  }
  if (c->graph) {
    (void)hipGraphExecDestroy(c->graph);
    c->graph = nullptr;
  }
  if (!keep_side) {
    for (auto &st : c->sstream)
      if (st) (void)hipStreamSynchronize(st);  // (rounds that found nothing to do may still be draining)
    for (auto &wk : c->side)
      for (auto &row : wk)
        for (auto &t : row)
          if (t) (void)hipGraphExecDestroy(t), t = nullptr;
  }
  for (auto &t : c->chunk)
    if (t) (void)hipGraphExecDestroy(t), t = nullptr;
  for (auto &row : c->tail)
    for (auto &t : row)
      if (t) (void)hipGraphExecDestroy(t), t = nullptr;
  for (auto &var : c->first)
    for (auto &plane : var)
      for (auto &row : plane)
        for (auto &t : row)
          if (t) (void)hipGraphExecDestroy(t), t = nullptr;
}

// The tail of a call whose solution went out early (lfvio_batch_optimize_begin) is still on the stream: wait for it and
// take the bookkeeping its graph left in the pinned block.  First statement of everything that touches the slots.
// pipelined_ok (lfvio_batch_optimize_begin only): a window uploaded behind a marginalization that is still running
// (lfvio_batch_upload_chained_device) is optimized behind it too — everything is ordered by the stream, nothing is waited for.
// The first graph of the next call is sized from the calls before it.  A pass the window does not need returns at once (~15 us of
// launches for the four kernels); a pass it needs and the graph does not carry is a round trip to the host and another graph
// launch (~60 us).  A resident window re-solved again and again needs the same number every time; the windows of a stream vary by
// one or two passes from frame to frame (4 .. 8 on the synthetic stream): the most any of the last four calls needed covers nearly
// all of them (measured on the stream: 0.881 ms per window with the last call's count, 0.862 with a fixed 8).
void predict(lfvio_ctx *c) {
  c->recent_passes[c->recent_head++ & 3] = c->last_passes;
  int m = 1;
  for (int k = 0; k < 4; k++) m = std::max(m, c->recent_passes[k]);
  c->predict_passes = m;
}
// The call whose loop has just ended on stream 0 left its prior to a worker on the second stream (tail_state 3, kernels_spec.h): wait
// for the worker's echo of the call's ticket — it follows the prior's arrival in slot 0 (and in the mailbox) behind a system-scope
// fence.  Stream 0 is idle when this is called.
int wait_side(lfvio_ctx *c) {
  if (!c->side_launched) return LFVIO_OK;
  c->side_launched = false;
  int ts = 0;
  if (c->side_known) ts = c->h_pending[2];
  else HIPCHK(c, hipMemcpy(&ts, c->d_base + offsetof(Slot, tail_state), sizeof ts, hipMemcpyDeviceToHost));  // (a tail graph of its own: rare)
  const int *echo = (const int *)c->h_mail + 7;
  if (ts == 3) {
    const auto t0 = std::chrono::steady_clock::now();
    while (__atomic_load_n(echo, __ATOMIC_ACQUIRE) != c->side_ticket) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) {
        c->err = "the marginalization handed to a worker stream did not arrive";
        return LFVIO_ERR_DEVICE;
      }
    }
  }
  if (__atomic_load_n(echo, __ATOMIC_ACQUIRE) == c->side_ticket) c->stat_ahead_hits++;
  ((volatile int *)c->h_mail)[6] = 0;  // (whatever runs k_setup on this slot next without workers of its own — a standalone marginalization, a debug sweep — publishes nothing)
  return LFVIO_OK;
}
constexpr const char *CHAIN_ERR_TEXT = "the prior this window was to take over on the device was not there (the marginalization before it produced none): it ran without a prior";
int join_inflight(lfvio_ctx *c, bool pipelined_ok = false) {
  if (c->pipelined && !c->inflight && pipelined_ok) return LFVIO_OK;
  if (c->unsynced || c->pipelined) {
    c->unsynced = false, c->pipelined = false;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->inflight)
      if (int rc = wait_side(c)) return rc;
  }
  if (!c->inflight) return LFVIO_OK;
  c->inflight = false;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const bool handed = c->side_launched && c->side_known && c->h_pending[2] == 3;  // the prior was a worker's (kernels_spec.h)
  if (int rc = wait_side(c)) return rc;
  if (c->inflight_first) {
    c->last_passes = std::max(c->h_pending[3], 1), c->last_iters = c->h_pending[4];
    if (!c->predicted_early) predict(c);
    c->predicted_early = false;
    if (c->h_pending[2] != 2 && !handed) {
      c->err = "the marginalization behind an early solution did not finish";
      return LFVIO_ERR_DEVICE;
    }
    const bool told = c->chain_err_told;  // (lfvio_batch_optimize_begin has reported it with the state: once is enough)
    c->chain_err_told = false;
    if (c->h_pending[5] && !told) {
      c->err = CHAIN_ERR_TEXT;
      return LFVIO_ERR_DEVICE;
    }
  }
  return LFVIO_OK;
}

// After a graph launch: the first of "the solution is in the mailbox" (true) and "everything enqueued has run" (false: the
// window was not done within these passes and the gated gauge fix did not run, or there is no mailbox for this window).
// (the flag is the sequence number of the resident window's upload: a window optimized behind the marginalization of the one
// before must not take that one's late prior flag for its own)
bool wait_early(lfvio_ctx *c) {
  int *flag = (int *)c->h_mail;
  const int want = c->info[0].mail_seq;
  for (;;) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == want) return true;
    if (hipStreamQuery(c->stream) != hipErrorNotReady) return __atomic_load_n(flag, __ATOMIC_ACQUIRE) == want;
  }
}

constexpr int SHADOW_MAX_LM = 1024;  // capacity (with reserve()'s headroom) up to which a one-window context carries a shadow slot
int reserve(lfvio_ctx *c, int batch, int maxN, int maxM) {
  if (c->d_base && batch <= c->batch && maxN <= c->L.maxN && maxM <= c->L.maxM) return LFVIO_OK;  // (the usual case: nothing to wait for)
  if (c->inflight) {  // the slot that holds the prior of the call in flight is about to be freed: collect it first
    if (!c->held) c->held.reset(new LfvioPrior);
    if (int rc = lfvio_batch_optimize_finish(c, c->held.get())) return rc;
    c->has_held = true;
  }
  if (int rc = join_inflight(c)) return rc;
  batch = std::max(batch, c->batch);
  maxN = std::max(maxN, c->L.maxN);
  maxM = std::max(maxM, c->L.maxM);
  destroy_graph(c);
  if (c->d_base) HIPCHK(c, hipFree(c->d_base));
  if (c->h_stage) HIPCHK(c, hipHostFree(c->h_stage));
  if (c->h_down) HIPCHK(c, hipHostFree(c->h_down));
  c->d_base = nullptr, c->h_stage = nullptr, c->h_down = nullptr;
  c->stage_busy = false;  // (hipFree above waited for the device)
  // grow with headroom so a slowly growing window does not re-allocate every frame
  Layout L = make_layout(maxN + maxN / 4 + 64, maxM + maxM / 4 + 256);
  // one small window: a shadow slot behind it for the marginalization run ahead (kernels_spec.h)
  c->shadow = batch == 1 && L.maxN <= SHADOW_MAX_LM && c->sstream[0] && c->sstream[lfvio_ctx::WORKERS - 1] && c->d_mail;
  const size_t slots = (size_t)batch + (c->shadow ? lfvio_ctx::WORKERS : 0);
  HIPCHK(c, hipMalloc((void **)&c->d_base, L.total * slots));
  HIPCHK(c, hipMemsetAsync(c->d_base, 0, L.total * slots, c->stream));
  HIPCHK(c, hipHostMalloc((void **)&c->h_stage, L.linw_end, hipHostMallocDefault));
  c->h_down_bytes = down_bytes(L.maxN);
  HIPCHK(c, hipHostMalloc((void **)&c->h_down, c->h_down_bytes, hipHostMallocDefault));
  c->L = L;
  c->batch = batch;
  c->info.assign(batch, SlotHostInfo());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return LFVIO_OK;
}

inline int local_size(int kind) {
  return (kind == LFVIO_BLOCK_POSE || kind == LFVIO_BLOCK_EX_POSE) ? 6 : (kind == LFVIO_BLOCK_SPEEDBIAS ? 9 : 1);
}
inline int tangent_off(int kind, int frame) {
  return kind == LFVIO_BLOCK_POSE ? off_pose(frame)
         : kind == LFVIO_BLOCK_SPEEDBIAS ? off_sb(frame)
         : kind == LFVIO_BLOCK_EX_POSE ? off_ex()
                                       : off_td();
}

// Structure of MarginalizationInfo for this window (estimator.cpp:833-1005): which blocks
// take part, which are dropped, canonical column order (dropped first, then poses / speed-bias
// by frame, ex pose, td) and the addr_shift of the kept blocks.
// N0 / nChunks0: what THIS slot sweeps; any0 / kmax0: whether the WINDOW has landmarks anchored at frame 0 and their longest track —
// the same thing unless the slot holds a rank's share of a landmark-sharded window, whose block structure is the whole window's
void plan_marg(const LfvioWindow *w, const LfvioPrior *pr, int flag, int N0, bool any0, int kmax0, int nChunks0, bool imu0_ok, MargPlan *mp) {
  std::memset(mp, 0, sizeof *mp);
  for (int c = 0; c < KP; c++) mp->col[c] = -1;
  bool present[4][LFVIO_NUM_FRAMES] = {}, dropped[4][LFVIO_NUM_FRAMES] = {};
  auto touch = [&](int kind, int frame, bool drop) {
    present[kind][frame] = true;
    if (drop) dropped[kind][frame] = true;
  };
  if (flag == LFVIO_MARGIN_OLD) {
    if (pr)
      for (int i = 0; i < pr->num_blocks; i++) {
        const int k = pr->blocks[i].kind, f = pr->blocks[i].frame;
        touch(k, f, (k == LFVIO_BLOCK_POSE || k == LFVIO_BLOCK_SPEEDBIAS) && f == 0);
      }
    mp->use_imu0 = (w->imu[0].sum_dt < 10.0) && imu0_ok;
    if (w->imu[0].sum_dt < 10.0) {
      touch(LFVIO_BLOCK_POSE, 0, true), touch(LFVIO_BLOCK_SPEEDBIAS, 0, true);
      touch(LFVIO_BLOCK_POSE, 1, false), touch(LFVIO_BLOCK_SPEEDBIAS, 1, false);
    }
    if (any0) {
      touch(LFVIO_BLOCK_POSE, 0, true);
      for (int j = 1; j < kmax0; j++) touch(LFVIO_BLOCK_POSE, j, false);
      touch(LFVIO_BLOCK_EX_POSE, 0, false);
      if (w->estimate_td) touch(LFVIO_BLOCK_TD, 0, false);
    }
    mp->N0 = N0;
    mp->nChunks0 = nChunks0;
    mp->use_visual = N0 > 0;
    mp->valid = 1;
  } else {
    bool touches = false;
    if (pr)
      for (int i = 0; i < pr->num_blocks; i++)
        if (pr->blocks[i].kind == LFVIO_BLOCK_POSE && pr->blocks[i].frame == LFVIO_WINDOW_SIZE - 1) touches = true;
    if (!touches) return;  // valid = 0
    for (int i = 0; i < pr->num_blocks; i++) {
      const int k = pr->blocks[i].kind, f = pr->blocks[i].frame;
      touch(k, f, k == LFVIO_BLOCK_POSE && f == LFVIO_WINDOW_SIZE - 1);
    }
    mp->valid = 1;
  }
  int pos = 0;
  for (int k = 0; k < 4; k++)
    for (int f = 0; f < LFVIO_NUM_FRAMES; f++)
      if (present[k][f] && dropped[k][f]) {
        const int o = tangent_off(k, f);
        for (int e = 0; e < local_size(k); e++) mp->col[o + e] = pos++;
      }
  mp->m15 = pos;
  int nb = 0;
  for (int k = 0; k < 4; k++)
    for (int f = 0; f < LFVIO_NUM_FRAMES; f++)
      if (present[k][f] && !dropped[k][f]) {
        const int o = tangent_off(k, f);
        mp->kind[nb] = k, mp->frame[nb] = f, mp->idx[nb] = pos - mp->m15;
        int sf = f;
        if (k == LFVIO_BLOCK_POSE || k == LFVIO_BLOCK_SPEEDBIAS) {
          if (flag == LFVIO_MARGIN_OLD) sf = f - 1;
          else if (f == LFVIO_WINDOW_SIZE) sf = LFVIO_WINDOW_SIZE - 1;
        }
        mp->shifted_frame[nb] = sf;
        nb++;
        for (int e = 0; e < local_size(k); e++) mp->col[o + e] = pos++;
      }
  mp->nb = nb;
  mp->n = pos - mp->m15;
}

// a prior carries n x n of its 172 x 172 Jacobian slots: copy what is there (46 KB instead of 240 KB for n = 76)
void copy_prior(LfvioPrior *dst, const LfvioPrior *src) {
  std::memcpy(dst, src, offsetof(LfvioPrior, linearized_jacobians));
  if (src->valid && src->n > 0 && src->n <= LFVIO_MAX_PRIOR_DIM) {
    std::memcpy(dst->linearized_jacobians, src->linearized_jacobians, sizeof(double) * src->n * src->n);
    std::memcpy(dst->linearized_residuals, src->linearized_residuals, sizeof(double) * src->n);
  }
}

}  // namespace
extern "C" int lfvio_batch_optimize_finish(lfvio_ctx *c, LfvioPrior *prior);
namespace {

int check_input_prior(lfvio_ctx *c, const LfvioPrior *pr) {
  if (pr && (pr->n <= 0 || pr->n > LFVIO_MAX_PRIOR_DIM || pr->num_blocks <= 0 || pr->num_blocks > LFVIO_MAX_PRIOR_BLOCKS)) {
    c->err = "malformed prior";
    return LFVIO_ERR_ARG;
  }
  if (pr) {
    // the blocks index present[kind][frame], prior_cmap and prior_inv below: refuse anything that would leave them
    for (int i = 0; i < pr->num_blocks; i++) {
      const int k = pr->blocks[i].kind, f = pr->blocks[i].frame, idx = pr->block_idx[i];
      const bool framed = k == LFVIO_BLOCK_POSE || k == LFVIO_BLOCK_SPEEDBIAS;
      if (k < LFVIO_BLOCK_POSE || k > LFVIO_BLOCK_TD || f < 0 || f >= LFVIO_NUM_FRAMES || (!framed && f != 0) || idx < 0 ||
          idx + local_size(k) > pr->n) {
        c->err = "malformed prior block (kind / frame / block_idx out of range)";
        return LFVIO_ERR_ARG;
      }
    }
  }
  return LFVIO_OK;
}

constexpr size_t LIN_SPLIT_WGS = 2048;  // launches of k_lin with more workgroups than this are issued role by role ...
constexpr int LINB_MAX_GROUPS = 500;     // ... in at most this many groups (two workgroups per CU: 512 at once, one of them the pose side's)
// A single window from this many landmarks on is linearized group by group (k_linb, kernels_linw.h).  A group is a latency of ~60 us
// however few there are; the role-by-role sweep grows with the window (measured, whole optimization(), profiles/r04/linb_times.txt:
// 20 000 landmarks 1.29 ms role by role / 1.34 by groups, 50 000: 1.58 / 1.48, 100 000: 1.94 / 1.63, 200 000: 3.17 / 2.39).
constexpr int LINB_MIN_LM = 40960;
constexpr int LIN_SPLIT_MIN_BATCH = 8;   // ... when they are a batch: ONE large window (100 000 landmarks: 64 us as one grid, 54 + 35 + 14 role by role) is better off with its roles overlapping

// Pack one window into the pinned staging blob and upload it to slot `slot`.
// chain (lfvio_batch_upload_chained): the window's prior is *chain, and if a call is still in flight on the context (the
// marginalization behind an early state), it is THAT call's prior: the landmark tables and gather lists of the new window —
// nine tenths of the host work of an upload — are packed while the device finishes it, then the prior is taken from the
// mailbox into *chain and the upload goes on with it.
// device_chain (lfvio_batch_upload_chained_device): the prior of the call in flight STAYS on the device — nothing is waited for or
// collected; the new window's prior has the structure the host planned for that marginalization (SlotHostInfo::marg_out) and the
// values k_prior_chain copies out of Slot::prior_out behind these copies, which the stream puts behind the marginalization.
int upload_window(lfvio_ctx *c, int slot, const LfvioWindow *w, int sharded = 0, int pose_side = 1, LfvioPrior *chain = nullptr, bool device_chain = false) {
  if (device_chain) {
    if (slot != 0 || sharded || !c->inflight || c->has_held || !c->info[0].resident) {
      c->err = "lfvio_batch_upload_chained_device: no call in flight on slot 0 (lfvio_batch_optimize_begin) whose prior could be taken over";
      return LFVIO_ERR_ARG;
    }
    const SlotHostInfo::MargOut &mo = c->info[0].marg_out[c->inflight_flag];
    if (!mo.valid) {
      c->err = "lfvio_batch_upload_chained_device: the marginalization in flight passes its input prior through (MARGIN_SECOND_NEW without a "
               "prior on the newest pose) — use lfvio_batch_upload_chained";
      return LFVIO_ERR_ARG;
    }
    if (!c->chain_struct) c->chain_struct.reset(new LfvioPrior);
    LfvioPrior *sp = c->chain_struct.get();
    std::memset(sp, 0, offsetof(LfvioPrior, linearized_jacobians));
    sp->valid = 1, sp->n = mo.n, sp->m = mo.m, sp->num_blocks = mo.nb;
    if (c->debug_break_chain) sp->n = mo.n + 1, c->debug_break_chain = false;  // (k_prior_chain finds a prior of mo.n rows: the failure path, end to end)
    for (int i = 0; i < mo.nb; i++) sp->blocks[i].kind = mo.kind[i], sp->blocks[i].frame = mo.frame[i], sp->block_idx[i] = mo.idx[i];
    chain = sp;
  }
  const bool chained = !device_chain && chain && (c->inflight || c->has_held) && slot == 0;
  const auto t_up0 = std::chrono::steady_clock::now();
  auto lap = [&](int k, std::chrono::steady_clock::time_point from) {
    const auto now = std::chrono::steady_clock::now();
    c->up_us[k] = std::chrono::duration<double>(now - from).count() * 1e6;
    return now;
  };
  if (!chained && !device_chain)
    if (int rc = join_inflight(c)) return rc;
  if (!w || w->num_landmarks < 0 || w->num_observations < 0) {
    c->err = "null window / negative sizes";
    return LFVIO_ERR_ARG;
  }
  const int N = w->num_landmarks, M = w->num_observations;
  if (N > 0 && (!w->start_frame || !w->obs_offset || !w->inv_depth || !w->obs_point || !w->obs_velocity ||
                !w->obs_cur_td || !w->obs_uv_y)) {
    c->err = "null landmark / observation arrays";
    return LFVIO_ERR_ARG;
  }
  if (N > 0 && (w->obs_offset[0] != 0 || w->obs_offset[N] != M)) {
    c->err = "obs_offset is not a CSR over num_observations";
    return LFVIO_ERR_ARG;
  }
  for (int l = 0; l < N; l++) {
    const int k = w->obs_offset[l + 1] - w->obs_offset[l], s = w->start_frame[l];
    if (k < 2 || s < 0 || s + k > LFVIO_NUM_FRAMES) {  // used_num >= 2, track inside the window
      c->err = "landmark with fewer than 2 observations or a track leaving the window";
      return LFVIO_ERR_ARG;
    }
  }
  const LfvioPrior *given = chain ? chain : w->prior;
  const LfvioPrior *pr = (given && given->valid) ? given : nullptr;  // (chained: not known before the prior section below)
  if (!chained)
    if (int rc = check_input_prior(c, pr)) return rc;
  if (w->estimate_td && !(w->row > 0.0)) {  // row_i = uv.y - ROW / 2 and TR / ROW (projection_td_factor.cpp:20-21, 54-55)
    c->err = "estimate_td needs row > 0";
    return LFVIO_ERR_ARG;
  }
  const Layout &L = c->L;
  char *h = c->h_stage;
  char *d = c->d_base + (size_t)slot * L.total;
  Slot *S = (Slot *)h;
  if (c->stage_busy) {  // the previous upload's copies out of the staging block (long done, unless uploads come back to back)
    c->stage_busy = false;
    HIPCHK(c, hipEventSynchronize(c->stage_event));
  }
  std::memset(S, 0, offsetof(Slot, x));
  S->N = N, S->M = M, S->NV = M - N;
  S->est_ex = w->estimate_extrinsic != 0, S->est_td = w->estimate_td != 0;
  S->max_iter = w->max_num_iterations;
  S->sharded = sharded, S->pose_side = pose_side;
  S->mail = (slot == 0 && !sharded && c->d_mail && w->num_landmarks <= MAIL_MAX_LM) ? (long long)(uintptr_t)c->d_mail : 0;
  // (the loop's side of kernels_spec.h costs a store per pass and a compare-and-swap at its end; whether workers are started is
  // decided per call, enqueue_solve)
  S->spec_on = (slot == 0 && c->shadow && c->marg_ahead && !sharded && S->mail && N <= SPEC_MAX_LM) ? 1 : 0;
  const int seq = c->mail_seq == 0x7fffffff ? 1 : c->mail_seq + 1;  // (never 0: the host clears the flags to 0)
  S->mail_seq = seq;
  for (int k = 0; k < 3; k++) S->g[k] = w->g[k];
  S->tr_over_row = w->row > 0.0 ? w->tr / w->row : 0.0;  // only the td factor reads it (row > 0 checked above)
  S->half_row = w->row / 2;
  S->sqrt_info = w->sqrt_info;
  S->fn_tol = c->fn_tol;
  S->init_radius = c->init_radius;
  std::memcpy(S->x0.pose, w->para_pose, sizeof S->x0.pose);
  std::memcpy(S->x0.sb, w->para_speed_bias, sizeof S->x0.sb);
  std::memcpy(S->x0.ex, w->para_ex_pose, sizeof S->x0.ex);
  S->x0.td = w->para_td;
  for (int i = 0; i < LFVIO_WINDOW_SIZE; i++) {
    S->imu[i] = w->imu[i];
    S->imu_active[i] = !(w->imu[i].sum_dt > 10.0);  // estimator.cpp:720
  }
  // ---- landmarks: stable bucket sort by (start_frame, track length)
  // What the host keeps about the slot (sizes, permutation, grid) is built beside it and committed where the copies are
  // enqueued: an upload refused on the way — or a chained one, which collects the prior of the window still resident in
  // between — leaves the slot's description matching what the device holds.
  SlotHostInfo &info = c->info[slot];
  std::vector<int> &perm = c->perm_build;
  perm.resize(N);
  {
    int count[16 * 16 + 1] = {0};
    for (int l = 0; l < N; l++) count[w->start_frame[l] * 16 + (w->obs_offset[l + 1] - w->obs_offset[l]) + 1]++;
    for (int k = 0; k < 256; k++) count[k + 1] += count[k];
    for (int l = 0; l < N; l++) perm[count[w->start_frame[l] * 16 + (w->obs_offset[l + 1] - w->obs_offset[l])]++] = l;
  }
  int *lm_start = (int *)(h + L.lm_start), *lm_cnt = (int *)(h + L.lm_cnt), *lm_obs0 = (int *)(h + L.lm_obs0);
  int *lm_perm = (int *)(h + L.lm_perm), *lm_woff = (int *)(h + L.lm_woff);
  double *lam0 = (double *)(h + L.lam0);
  double *obs[8];
  for (int k = 0; k < 8; k++) obs[k] = (double *)(h + L.obs[k]);
  int o = 0, N0 = 0, kmax0 = 0;
  int pair_count[NPAIR + 1] = {0};
  for (int dl = 0; dl < N; dl++) {
    const int l = perm[dl];
    const int s = w->start_frame[l], k = w->obs_offset[l + 1] - w->obs_offset[l], o0 = w->obs_offset[l];
    lm_start[dl] = s, lm_cnt[dl] = k, lm_obs0[dl] = o, lm_perm[dl] = l;
    lm_woff[dl] = dl == 0 ? 0 : lm_woff[dl - 1] + w_row_len(lm_cnt[dl - 1]);
    lam0[dl] = w->inv_depth[l];
    if (s == 0) N0++, kmax0 = std::max(kmax0, k);
    for (int q = 0; q < k; q++, o++) {
      const int src = o0 + q;
      obs[0][o] = w->obs_point[3 * src], obs[1][o] = w->obs_point[3 * src + 1], obs[2][o] = w->obs_point[3 * src + 2];
      obs[3][o] = w->obs_velocity[3 * src], obs[4][o] = w->obs_velocity[3 * src + 1], obs[5][o] = w->obs_velocity[3 * src + 2];
      obs[6][o] = w->obs_cur_td[src];
      obs[7][o] = w->obs_uv_y[src];
      if (q > 0) pair_count[s * 11 + s + q + 1]++;
    }
  }
  // ---- pair-major list of the non-anchor observations + chunk table
  for (int p = 0; p < NPAIR; p++) pair_count[p + 1] += pair_count[p];
  int *pm_obs = (int *)(h + L.pm_obs), *pm_lm = (int *)(h + L.pm_lm);
  {
    int cursor[NPAIR];
    for (int p = 0; p < NPAIR; p++) cursor[p] = pair_count[p];
    for (int dl = 0; dl < N; dl++) {
      const int s = lm_start[dl], k = lm_cnt[dl];
      for (int q = 1; q < k; q++) {
        const int p = s * 11 + s + q;
        pm_obs[cursor[p]] = lm_obs0[dl] + q;
        pm_lm[cursor[p]] = dl;
        cursor[p]++;
      }
    }
  }
  int *chunk_pair = (int *)(h + L.chunk_pair), *chunk_begin = (int *)(h + L.chunk_begin), *chunk_end = (int *)(h + L.chunk_end);
  int nChunks = 0;
  for (int p = 0; p < NPAIR; p++) {
    S->pair_chunk0[p] = nChunks;
    for (int b = pair_count[p]; b < pair_count[p + 1]; b += CHUNK_MAX) {
      if (nChunks >= L.capChunks) {
        c->err = "chunk table overflow";
        return LFVIO_ERR_ARG;
      }
      chunk_pair[nChunks] = p, chunk_begin[nChunks] = b, chunk_end[nChunks] = std::min(b + CHUNK_MAX, pair_count[p + 1]);
      nChunks++;
    }
  }
  S->pair_chunk0[NPAIR] = nChunks;
  S->nChunks = nChunks;
  lm_woff[N] = N == 0 ? 0 : lm_woff[N - 1] + w_row_len(lm_cnt[N - 1]);
  S->nLmBlocks = (N + LM_BLOCK - 1) / LM_BLOCK;
  // one part per landmark workgroup of k_lin, which forms it from its LDS tile: 64 landmarks, or 32 in the 8-lanes-per-track form
  S->lm_half = (N <= SPEC_MAX_LM && c->lm_half) ? 1 : 0;
  S->schur_lm = S->lm_half ? SCHUR_LM / 2 : SCHUR_LM;
  S->nSchurParts = S->lm_half ? (N + SCHUR_LM / 2 - 1) / (SCHUR_LM / 2) : S->nLmBlocks;
  // ---- k_linw (kernels_linw.h): strips of <= 64 landmarks of one start frame, dealt to the four waves of the window's
  //      workgroup (all strips of a start frame on one wave: it is the one writer of that frame's pair blocks), and the
  //      observations once more in the orders its lanes read them — anchors by landmark, the others pair-major.
  bool linw = c->linw_mode != 0 && !sharded && N <= SPEC_MAX_LM && (c->batch >= LIN_SPLIT_MIN_BATCH || c->linw_mode == 2);
  // a large single window: the same strips as groups of four of one start frame, one workgroup each (k_linb)
  bool linb = c->linw_mode != 0 && !linw && N >= (c->linw_mode == 2 ? SPEC_MAX_LM + 1 : LINB_MIN_LM);  // (a rank's share of a sharded window too)
  int linb_ng = 0;
  if (linw || linb) {
    LinwPlan &P = S->linw;
    int begin_s[LFVIO_NUM_FRAMES + 1];
    {
      int dl = 0;
      for (int s = 0; s <= LFVIO_NUM_FRAMES; s++) {
        while (dl < N && lm_start[dl] < s) dl++;
        begin_s[s] = dl;
      }
    }
    struct Strip {
      int lm0, nlm, start, kmax;
    };
    std::vector<Strip> by_start[LFVIO_NUM_FRAMES];
    int cost[LFVIO_NUM_FRAMES] = {0}, n_strips = 0;
    for (int s = 0; s < LFVIO_NUM_FRAMES; s++) {
      {
        int f = begin_s[s];  // (ascending track length inside a start frame)
        for (int o2 = 0; o2 < 12; o2++) {
          while (f < begin_s[s + 1] && lm_cnt[f] <= o2) f++;
          P.firstl[s][o2] = f;
        }
      }
      for (int l0 = begin_s[s]; l0 < begin_s[s + 1] && !linb; l0 += 64) {
        const int n = std::min(64, begin_s[s + 1] - l0);
        by_start[s].push_back(Strip{l0, n, s, lm_cnt[l0 + n - 1]});  // (ascending track length inside a start frame: the last one is the longest)
        cost[s] += lm_cnt[l0 + n - 1] - 1;
        n_strips++;
      }
    }
    auto observation_copies = [&] {
      for (int p = 0; p <= NPAIR; p++) P.pair_obs0[p] = pair_count[p];
      double *anc[8], *pmo[8];
      for (int k = 0; k < 8; k++) anc[k] = (double *)(h + L.anc[k]), pmo[k] = (double *)(h + L.pmo[k]);
      for (int dl = 0; dl < N; dl++)
        for (int k = 0; k < 8; k++) anc[k][dl] = obs[k][lm_obs0[dl]];
      for (int q = 0; q < M - N; q++)
        for (int k = 0; k < 8; k++) pmo[k][q] = obs[k][pm_obs[q]];
      unsigned char *pm_pair = (unsigned char *)(h + L.pm_pair);
      for (int p = 0; p < NPAIR; p++)
        for (int q = pair_count[p]; q < pair_count[p + 1]; q++) pm_pair[q] = (unsigned char)p;
    };
    if (linb) {
      // the groups of k_linb: consecutive strips of one start frame sized by a cost model, the most expensive first (linb_plan.h)
      const std::vector<LinbGroup> groups = linb_plan_groups(begin_s, LFVIO_NUM_FRAMES, lm_cnt, LM_BLOCK, LINB_MAX_GROUPS);
      int *g_lm0 = (int *)(h + L.linb_lm0), *g_ns = (int *)(h + L.linb_ns);
      linb_ng = (int)groups.size();
      if (linb_ng > L.capSchurParts) linb = false, linb_ng = 0;  // (the partials live in the Schur partials' space)
      else {
        for (int g2 = 0; g2 < linb_ng; g2++) g_lm0[g2] = groups[g2].lm0, g_ns[g2] = groups[g2].n | (groups[g2].s << 16);
        P.big = 1, P.ng = linb_ng, P.wt_ld = L.capLmBlocks * LM_BLOCK;
        for (int p = 0; p <= NPAIR; p++) P.pair_obs0[p] = pair_count[p];  // (the observations' copies are made on the device: k_linb_gather)
      }
    } else if (n_strips > LINW_MAX_STRIPS) linw = false;
    else {
      // longest-processing-time first over the start frames
      int order[LFVIO_NUM_FRAMES], load[LINW_WAVES] = {0};
      for (int s = 0; s < LFVIO_NUM_FRAMES; s++) order[s] = s;
      std::stable_sort(order, order + LFVIO_NUM_FRAMES, [&](int a, int b) { return cost[a] > cost[b]; });
      std::vector<Strip> per_wave[LINW_WAVES];
      for (int k = 0; k < LFVIO_NUM_FRAMES; k++) {
        const int s = order[k];
        if (by_start[s].empty()) continue;
        int w = 0;
        for (int q = 1; q < LINW_WAVES; q++)
          if (load[q] < load[w]) w = q;
        load[w] += cost[s];
        per_wave[w].insert(per_wave[w].end(), by_start[s].begin(), by_start[s].end());
      }
      int t = 0;
      for (int w = 0; w < LINW_WAVES; w++) {
        P.wave_first[w] = t;
        for (const Strip &st : per_wave[w]) P.lm0[t] = (short)st.lm0, P.nlm[t] = (short)st.nlm, P.start[t] = (short)st.start, P.kmax[t] = (short)st.kmax, t++;
      }
      P.wave_first[LINW_WAVES] = t;
      P.n_strips = t;
      observation_copies();
    }
    P.ok = (linw || linb) ? 1 : 0;
  }
  int used_items = 0;
  bool lists_cached = false;
  // ---- gather lists of k_sum: which Gram entries (chunk or, for large windows, frame pair; local index of the
  //      20 x 20 block [Pi th_i Pj th_j tic th_ic td | r]) add up to each packed H_pp / g_p entry.  Units ascend, so the
  //      marginalization's subset (pairs (0, j)) is a prefix of every list.
  {
    S->pre_gram = nChunks > PRE_CHUNK_LIMIT ? 1 : 0;
    auto col = [](int l, int i, int j) { return l < 6 ? 6 * i + l : l < 12 ? 6 * j + (l - 6) : l < 18 ? 66 + (l - 12) : 72; };
    const int units = S->pre_gram ? NPAIR : nChunks;
    int *sum_off = (int *)(h + L.sum_off), *sum_end_marg = (int *)(h + L.sum_end_marg), *sum_items = (int *)(h + L.sum_items);
    const int marg_units = S->pre_gram ? 11 : S->pair_chunk0[11];  // pairs (0, j) / their chunks
    // Which (frame pair, local entry) pairs feed an H_pp / g_p entry is the same for every window: a table built once.
    // Only the visual entries have a list (rows < KC: the packed prefix [0, KC (KC + 1) / 2) and g_p[0, KC)); k_sum reads
    // the bounds of the others but never uses them.  Per upload the lists are ONE pass over those 2 774 entries — the
    // pairs of the entry in ascending order, each with its chunks in ascending order, so units ascend as before.
    struct PairLocal {
      short pair, local;
    };
    struct EntryTable {
      std::vector<int> first;       // [VIS + 1] into items
      std::vector<PairLocal> items;
    };
    constexpr int VIS_PACKED = SUM_VIS_PACKED, VIS = SUM_VIS;  // visual entries: packed prefix, then g_p
    static_assert(SUM_VIS_PACKED == KC * (KC + 1) / 2 && SUM_VIS == SUM_VIS_PACKED + KC, "compact gather index");
    static const EntryTable table = [&] {
      EntryTable t;
      std::vector<std::vector<PairLocal>> per(VIS);
      for (int pp = 0; pp < NPAIR; pp++) {
        const int i = pp / 11, j = pp % 11;
        if (i >= j) continue;
        for (int lp = 0; lp < 19; lp++) {
          const int cp = col(lp, i, j);
          for (int lq = lp; lq < 20; lq++) {
            const int e = lq == 19 ? VIS_PACKED + cp : col(lq, i, j) * (col(lq, i, j) + 1) / 2 + cp;
            per[e].push_back(PairLocal{(short)pp, (short)(lp * 20 - (lp * (lp - 1)) / 2 + (lq - lp))});
          }
        }
      }
      t.first.assign(VIS + 1, 0);
      for (int e = 0; e < VIS; e++) {
        t.first[e + 1] = t.first[e] + (int)per[e].size();
        t.items.insert(t.items.end(), per[e].begin(), per[e].end());
      }
      return t;
    }();
    int n_items = 0;
    bool overflow = false;
    {
      // the lists depend on nothing but this table: the ones the device holds from the last upload of the slot still stand
      // if it has not changed (the pass below over ~11 500 items is the largest single part of an upload)
      std::vector<int> key(S->pair_chunk0, S->pair_chunk0 + NPAIR + 1);
      key.push_back(S->pre_gram);
      lists_cached = info.uploaded && info.list_items >= 0 && key == info.list_key;
      if (!lists_cached) info.list_key.swap(key), info.list_items = -1;
    }
    if (lists_cached) n_items = info.list_items;
    for (int v = 0; v < VIS && !overflow && !lists_cached; v++) {
      const int e = v;  // the bounds are stored by the compact index
      sum_off[e] = n_items;
      int marg_end = n_items;
      for (int k = table.first[v]; k < table.first[v + 1]; k++) {
        const int pp = table.items[k].pair, local = table.items[k].local;
        const int c0 = S->pair_chunk0[pp], c1 = S->pair_chunk0[pp + 1];
        if (c1 == c0) continue;
        if (n_items + (c1 - c0) > SUM_ITEMS_CAP) {
          overflow = true;
          break;
        }
        if (S->pre_gram) {
          sum_items[n_items++] = pp * NGP + local;
        } else {
          for (int ch = c0; ch < c1; ch++) sum_items[n_items++] = ch * NGP + local;
        }
        if (pp < 11) marg_end = n_items;  // pairs (0, j): the marginalization's subset, a prefix of the list
      }
      sum_end_marg[e] = marg_end;
    }
    if (!lists_cached) sum_off[VIS] = n_items;
    used_items = n_items;
    if (overflow) {
      c->err = "gather list overflow";
      return LFVIO_ERR_ARG;
    }
    (void)units;
    (void)marg_units;
  }
  // ---- prior
  auto t_up1 = lap(0, t_up0);
  c->up_us[1] = 0.0;
  if (chained) {
    // everything above was host work on the staging block; the prior of the call in flight is needed from here on
    // (c->info[0] still describes THAT window's prior: marg_n, the input prior a pass-through hands back)
    if (int rc = lfvio_batch_optimize_finish(c, chain)) return rc;
    pr = chain->valid ? chain : nullptr;
    if (int rc = check_input_prior(c, pr)) return rc;
    t_up1 = lap(1, t_up1);
  }
  for (int c2 = 0; c2 < KP; c2++) S->prior_inv[c2] = -1;
  if (pr) {
    S->prior_valid = 1, S->prior_n = pr->n, S->prior_nb = pr->num_blocks;
    for (int i = 0; i < pr->num_blocks; i++) {
      S->prior_kind[i] = pr->blocks[i].kind, S->prior_frame[i] = pr->blocks[i].frame, S->prior_idx[i] = pr->block_idx[i];
      if (!device_chain) std::memcpy(S->prior_x0[i], pr->block_x0[i], sizeof(double) * 9);  // (device_chain: k_prior_chain fills them)
      const int to = tangent_off(pr->blocks[i].kind, pr->blocks[i].frame);
      for (int e = 0; e < local_size(pr->blocks[i].kind); e++) {
        S->prior_cmap[pr->block_idx[i] + e] = to + e;
        S->prior_inv[to + e] = pr->block_idx[i] + e;
      }
    }
    if (!device_chain) {
      std::memcpy(h + L.prior_J, pr->linearized_jacobians, sizeof(double) * pr->n * pr->n);
      std::memcpy(h + L.prior_r, pr->linearized_residuals, sizeof(double) * pr->n);
    }
  }
  {
    // (a rank's share of a sharded window: the other ranks' frame-0 landmarks shape the prior's blocks too — lfvio_shard_begin)
    const bool whole = sharded && c->shard_kmax0 >= 0;
    plan_marg(w, pr, LFVIO_MARGIN_OLD, N0, whole ? c->shard_kmax0 > 0 : N0 > 0, whole ? c->shard_kmax0 : kmax0, S->pair_chunk0[11], true, &S->marg[0]);
  }
  plan_marg(w, pr, LFVIO_MARGIN_SECOND_NEW, 0, false, 0, 0, true, &S->marg[1]);
  for (int f = 0; f < 2; f++)
    if (S->marg[f].valid && (S->marg[f].n > 76 || S->marg[f].m15 + S->marg[f].n > 92)) {
      // k_marg_solve keeps the dense system in LDS: sized for what the reference's own marginalization produces
      // (10 poses + ex + td + one speed/bias = 76 kept, 15 dropped), not for an arbitrary hand-made prior
      c->err = "prior keeps more than 76 tangent dimensions";
      return LFVIO_ERR_ARG;
    }
  // ---- device pointers
  S->lm_start.set(S, L.lm_start), S->lm_cnt.set(S, L.lm_cnt), S->lm_obs0.set(S, L.lm_obs0), S->lm_perm.set(S, L.lm_perm);
  S->lm_woff.set(S, L.lm_woff);
  S->lam0.set(S, L.lam0);
  for (int k = 0; k < 8; k++) S->obs[k].set(S, L.obs[k]);
  S->pm_obs.set(S, L.pm_obs), S->pm_lm.set(S, L.pm_lm);
  for (int k = 0; k < 8; k++) S->anc[k].set(S, L.anc[k]), S->pmo[k].set(S, L.pmo[k]);
  S->pm_pair.set(S, L.pm_pair);
  S->linb_lm0.set(S, L.linb_lm0), S->linb_ns.set(S, L.linb_ns);
  S->chunk_pair.set(S, L.chunk_pair), S->chunk_begin.set(S, L.chunk_begin), S->chunk_end.set(S, L.chunk_end);
  S->prior_J.set(S, L.prior_J), S->prior_r.set(S, L.prior_r);
  S->sum_off.set(S, L.sum_off), S->sum_end_marg.set(S, L.sum_end_marg), S->sum_items.set(S, L.sum_items);
  // ---- commit: from here on the device copy changes, and the slot's description with it
  info.resident = false;
  info.N = N, info.M = M;
  info.prior_n = S->prior_valid ? S->prior_n : 0;
  info.max_iter = w->max_num_iterations, info.max_seconds = w->max_solver_time_in_seconds;
  info.perm.swap(perm);
  info.gLm = S->nLmBlocks, info.gLw = S->nSchurParts, info.gCh = nChunks, info.gSc = S->nSchurParts;
  info.has_in_prior = pr != nullptr;
  info.in_prior_device = pr && device_chain;
  if (pr && device_chain) std::memcpy(&info.in_prior, pr, offsetof(LfvioPrior, linearized_jacobians));  // (structure; the values are the device's)
  else if (pr) copy_prior(&info.in_prior, pr);
  for (int f = 0; f < 2; f++) {
    SlotHostInfo::MargOut &mo = info.marg_out[f];
    const MargPlan &mp = S->marg[f];
    mo.valid = mp.valid, mo.n = mp.n, mo.m = mp.m15 + mp.N0, mo.nb = mp.nb;
    for (int i = 0; i < mp.nb; i++) mo.kind[i] = mp.kind[i], mo.frame[i] = mp.shifted_frame[i], mo.idx[i] = mp.idx[i];
  }
  info.mail_seq = c->mail_seq = seq;
  info.spec_on = S->spec_on != 0;
  {
    // (a little below the device's own threshold: a "yes" too many costs a few percent of one sweep, a "no" too many would put a
    // table with the residual chain's flavour in front of a kernel that does not look for it)
    auto off = [](const double *q) { return std::fabs((q[3] * q[3] + q[0] * q[0] + q[1] * q[1] + q[2] * q[2]) - 1.0) > 0.5 * TAB_OFF_SPHERE; };
    bool any = false;
    for (int f = 0; f < LFVIO_NUM_FRAMES; f++) any = any || off(w->para_pose[f] + 3);
    const bool ex = off(w->para_ex_pose + 3);
    info.offs_first = any || ex, info.offs_all = ex && !w->estimate_extrinsic;
  }
  info.marg_n = std::max(S->marg[0].valid ? S->marg[0].n : 0, S->marg[1].valid ? S->marg[1].n : 0);
  info.linw_ok = linw;
  info.linb_ok = linb, info.linb_ng = linb_ng;
  // (an upload behind a call in flight whose prior a worker on the second stream may own: it reads this slot's inputs until it delivers)
  if (slot == 0 && c->side_launched) hipLaunchKernelGGL(k_spec_wait, dim3(1), dim3(64), 0, c->stream, d);
  // header prefix + input arrays (two copies: the work-pointer part of the header is written once below)
  HIPCHK(c, hipMemcpyAsync(d, h, offsetof(Slot, x), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d + L.in_begin, h + L.in_begin, (lists_cached ? L.sum_off : L.sum_items + (size_t)used_items * 4) - L.in_begin, hipMemcpyHostToDevice,
                           c->stream));
  if (pr && !device_chain) HIPCHK(c, hipMemcpyAsync(d + L.prior_J, h + L.prior_J, sizeof(double) * pr->n * pr->n, hipMemcpyHostToDevice, c->stream));
  if (device_chain) {
    hipLaunchKernelGGL(k_prior_chain, dim3(1, 1), dim3(256), 0, c->stream, d, L.total);  // (the slot's own blob as base)
    // the call that was in flight is now only work on the stream in front of this window's: nothing of it is left to collect
    c->inflight = false, c->pipelined = true;
  }
  if (linw) HIPCHK(c, hipMemcpyAsync(d + L.linw_begin, h + L.linw_begin, L.pm_pair + (size_t)std::max(M - N, 0) - L.linw_begin, hipMemcpyHostToDevice, c->stream));
  if (linb) HIPCHK(c, hipMemcpyAsync(d + L.linb_lm0, h + L.linb_lm0, L.linw_end - L.linb_lm0, hipMemcpyHostToDevice, c->stream));
  info.list_items = used_items;  // (only now: an upload refused half-way leaves the key without lists on the device)
  if (!info.uploaded) {
    // work-array pointers: fixed per slot until the next reserve()
    Slot W;
    std::memset(&W, 0, sizeof W);
    W.lam[0].set(&W, L.lam[0]), W.lam[1].set(&W, L.lam[1]);
    for (int k = 0; k < SPEC_EXTRA; k++) W.lamE[k].set(&W, L.lamE[k]);
    W.cost_partE.set(&W, L.cost_partE);
    W.prior_A.set(&W, L.prior_A);
    W.a.set(&W, L.a), W.b.set(&W, L.b), W.W.set(&W, L.W), W.Wt.set(&W, L.Wt);
    W.scale_l.set(&W, L.scale_l), W.grad_l.set(&W, L.grad_l), W.gn_l.set(&W, L.gn_l);
    W.diag_l.set(&W, L.diag_l), W.einv_l.set(&W, L.einv_l), W.d1.set(&W, L.d1), W.d2.set(&W, L.d2);
    W.gram_part.set(&W, L.gram_part), W.pairG.set(&W, L.pairG);
    W.schur_part.set(&W, L.schur_part);
    W.xch.set(&W, L.xch);
    W.schur_sum.set(&W, L.xch + (size_t)XOFF_S * 8), W.gp.set(&W, L.xch + (size_t)XOFF_G * 8), W.Hpp.set(&W, L.xch + (size_t)XOFF_H * 8);
    W.lm_part.set(&W, L.lm_part), W.cost_part.set(&W, L.cost_part), W.imu_out.set(&W, L.imu_out), W.imu_raw.set(&W, L.imu_raw);
    W.mscr.set(&W, L.mscr);
    W.eig_aux.set(&W, L.eig_aux);
    // field-by-field so that only pointer members are touched
#define PUTP(field) HIPCHK(c, hipMemcpyAsync(d + offsetof(Slot, field), &W.field, sizeof W.field, hipMemcpyHostToDevice, c->stream))
    PUTP(lam); PUTP(lamE); PUTP(cost_partE);
    PUTP(prior_A);
    PUTP(a); PUTP(b); PUTP(W); PUTP(Wt); PUTP(scale_l); PUTP(grad_l); PUTP(gn_l); PUTP(diag_l); PUTP(einv_l); PUTP(d1); PUTP(d2);
    PUTP(gram_part); PUTP(pairG); PUTP(schur_part); PUTP(schur_sum); PUTP(xch); PUTP(gp); PUTP(lm_part); PUTP(cost_part); PUTP(imu_out); PUTP(imu_raw);
    PUTP(Hpp);
    PUTP(mscr); PUTP(eig_aux);
#undef PUTP
    if (c->shadow && slot == 0) {
      // a shadow slot's work arrays are its own, laid out like every slot's (the members are self-relative: the same bytes) —
      // but J0^T J0 of the input prior, which k_setup forms once per call, is read from slot 0
      GP<double> prior_A[lfvio_ctx::WORKERS];
      for (int wk = 0; wk < lfvio_ctx::WORKERS; wk++) {
        char *ds = c->d_base + (size_t)(c->batch + wk) * L.total;
#define PUTS(field) HIPCHK(c, hipMemcpyAsync(ds + offsetof(Slot, field), &W.field, sizeof W.field, hipMemcpyHostToDevice, c->stream))
        PUTS(lam); PUTS(lamE); PUTS(cost_partE);
        PUTS(a); PUTS(b); PUTS(W); PUTS(Wt); PUTS(scale_l); PUTS(grad_l); PUTS(gn_l); PUTS(diag_l); PUTS(einv_l); PUTS(d1); PUTS(d2);
        PUTS(gram_part); PUTS(pairG); PUTS(schur_part); PUTS(schur_sum); PUTS(xch); PUTS(gp); PUTS(lm_part); PUTS(cost_part); PUTS(imu_out); PUTS(imu_raw);
        PUTS(Hpp);
        PUTS(mscr); PUTS(eig_aux);
#undef PUTS
        prior_A[wk].off = W.prior_A.off - (long long)((size_t)(c->batch + wk) * L.total);
        HIPCHK(c, hipMemcpyAsync(ds + offsetof(Slot, prior_A), &prior_A[wk], sizeof prior_A[wk], hipMemcpyHostToDevice, c->stream));
      }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));  // W is on the stack
    info.uploaded = true;
  }
  if (linb) hipLaunchKernelGGL(k_linb_gather, dim3((std::max(N, M - N) + 255) / 256, 1), dim3(256), 0, c->stream, d, L.total);  // (the slot's own blob as base)
  // The staging block is reused by the next upload: that one waits for these copies (an event), this one does not — what
  // follows an upload is an optimization on the same stream, and a round trip to the device saved here is ~30 us of the call.
  const auto t_up2 = lap(2, t_up1);
  if (!c->stage_event) HIPCHK(c, hipEventCreateWithFlags(&c->stage_event, hipEventDisableTiming));
  HIPCHK(c, hipEventRecord(c->stage_event, c->stream));
  c->stage_busy = true;
  lap(3, t_up2);
  info.resident = true;
  return LFVIO_OK;
}

// A HIPCHK that returns in the middle of a stream capture would leave the stream capturing (every later call on it fails):
// the guard ends the capture and drops the partial graph on any early return.
struct CaptureGuard {
  hipStream_t stream;
  bool active = true;
  explicit CaptureGuard(hipStream_t s) : stream(s) {}
  hipError_t end(hipGraph_t *g) {
    active = false;
    return hipStreamEndCapture(stream, g);
  }
  ~CaptureGuard() {
    if (!active) return;
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(stream, &g);
    if (g) (void)hipGraphDestroy(g);
  }
};

struct Grid {
  int lm, ch, sc, lw;
  int ch_raw;  // largest chunk count of a slot before the rounding of `ch` (what decides the two-level sums: upload_window's pre_gram)
};
constexpr int CH_BUCKET = 16;

Grid grid_for(lfvio_ctx *c, int count) {
  Grid g{1, 1, 1, 1, 1};
  for (int s = 0; s < count; s++) {
    g.ch_raw = std::max(g.ch_raw, c->info[s].gCh);
    g.lm = std::max(g.lm, c->info[s].gLm);
    g.lw = std::max(g.lw, c->info[s].gLw);
    g.ch = std::max(g.ch, c->info[s].gCh);
    g.sc = std::max(g.sc, c->info[s].gSc);
  }
  // The launch dimensions are the key of the captured graphs, and in a stream of windows from one estimator the number of
  // frame-pair chunks changes with almost every frame: round it up to whole groups of four workgroups (a workgroup of
  // k_lin takes four chunks) so that neighbouring windows share a graph.  Every kernel tests its block index against
  // the slot's own counts (the grid of a resident batch is the maximum over its slots anyway), so spare blocks return.
  g.ch = (g.ch + CH_BUCKET - 1) / CH_BUCKET * CH_BUCKET;
  return g;
}

// a resident batch: k_lin role by role (and without the transposed copy of W: its k_dogleg reads the compact rows)
bool lin_split(int count, const Grid &g) {
  return count >= LIN_SPLIT_MIN_BATCH && (size_t)count * (g.lw + (g.ch + 3) / 4 + LFVIO_WINDOW_SIZE + 1) > LIN_SPLIT_WGS;
}

// offs: the visual roles in the instantiation that knows the off-sphere flavour (k_lin, kernels_lin.h); the default is the one that is
// always right, the loop's launches say what they need (slots_offs)
void launch_lin(lfvio_ctx *c, int count, const Grid &g, int mode, bool offs = true) {
  const int gram_wgs = (g.ch + 3) / 4;  // one chunk per wave
  const size_t st = c->L.total;
  if (lin_split(count, g)) {
    // A resident batch: the roles go out as separate launches, each of a kernel compiled for that role alone.  Measured at 512 windows of 300 landmarks:
    // landmark role 115 us + Gram role 175 us + IMU / prior roles 104 us on their own, 679 us as ONE grid — workgroups of four
    // different code paths side by side on every CU (the sweep is ~30 KB of straight-line code per role) do not share an
    // instruction cache well; two more launches cost 9 us.
    if (offs) {
      hipLaunchKernelGGL((k_lin<LIN_ROLE_LM, true>), dim3(g.lw, count), dim3(LIN_THREADS), 0, c->stream, c->d_base, st, mode, g.lw, 0);
      hipLaunchKernelGGL((k_lin<LIN_ROLE_GRAM, true>), dim3(gram_wgs, count), dim3(LIN_THREADS), 0, c->stream, c->d_base, st, mode | MODE_NOCOUNT, 0, gram_wgs);
    } else {
      hipLaunchKernelGGL((k_lin<LIN_ROLE_LM, false>), dim3(g.lw, count), dim3(LIN_THREADS), 0, c->stream, c->d_base, st, mode, g.lw, 0);
      hipLaunchKernelGGL((k_lin<LIN_ROLE_GRAM, false>), dim3(gram_wgs, count), dim3(LIN_THREADS), 0, c->stream, c->d_base, st, mode | MODE_NOCOUNT, 0, gram_wgs);
    }
    const bool raw = (mode & (MODE_GATED - 1)) == MODE_SOLVE && !(mode & (MODE_GATED | MODE_DECIDE));
    if (raw) hipLaunchKernelGGL(k_imu_raw, dim3((count * LFVIO_WINDOW_SIZE + 63) / 64), dim3(64), 0, c->stream, c->d_base, st, count);
    if (raw) hipLaunchKernelGGL((k_lin<LIN_ROLE_POSE_RAW, false>), dim3(LFVIO_WINDOW_SIZE + 1, count), dim3(LIN_THREADS), 0, c->stream, c->d_base, st, mode | MODE_NOCOUNT, 0, 0);
    else hipLaunchKernelGGL((k_lin<LIN_ROLE_POSE, false>), dim3(LFVIO_WINDOW_SIZE + 1, count), dim3(LIN_THREADS), 0, c->stream, c->d_base, st, mode | MODE_NOCOUNT, 0, 0);
    return;
  }
  if (offs) hipLaunchKernelGGL((k_lin<LIN_ROLE_ALL, true>), dim3(g.lw + gram_wgs + LFVIO_WINDOW_SIZE + 1, count), dim3(LIN_THREADS), 0, c->stream, c->d_base, st, mode, g.lw, gram_wgs);
  else hipLaunchKernelGGL((k_lin<LIN_ROLE_ALL, false>), dim3(g.lw + gram_wgs + LFVIO_WINDOW_SIZE + 1, count), dim3(LIN_THREADS), 0, c->stream, c->d_base, st, mode, g.lw, gram_wgs);
}
// what the sweeps of slots [0, count) need: bit 0 the sweep at the uploaded state, bit 1 every sweep
int slots_offs(const lfvio_ctx *c, int count) {
  int r = 0;
  for (int k = 0; k < count; k++) r |= (c->info[k].offs_first ? 1 : 0) | (c->info[k].offs_all ? 2 : 0);
  return r;
}

// fixed-order reduction of the partials; two levels once a single k_sum thread would have to walk hundreds of them
void launch_sum(lfvio_ctx *c, int count, const Grid &g, int mode) {
  const size_t st = c->L.total;
  // (from the unrounded chunk count: a slot has pre_gram set iff ITS count exceeds the limit — with the rounded one a window
  // of 241 or 242 chunks launched k_presum for gather lists that do not use its output)
  const int pre = (g.ch_raw > PRE_CHUNK_LIMIT || g.sc > 4 * PRE_GROUP) ? 1 : 0;
  const int groups = (g.sc + PRE_GROUP - 1) / PRE_GROUP;
  if (pre) hipLaunchKernelGGL(k_presum, dim3(NPAIR + (SCHUR_LEN / 256) * groups + 1, count), dim3(256), 0, c->stream, c->d_base, st, mode, groups);
  const Layout &L = c->L;
  const SumArgs sa{(long long)L.sum_off, (long long)L.sum_end_marg, (long long)L.sum_items, (long long)L.gram_part, (long long)L.pairG, (long long)L.imu_out,
                   (long long)L.prior_A};
  hipLaunchKernelGGL(k_sum, dim3(HPP_BLOCKS + SCHUR_LEN / 256 + 1, count), dim3(256), 0, c->stream, c->d_base, st, mode, pre, sa);
}

LinwArgs linw_args(const lfvio_ctx *c) {
  const Layout &L = c->L;
  LinwArgs a;
  a.anc0 = (long long)L.anc[0], a.anc_stride = (long long)(L.anc[1] - L.anc[0]);
  a.pmo0 = (long long)L.pmo[0], a.pmo_stride = (long long)(L.pmo[1] - L.pmo[0]);
  a.Wt = (long long)L.Wt, a.lam[0] = (long long)L.lam[0], a.lam[1] = (long long)L.lam[1];
  a.a = (long long)L.a, a.b = (long long)L.b, a.scale_l = (long long)L.scale_l, a.diag_l = (long long)L.diag_l, a.grad_l = (long long)L.grad_l;
  a.einv_l = (long long)L.einv_l, a.imu_out = (long long)L.imu_out;
  a.Hpp = (long long)(L.xch + (size_t)XOFF_H * 8), a.gp = (long long)(L.xch + (size_t)XOFF_G * 8), a.schur_sum = (long long)(L.xch + (size_t)XOFF_S * 8);
  a.part = (long long)L.schur_part, a.wt_ld = L.capLmBlocks * LM_BLOCK;
  a.asm_tab = c->d_lwt;
  return a;
}
void launch_linw(lfvio_ctx *c, int count, int mode_bits = MODE_SOLVE, bool offs = true) {
  if (offs) hipLaunchKernelGGL(k_linw<true>, dim3(1, count), dim3(LW_THREADS), LW_LDS_BYTES, c->stream, c->d_base, c->L.total, linw_args(c), mode_bits);
  else hipLaunchKernelGGL(k_linw<false>, dim3(1, count), dim3(LW_THREADS), LW_LDS_BYTES, c->stream, c->d_base, c->L.total, linw_args(c), mode_bits);
}

// lw: the pass was linearized by k_linw — H_pp holds the visual terms of its camera part only, the solve adds the rest on load
void launch_solve(lfvio_ctx *c, int count, bool lw = false) {
  const size_t st = c->L.total;
  const long long xo = (long long)c->L.xch, io = (long long)c->L.imu_out, po = (long long)c->L.prior_A;
  if (lw)
    hipLaunchKernelGGL(k_solve_dense<true>, dim3(1, count), dim3(SOLVE_THREADS), SOLVE_LDS, c->stream, c->d_base, st, xo, io, po, (const int *)c->d_asm);
  else
    hipLaunchKernelGGL(k_solve_dense<false>, dim3(1, count), dim3(SOLVE_THREADS), SOLVE_LDS, c->stream, c->d_base, st, xo, io, po, (const int *)nullptr);
}

// A resident batch whose windows all carry a LinwPlan is linearized window by window (kernels_linw.h) instead of role by role.
bool use_linw(lfvio_ctx *c, int count, const Grid &g, int mode) {
  // (the solve passes, and the marginalization's sweep behind them)
  const int m = mode & (MODE_GATED - 1);
  const bool solve = mode == MODE_SOLVE, marg_sweep = m >= MODE_MARG && !(mode & (MODE_DECIDE | MODE_NOCOUNT));
  if (!(solve || marg_sweep) || c->linw_mode == 0 || c->shard_active) return false;
  if (c->linw_mode != 2 && !lin_split(count, g)) return false;
  for (int s = 0; s < count; s++)
    if (!c->info[s].linw_ok) return false;
  return true;
}

// k_setup's launch: grid.x and the bits of its last argument.  A resident batch is bound by the number of workgroups the launch
// dispatches (21 per window, 16 of them for a prior that ONE workgroup handles there: 10 752 workgroups for 512 windows, ~10 ns each),
// so it goes out with the compact grid — state, IMU roots, one prior workgroup, inverse depths — when every resident prior fits
// that workgroup's 4 x 4 tiling (n <= 88: kernels_lin.h).
struct SetupLaunch {
  int gx, bits;
};
SetupLaunch setup_launch(lfvio_ctx *c, int count, int lm, bool zero_wt) {
  bool compact = count >= 8;
  for (int s = 0; compact && s < count; s++) compact = c->info[s].prior_n <= SETUP_TILED_MAXN;
  return {(compact ? 3 : SETUP_WGS) + (lm + 3) / 4, (zero_wt ? 1 : 0) | (compact ? 4 : 0)};
}

// A large single window that carries a group list is linearized group by group (k_linb + k_sumb) instead of role by role; the
// marginalization's sweep of such a window stays with the roles (one launch per call).
bool use_linb(lfvio_ctx *c, int count, const Grid &g, int mode) {
  if (mode != MODE_SOLVE || c->linw_mode == 0 || c->shard_active) return false;
  for (int s = 0; s < count; s++)
    if (!c->info[s].linb_ok) return false;
  return count > 0;
}
// (grid of k_linb: the groups of the largest resident window and the pose side's workgroup, rounded up so that a captured graph
// serves the next window of about that size too — a workgroup past a slot's own count returns at once; part of the graphs' key)
int linb_grid(const lfvio_ctx *c, int count) {
  int ng = 0;
  for (int s = 0; s < count; s++) ng = std::max(ng, c->info[s].linb_ng);
  return (ng + 1 + 63) / 64 * 64;
}
void launch_linb(lfvio_ctx *c, int count, bool offs = true) {
  if (offs) hipLaunchKernelGGL(k_linb<true>, dim3(linb_grid(c, count), count), dim3(LW_THREADS), LW_LDS_BYTES, c->stream, c->d_base, c->L.total, linw_args(c));
  else hipLaunchKernelGGL(k_linb<false>, dim3(linb_grid(c, count), count), dim3(LW_THREADS), LW_LDS_BYTES, c->stream, c->d_base, c->L.total, linw_args(c));
  hipLaunchKernelGGL(k_sumb, dim3(LINB_SUM_GRID, count), dim3(LINB_SUM_THREADS), 0, c->stream, c->d_base, c->L.total, linw_args(c));
}

// speculate: small windows evaluate the steps for radius, radius / 2, radius / 4 in every pass (dev_types.h, SPEC_EXTRA)
// first / last: position of the pass in the sequence being issued (a graph, or a plain run of passes).  For small windows
// the trust-region bookkeeping of a pass rides in the prologue of the NEXT pass's k_lin (MODE_DECIDE, one launch less per
// pass); k_decide itself is only launched behind the last pass, so that the header is final where the sequence ends.
// gauge: the gated gauge fix follows this (last) pass — returns true if it went out with the bookkeeping (k_decide_gauge)
// offs: see launch_lin
bool launch_iteration(lfvio_ctx *c, int count, const Grid &g, int mode, bool speculate = false, bool first = true, bool last = true, bool gauge = false, bool offs = true) {
  const size_t st = c->L.total;
  const bool solve = (mode & (MODE_GATED - 1)) == MODE_SOLVE && !(mode & MODE_GATED);
  // (latency of few windows only: in a resident batch every workgroup of k_lin repeating the decision costs more of the
  // GPU than the launch it saves)
  const bool lw = use_linw(c, count, g, mode), lb = !lw && use_linb(c, count, g, mode);
  const bool merge = solve && !lw && g.lm <= DOGLEG_INLINE_BLOCKS && (size_t)count * (g.lw + (g.ch + 3) / 4 + LFVIO_WINDOW_SIZE + 1) <= LIN_SPLIT_WGS;
  if (lw) {
    // the window-resident sweep: one workgroup per window — pose-side factors, visual sweep, Schur; it counts the pass
    launch_linw(c, count, mode, offs);
  } else if (lb) {
    launch_linb(c, count, offs);
  } else {
    launch_lin(c, count, g, mode | (merge && !first ? MODE_DECIDE : 0), offs);
    launch_sum(c, count, g, mode);
  }
  if ((mode & (MODE_GATED - 1)) == MODE_SOLVE) {
    launch_solve(c, count, lw);  // (k_sumb leaves the complete matrix)
    // small windows: the landmark back-substitution rides inside k_dogleg (one launch less per pass)
    const bool inl = g.lm <= DOGLEG_INLINE_BLOCKS;
    const int spec = speculate && inl ? c->spec_count : 1;
    const int nb = g.lm + LFVIO_WINDOW_SIZE + 1;
    // few small windows: the step and the cost of its candidates in one launch (k_step)
    const bool split = lin_split(count, g) || lw;
    const bool fuse = inl && !split && !c->shard_active && (size_t)count * nb <= 2048;
    // a resident batch on the window-resident path: step, candidate cost and bookkeeping as ONE launch, one workgroup per window
    const bool stepw = lw && inl && spec == 1;
    if (stepw) {
      hipLaunchKernelGGL(k_stepw, dim3(1, count), dim3(STEPW_LAUNCH_THREADS), 0, c->stream, c->d_base, st);
      return false;
    }
    if (!inl && lb) hipLaunchKernelGGL(k_backsub_wt, dim3(g.lm, count), dim3(64), 0, c->stream, c->d_base, st, c->L.capLmBlocks * LM_BLOCK);
    else if (!inl) hipLaunchKernelGGL(k_backsub, dim3(g.lm, count), dim3(64), 0, c->stream, c->d_base, st);
    if (fuse) hipLaunchKernelGGL(k_step, dim3(spec * nb, count), dim3(DOGLEG_INLINE_THREADS), 0, c->stream, c->d_base, st, g.lm, spec);
    else {
      if (inl && (!split || lw)) hipLaunchKernelGGL((k_dogleg<true, true>), dim3(spec, count), dim3(DOGLEG_INLINE_THREADS), 0, c->stream, c->d_base, st, spec);
      else if (inl) hipLaunchKernelGGL((k_dogleg<true, false>), dim3(spec, count), dim3(DOGLEG_INLINE_THREADS), 0, c->stream, c->d_base, st, spec);
      else hipLaunchKernelGGL(k_dogleg<false>, dim3(spec, count), dim3(128), 0, c->stream, c->d_base, st, spec);
    }
    // four lanes per track while the GPU has room for the extra waves (latency of few windows), one when a batch fills it
    if (fuse) {
    } else if ((size_t)count * nb <= 2048)  // (four waves per workgroup: 8 192 waves = eight per SIMD)
      hipLaunchKernelGGL(k_cost<4>, dim3(spec * nb, count), dim3(256), 0, c->stream, c->d_base, st, g.lm, spec);
    else if (spec == 1 && !c->shard_active) {
      hipLaunchKernelGGL((k_cost<1, false>), dim3(g.lm + 1, count), dim3(64), 0, c->stream, c->d_base, st, g.lm, 1);
      hipLaunchKernelGGL(k_cost_imu, dim3((count * LFVIO_WINDOW_SIZE + 63) / 64), dim3(64), 0, c->stream, c->d_base, st, count);
    } else
      hipLaunchKernelGGL(k_cost<1>, dim3(spec * nb, count), dim3(64), 0, c->stream, c->d_base, st, g.lm, spec);
    if (merge && last && gauge) {
      hipLaunchKernelGGL(k_decide_gauge, dim3(1, count), dim3(128), 0, c->stream, c->d_base, st, c->publish ? 1 : 0);
      return true;
    }
    if (!merge || last) hipLaunchKernelGGL(k_decide, dim3(1, count), dim3(64), 0, c->stream, c->d_base, st);
  }
  return false;
}

// number of slots that still need passes (tail = 0: solve not done; tail = 1: gated marginalization not finished)
// out[1]: the largest number of passes a slot has used so far (k_lin counts them)
__global__ void k_pending(char *base, size_t stride, int count, int *out, int tail) {
  int n = 0, used = 0;
  for (int s = threadIdx.x; s < count; s += 64) {
    const Slot *S = (const Slot *)(base + (size_t)s * stride);
    n += (tail ? S->tail_state == 2 : S->tr.done != 0) ? 0 : 1;
    used = max(used, S->passes_used);
  }
  used = __reduce_max_sync(~0ull, used);
  if (threadIdx.x == 0) out[1] = used;
  n = __reduce_add_sync(~0ull, n);
  if (threadIdx.x == 0) *out = n;
}

// a slot that is still not done when the passes are used up ends as it is (NO_CONVERGENCE): the gated tail takes it then
__global__ void k_force_done(char *base, size_t stride, int count) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < count) {
    Slot *S = (Slot *)(base + (size_t)s * stride);
    if (!S->tr.done) S->tr.done = 1;
    if (S->spec_on && S->tail_state == 0) spec_closing(S);  // (the gated gauge fix follows: kernels_spec.h)
  }
}

// Enqueue the trust-region loop for slots [0, count): max_iter Ceres iterations plus spare
// passes for mu-retries (a failed Cholesky consumes a pass but not an iteration).
//   adaptive = false: the whole loop as one static graph, nothing but enqueues (lfvio_batch_optimize_async).
//   adaptive = true (synchronous entry points): the passes go out in chunks of SOLVE_CHUNK and the host reads the number
//     of unfinished slots in between.  A pass costs its ~30 us of launches and first loads whether or not the loop is
//     already done, and with the speculative candidates of small windows nine iterations are four passes, not twelve.
constexpr int SOLVE_CHUNK = 2, MAX_FIRST_PASSES = 12;  // continuation chunk; longest first graph
constexpr int SIDE_ROUNDS = 4;  // rounds of the workers' graph (kernels_spec.h): accepted states a call can have a prior started for
int enqueue_marg(lfvio_ctx *c, int count, int flag, bool standalone, bool gated = false);
// fused_flag >= 0 (adaptive only): gauge fix + marginalization ride in the graph of the first chunk, gated per slot on
// `done` — the common case (every window done within the first chunk) is ONE graph launch and one synchronization; *tail_done
// tells the caller whether anything is left for the (gated) tail graph.
// max_seconds > 0 (adaptive only): Ceres' max_solver_time_in_seconds (estimator.cpp:815-822) — TrustRegionMinimizer tests the
// wall clock at the top of every iteration and stops with NO_CONVERGENCE; here the host tests it between graph launches
// (the only points where it sees the loop) and ends the open slots the same way (k_force_done).  The first graph is sized
// from the previous call as without a cap, shortened only when the cap is tighter than that many passes would take.
// early (adaptive, fused, one window with a mailbox): return as soon as the solution is in the mailbox — the rest of the graph
// (the marginalization) is still running then and c->inflight says so.
int enqueue_solve(lfvio_ctx *c, int count, int max_iter, bool adaptive, int fused_flag = -1, bool *tail_done = nullptr,
                  double max_seconds = -1.0, bool early = false) {
  if (int rc = join_inflight(c, early)) return rc;
  const Grid g = grid_for(c, count);
  const int passes = std::max(max_iter, 0) + 4;
  if (adaptive && c->use_graph) {
    const bool speculate = (size_t)count * (g.lm + LFVIO_WINDOW_SIZE + 1) <= 512;
    // How many candidates a pass prepares follows the previous call (a stream of windows from one estimator is steady): where a
    // pass covered two iterations or more — most steps rejected: the bench window's nine iterations take four passes with three
    // candidates, three with four — the fourth candidate saves a pass; where nearly every step is accepted it would only be
    // evaluated (measured with a fixed count: 0.532 / 0.511 ms resident with 3 / 4, 0.950 / 0.963 ms on the stream).
    if (speculate && !c->fixed_spec) c->spec_count = (c->last_iters > 0 && c->last_iters >= 2 * c->last_passes) ? std::min(4, 1 + SPEC_EXTRA) : 3;
    if (!c->d_pending) {
      HIPCHK(c, hipMalloc((void **)&c->d_pending, 256));
      HIPCHK(c, hipHostMalloc((void **)&c->h_pending, 256, hipHostMallocDefault));
    }
    const int lwk = (use_linw(c, count, g, MODE_SOLVE) ? 1 : use_linb(c, count, g, MODE_SOLVE) ? 2 + 4 * linb_grid(c, count) : 0);
    const int offs = slots_offs(c, count);
    const int setup_bits = setup_launch(c, count, g.lm, lwk == 1).bits;  // (the compact grid of k_setup is part of the captured launch)
    if (c->k_batch != count || c->k_lm != g.lm || c->k_ch != g.ch || c->k_sc != g.sc || c->k_spec != (int)speculate || c->k_linw != lwk || c->k_offs != offs || c->k_setup != setup_bits) {
      destroy_graph(c, c->k_batch == count && ((c->k_offs ^ offs) & 2) == 0);  // (the workers' graphs: per context, but for the fixed-extrinsic bit)
      c->k_batch = count, c->k_lm = g.lm, c->k_ch = g.ch, c->k_sc = g.sc, c->k_spec = (int)speculate, c->k_linw = lwk, c->k_offs = offs, c->k_setup = setup_bits;
    }
    if (tail_done) *tail_done = false;
    const bool fuse = fused_flag >= 0 && fused_flag < 2;
    // The first graph carries as many passes as the previous call on this context needed (a stream of windows from one
    // estimator is steady: the bench window takes 4, windows whose steps are mostly accepted 5 to 8), then — fused —
    // the gated gauge fix + marginalization; whatever is still pending afterwards continues in chunks of SOLVE_CHUNK.
    const bool capped = max_seconds > 0.0;
    const auto t_start = std::chrono::steady_clock::now();
    // With a cap the first graph is still the predicted one — a cap of SOLVER_TIME = 0.04 s (the shipped default) is a few
    // hundred passes away and must not cost the call its single launch — unless the cap is so tight that the predicted
    // graph could overrun it: then no more passes than fit (at the measured time per pass), down to a chunk of SOLVE_CHUNK.
    int first_passes = std::min(std::max(c->fixed_passes > 0 ? c->fixed_passes : c->predict_passes, 1), std::min(passes, MAX_FIRST_PASSES));
    if (capped) first_passes = std::max(std::min(SOLVE_CHUNK, passes), std::min(first_passes, (int)std::min(max_seconds / c->pass_seconds, 1e6)));
    const int sv = speculate && c->spec_count > 3 ? 1 : 0;  // (both variants stay captured: a stream may alternate)
    hipGraphExec_t &first_graph = c->first[sv][c->publish ? 1 : 0][fuse ? 1 + fused_flag : 0][first_passes];
    auto capture = [&](hipGraphExec_t *out, bool setup, int npass, int tail_flag) -> int {
      hipGraph_t graph;
      int rc = LFVIO_OK;
      HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      CaptureGuard guard(c->stream);
      if (setup)
        { const SetupLaunch sl_ = setup_launch(c, count, g.lm, use_linw(c, count, g, MODE_SOLVE)); hipLaunchKernelGGL(k_setup, dim3(sl_.gx, count), dim3(256), 0, c->stream, c->d_base, c->L.total, MODE_SOLVE, sl_.bits); }
      bool gauged = false;
      // (the sweep behind k_setup linearizes at the uploaded state: the only point of a call that can hold a quaternion off the unit sphere,
      // a fixed extrinsic aside — slots_offs)
      for (int it = 0; it < npass; it++)
        gauged = launch_iteration(c, count, g, MODE_SOLVE, speculate, it == 0, it == npass - 1, tail_flag >= 0, (offs & 2) || (setup && it == 0 && (offs & 1)));
      if (tail_flag >= 0) {
        if (!gauged) {
          hipLaunchKernelGGL(k_gauge, dim3(1 + (g.lm + 1) / 2, count), dim3(128), 0, c->stream, c->d_base, c->L.total, 1);
          if (c->publish) hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, c->stream, c->d_base, c->L.total);  // (k_decide_gauge does it itself)
        }
        rc = enqueue_marg(c, count, tail_flag, false, true);
      }
      if (tail_flag >= 0 && count == 1) {
        // one window: its {tail_state, passes_used} pair is the answer — copied as it is, no k_pending launch
        HIPCHK(c, hipMemcpyAsync(c->h_pending + 2, c->d_base + offsetof(Slot, tail_state), 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));  // (.. chain_err)
      } else {
        hipLaunchKernelGGL(k_pending, dim3(1), dim3(64), 0, c->stream, c->d_base, c->L.total, count, c->d_pending, tail_flag >= 0 ? 1 : 0);
        HIPCHK(c, hipMemcpyAsync(c->h_pending, c->d_pending, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
      }
      HIPCHK(c, guard.end(&graph));
      if (rc) {
        (void)hipGraphDestroy(graph);
        return rc;
      }
      HIPCHK(c, hipGraphInstantiate(out, graph, nullptr, nullptr, 0));
      HIPCHK(c, hipGraphDestroy(graph));
      return LFVIO_OK;
    };
    if (!first_graph) {
      const int rc = capture(&first_graph, true, first_passes, fuse ? fused_flag : -1);
      if (rc) return rc;
    }
    if (!c->chunk[sv]) {
      const int rc = capture(&c->chunk[sv], false, SOLVE_CHUNK, -1);
      if (rc) return rc;
    }
    // Workers for the marginalization run ahead (kernels_spec.h): one small window on the merged launch sequence (its loop ends in
    // k_decide_gauge), with a shadow slot and a mailbox.  SIDE_ROUNDS rounds of [k_spec_begin, k_lin, k_sum, k_marg_solve] on the shadow
    // slot, captured once per marginalization flag and hand-over variant; a round that finds nothing to do is four launches that return.
    // (not behind a device-chained upload: there the marginalization already runs beside the host's packing of the next window, and the next
    // window's upload would have to wait for a worker instead of following the stream)
    const bool ahead = fuse && count == 1 && c->info[0].spec_on && c->shadow && c->marg_ahead  && !c->shard_active && !c->pipelined &&
                       g.lm <= DOGLEG_INLINE_BLOCKS && g.ch_raw <= PRE_CHUNK_LIMIT && g.sc <= 4 * PRE_GROUP && !use_linw(c, count, g, MODE_SOLVE) &&
                       !use_linb(c, count, g, MODE_SOLVE);
    for (int wk = 0; ahead && wk < lfvio_ctx::WORKERS; wk++) {
      hipGraphExec_t *side_graph = &c->side[wk][fused_flag][c->publish ? 1 : 0];
      if (*side_graph) continue;
      hipStream_t ss = c->sstream[wk];
      hipGraph_t graph;
      HIPCHK(c, hipStreamBeginCapture(ss, hipStreamCaptureModeThreadLocal));
      CaptureGuard guard(ss);
      const size_t back = (size_t)(c->batch + wk) * c->L.total;
      char *sh = c->d_base + back;
      // launch dimensions for the largest window the merged sequence takes (spare workgroups test their index against the slot's own
      // counts and return): one capture serves every window of the context
      const Layout &L = c->L;
      const int mode = (MODE_MARG + fused_flag) | MODE_GATED, gram_wgs = (L.capChunks + 3) / 4, lw_cap = 2 * DOGLEG_INLINE_BLOCKS;
      const int pre = 0, groups = 1;  // (k_presum is for windows of thousands of landmarks)
      // (the gather lists are inputs: the shadow's copies of those members lead back into slot 0, and so do these offsets)
      const SumArgs sa{(long long)L.sum_off - (long long)back, (long long)L.sum_end_marg - (long long)back, (long long)L.sum_items - (long long)back,
                       (long long)L.gram_part, (long long)L.pairG, (long long)L.imu_out, (long long)L.prior_A - (long long)back};
      for (int r = 0; r < SIDE_ROUNDS; r++) {
        hipLaunchKernelGGL(k_spec_begin, dim3(1), dim3(128), 0, ss, sh, back);
        // (a re-anchored state: on the sphere but for a fixed extrinsic)
        if (offs & 2) hipLaunchKernelGGL((k_lin<LIN_ROLE_ALL, true>), dim3(lw_cap + gram_wgs + LFVIO_WINDOW_SIZE + 1, 1), dim3(LIN_THREADS), 0, ss, sh, back, mode, lw_cap, gram_wgs);
        else hipLaunchKernelGGL((k_lin<LIN_ROLE_ALL, false>), dim3(lw_cap + gram_wgs + LFVIO_WINDOW_SIZE + 1, 1), dim3(LIN_THREADS), 0, ss, sh, back, mode, lw_cap, gram_wgs);
        if (pre) hipLaunchKernelGGL(k_presum, dim3(NPAIR + (SCHUR_LEN / 256) * groups + 1, 1), dim3(256), 0, ss, sh, back, mode, groups);
        hipLaunchKernelGGL(k_sum, dim3(HPP_BLOCKS + SCHUR_LEN / 256 + 1, 1), dim3(256), 0, ss, sh, back, mode, pre, sa);
        hipLaunchKernelGGL(k_marg_solve<true>, dim3(1, 1), dim3(MARG_THREADS), MARG_LDS, ss, sh, back,
                           fused_flag | (c->force_eig ? 256 : 0) | 512 | (c->publish ? 1024 : 0));
      }
      HIPCHK(c, guard.end(&graph));
      HIPCHK(c, hipGraphInstantiate(side_graph, graph, nullptr, nullptr, 0));
      HIPCHK(c, hipGraphDestroy(graph));
    }
    c->side_launched = false, c->side_known = false;
    c->stat_chunks = 0;
    for (int done_passes = 0; done_passes < passes;) {
      c->stat_chunks++;
      const auto t_launch = std::chrono::steady_clock::now();
      const bool watch = early && done_passes == 0 && fuse && c->publish;
      if (watch) __atomic_store_n((int *)c->h_mail + 1, 0, __ATOMIC_RELAXED), __atomic_store_n((int *)c->h_mail, 0, __ATOMIC_RELEASE);
      // (ticket 0: a call without workers — rounds left over from an earlier call must not take its states: spec_publish)
      if (done_passes == 0 && c->h_mail && count == 1) ((volatile int *)c->h_mail)[6] = ahead ? (c->side_ticket = (c->side_ticket == 0x7fffffff ? 1 : c->side_ticket + 1)) : 0;
      HIPCHK(c, hipGraphLaunch(done_passes == 0 ? first_graph : c->chunk[sv], c->stream));
      if (ahead && done_passes == 0) {  // (behind the loop's graph: on a shared hardware queue the workers would otherwise wait in front of it)
        for (int wk = 0; wk < lfvio_ctx::WORKERS; wk++) HIPCHK(c, hipGraphLaunch(c->side[wk][fused_flag][c->publish ? 1 : 0], c->sstream[wk]));
        c->side_launched = true, c->stat_ahead_calls++;
      }
      if (c->pipelined && done_passes == 0) c->up_us[1] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_launch).count() * 1e6;  // (debug: lfvio_debug_upload_times)
      if (watch && wait_early(c)) {  // the window was done inside the first graph: its tail follows in the same graph
        c->inflight = true, c->inflight_first = true;
        c->side_known = c->side_launched;
        if (tail_done) *tail_done = true;
        return LFVIO_OK;
      }
      HIPCHK(c, hipStreamSynchronize(c->stream));
      const bool first = done_passes == 0;
      if (!first) {  // time per pass, for sizing a capped call's first graph (continuation chunks carry nothing but passes)
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_launch).count() / SOLVE_CHUNK;
        c->pass_seconds = 0.75 * c->pass_seconds + 0.25 * dt;
      }
      done_passes += first ? first_passes : SOLVE_CHUNK;
      if (first && fuse && count == 1) c->h_pending[0] = c->h_pending[2] >= 2 ? 0 : 1, c->h_pending[1] = c->h_pending[3], c->last_iters = c->h_pending[4];  // (3: the prior is a worker's)
      if (first && fuse && count == 1 && c->h_pending[0] == 0) c->side_known = c->side_launched;
      if (first && fuse && count == 1 && c->h_pending[5]) {
        c->err = CHAIN_ERR_TEXT;
        return LFVIO_ERR_DEVICE;
      }
      if (c->h_pending[0] == 0) {
        if (tail_done && fuse && first) *tail_done = true;
        break;
      }
      if (capped && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() >= max_seconds) {
        // "Maximum solver time reached": the open slots end where they are, termination stays NO_CONVERGENCE
        hipLaunchKernelGGL(k_force_done, dim3((count + 63) / 64), dim3(64), 0, c->stream, c->d_base, c->L.total, count);
        HIPCHK(c, hipStreamSynchronize(c->stream));
        break;
      }
    }
    c->last_passes = std::max(c->h_pending[1], 1);  // passes the slowest window has used
    predict(c);
    HIPCHK(c, hipGetLastError());
    // (a call that ended inside its first graph: the prior may be on its way from a worker; one that continues with a tail graph
    // is waited for by the join in front of whatever comes next)
    if (tail_done && *tail_done)
      if (int rc = wait_side(c)) return rc;
    return LFVIO_OK;
  }
  { const SetupLaunch sl_ = setup_launch(c, count, grid_for(c, count).lm, use_linw(c, count, g, MODE_SOLVE)); hipLaunchKernelGGL(k_setup, dim3(sl_.gx, count), dim3(256), 0, c->stream, c->d_base, c->L.total, MODE_SOLVE, sl_.bits); }
  const int offs_s = slots_offs(c, count);  // (the pass behind k_setup sweeps at the uploaded state: launch_lin)
  if (c->use_graph) {
    const int lwg = (use_linw(c, count, g, MODE_SOLVE) ? 1 : use_linb(c, count, g, MODE_SOLVE) ? 2 + 4 * linb_grid(c, count) : 0);
    if (!c->graph || c->g_batch != count || c->g_lm != g.lm || c->g_ch != g.ch || c->g_sc != g.sc || c->g_iters != passes || c->g_linw != lwg || c->g_offs != offs_s) {
      if (c->graph) (void)hipGraphExecDestroy(c->graph), c->graph = nullptr;
      hipGraph_t graph;
      HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      CaptureGuard guard(c->stream);
      for (int it = 0; it < passes; it++) launch_iteration(c, count, g, MODE_SOLVE, false, it == 0, it == passes - 1, false, (offs_s & 2) || (it == 0 && (offs_s & 1)));
      HIPCHK(c, guard.end(&graph));
      HIPCHK(c, hipGraphInstantiate(&c->graph, graph, nullptr, nullptr, 0));
      HIPCHK(c, hipGraphDestroy(graph));
      c->g_batch = count, c->g_lm = g.lm, c->g_ch = g.ch, c->g_sc = g.sc, c->g_iters = passes, c->g_linw = lwg, c->g_offs = offs_s;
    }
    HIPCHK(c, hipGraphLaunch(c->graph, c->stream));
  } else {
    for (int it = 0; it < passes; it++) launch_iteration(c, count, g, MODE_SOLVE, false, it == 0, it == passes - 1, false, (offs_s & 2) || (it == 0 && (offs_s & 1)));
  }
  HIPCHK(c, hipGetLastError());
  return LFVIO_OK;
}

int enqueue_marg(lfvio_ctx *c, int count, int flag, bool standalone, bool gated) {
  if (int rc = join_inflight(c, gated)) return rc;  // (gated: part of a graph of lfvio_batch_optimize*, which has joined — or may go out behind a pipelined upload)
  const Grid g = grid_for(c, count);
  const int mode = (MODE_MARG + flag) | (gated ? MODE_GATED : 0);
  if (standalone)
    { const SetupLaunch sl_ = setup_launch(c, count, grid_for(c, count).lm, use_linw(c, count, g, mode)); hipLaunchKernelGGL(k_setup, dim3(sl_.gx, count), dim3(256), 0, c->stream, c->d_base, c->L.total, mode, sl_.bits); }
  // (standalone: the sweep is at the uploaded state; gated: at the re-anchored solution, on the sphere but for a fixed extrinsic)
  launch_iteration(c, count, g, mode, false, true, true, false, standalone ? slots_offs(c, count) != 0 : (slots_offs(c, count) & 2) != 0);
  hipLaunchKernelGGL(k_marg_solve<false>, dim3(1, count), dim3(MARG_THREADS), MARG_LDS, c->stream, c->d_base, c->L.total,
                     flag | (c->force_eig ? 256 : 0) | (gated ? 512 : 0) | (gated && c->publish ? 1024 : 0));
  HIPCHK(c, hipGetLastError());
  return LFVIO_OK;
}

// Results come back through one pinned block [x[2] | tr | lam[0] | lam[1] | prior]: every copy a call needs is enqueued and
// the stream is synchronized ONCE (a round trip costs ~12 us; the drop-in call used to make four).  Which of the two
// inverse-depth buffers is current is only known from `tr`, so small windows fetch both; windows beyond ONE_TRIP_LM
// landmarks take a second trip for the one that matters instead of moving megabytes for nothing.
constexpr int ONE_TRIP_LM = 8192;
struct Fetched {
  const FrameState *xs;
  const TRState *tr;
  const double *lam[2];
  LfvioPrior *prior;
};

int fetch(lfvio_ctx *c, int slot, bool want_sol, bool want_prior, Fetched *f) {
  if (int rc = join_inflight(c)) return rc;
  const Layout &L = c->L;
  char *d = c->d_base + (size_t)slot * L.total;
  const SlotHostInfo &info = c->info[slot];
  if (!info.resident) {
    c->err = "slot not uploaded";
    return LFVIO_ERR_ARG;
  }
  char *hd = c->h_down;
  const size_t hdr = sizeof(FrameState) * 2 + sizeof(TRState), lam_bytes = (size_t)info.N * 8;
  char *h_lam0 = hd + hdr, *h_lam1 = h_lam0 + align_up(lam_bytes + 8, 64), *h_prior = h_lam1 + align_up(lam_bytes + 8, 64);
  f->xs = (const FrameState *)hd, f->tr = (const TRState *)(hd + sizeof(FrameState) * 2);
  f->lam[0] = (const double *)h_lam0, f->lam[1] = (const double *)h_lam1, f->prior = (LfvioPrior *)h_prior;
  const bool both = info.N > 0 && info.N <= ONE_TRIP_LM;
  if (want_sol) {
    HIPCHK(c, hipMemcpyAsync(hd, d + offsetof(Slot, x), hdr, hipMemcpyDeviceToHost, c->stream));
    if (both) {
      HIPCHK(c, hipMemcpyAsync(h_lam0, d + L.lam[0], lam_bytes, hipMemcpyDeviceToHost, c->stream));
      HIPCHK(c, hipMemcpyAsync(h_lam1, d + L.lam[1], lam_bytes, hipMemcpyDeviceToHost, c->stream));
    }
  }
  if (want_prior) {
    // the dimension of the prior a marginalization of this window can produce is known since the upload (plan_marg)
    const size_t head = offsetof(LfvioPrior, linearized_jacobians), n = (size_t)info.marg_n;
    HIPCHK(c, hipMemcpyAsync(h_prior, d + offsetof(Slot, prior_out), head + sizeof(double) * n * n, hipMemcpyDeviceToHost, c->stream));
    if (n)
      HIPCHK(c, hipMemcpyAsync(h_prior + offsetof(LfvioPrior, linearized_residuals), d + offsetof(Slot, prior_out) + offsetof(LfvioPrior, linearized_residuals),
                               sizeof(double) * n, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (want_sol && !both && info.N > 0 && !f->tr->error) {
    HIPCHK(c, hipMemcpyAsync(f->tr->cur ? h_lam1 : h_lam0, d + L.lam[f->tr->cur], lam_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return LFVIO_OK;
}

// nothing is written to the caller's outputs before the result is known to be usable (the header promises untouched
// outputs on error): check_* first, then unpack_*
int check_solution(lfvio_ctx *c, int slot, const Fetched &f) {
  const SlotHostInfo &info = c->info[slot];
  const TRState *tr = f.tr;
  if (tr->error) {
    c->err = "non-finite cost";
    return tr->error;
  }
  const FrameState &x = f.xs[tr->cur];
  const double *lam = f.lam[tr->cur];
  bool finite = std::isfinite(tr->x_cost);
  for (int k = 0; finite && k < (int)(sizeof(FrameState) / 8); k++) finite = std::isfinite(((const double *)&x)[k]);
  for (int dl = 0; finite && dl < info.N; dl++) finite = std::isfinite(lam[dl]);
  if (!finite) {
    c->err = "non-finite state";
    return LFVIO_ERR_NONFINITE;
  }
  return LFVIO_OK;
}

void unpack_solution(lfvio_ctx *c, int slot, const Fetched &f, LfvioSolution *out) {
  const SlotHostInfo &info = c->info[slot];
  const TRState *tr = f.tr;
  const FrameState &x = f.xs[tr->cur];
  const double *lam = f.lam[tr->cur];
  std::memcpy(out->para_pose, x.pose, sizeof x.pose);
  std::memcpy(out->para_speed_bias, x.sb, sizeof x.sb);
  std::memcpy(out->para_ex_pose, x.ex, sizeof x.ex);
  out->para_td = x.td;
  if (out->inv_depth)
    for (int dl = 0; dl < info.N; dl++) out->inv_depth[info.perm[dl]] = lam[dl];
  out->num_iterations = tr->trace_len;
  out->num_successful_steps = tr->num_succ;
  out->num_unsuccessful_steps = tr->num_unsucc;
  out->termination = tr->termination;
  out->initial_cost = tr->initial_cost;
  out->final_cost = tr->x_cost;
  std::memset(out->trace, 0, sizeof out->trace);
  for (int k = 0; k < tr->trace_len && k < LFVIO_MAX_TRACE; k++) out->trace[k] = tr->trace[k];
}

// returns LFVIO_OK with *pass = true when nothing was marginalized and the input prior stays
int check_prior(lfvio_ctx *c, int slot, const Fetched &f, bool *pass) {
  const SlotHostInfo &info = c->info[slot];
  const LfvioPrior *hp = f.prior;
  *pass = hp->valid == -1;
  if (*pass) return LFVIO_OK;
  const int n = hp->n;
  if (hp->valid != 1 || n <= 0 || n > LFVIO_MAX_PRIOR_DIM || n > info.marg_n) {
    c->err = "marginalization produced no prior";
    return LFVIO_ERR_DEVICE;
  }
  bool finite = true;
  for (int k = 0; finite && k < n * n; k++) finite = std::isfinite(hp->linearized_jacobians[k]);
  for (int k = 0; finite && k < n; k++) finite = std::isfinite(hp->linearized_residuals[k]);
  if (!finite) {
    c->err = "non-finite prior";
    return LFVIO_ERR_NONFINITE;
  }
  return LFVIO_OK;
}

void unpack_prior(lfvio_ctx *c, int slot, const Fetched &f, bool pass, LfvioPrior *out) {
  const SlotHostInfo &info = c->info[slot];
  if (pass) {
    if (info.has_in_prior) copy_prior(out, &info.in_prior);
    else out->valid = 0;
    return;  // (a prior whose values stayed on the device: fetch_device_prior below, by the callers that can synchronize)
  }
  const LfvioPrior *hp = f.prior;
  const int n = hp->n;
  std::memcpy(out, hp, offsetof(LfvioPrior, linearized_jacobians));
  std::memcpy(out->linearized_jacobians, hp->linearized_jacobians, sizeof(double) * n * n);
  std::memcpy(out->linearized_residuals, hp->linearized_residuals, sizeof(double) * n);
}

// The input prior of a window that took it over on the device (lfvio_batch_upload_chained_device) and whose own marginalization
// passes it through: its values have never been on the host — they are read out of the slot now (a synchronization and three
// small copies, in a case the reference meets once per MARGIN_SECOND_NEW without a prior on the newest pose).
int fetch_device_prior(lfvio_ctx *c, int slot, LfvioPrior *out) {
  SlotHostInfo &info = c->info[slot];
  if (!info.in_prior_device) return LFVIO_OK;
  const char *d = c->d_base + (size_t)slot * c->L.total;
  LfvioPrior *p = &info.in_prior;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(p->linearized_jacobians, d + c->L.prior_J, sizeof(double) * p->n * p->n, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(p->linearized_residuals, d + c->L.prior_r, sizeof(double) * p->n, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(p->block_x0, d + offsetof(Slot, prior_x0), sizeof(double) * 9 * p->num_blocks, hipMemcpyDeviceToHost));
  info.in_prior_device = false;
  if (out) copy_prior(out, p);
  return LFVIO_OK;
}

int download(lfvio_ctx *c, int slot, LfvioSolution *sol, LfvioPrior *prior) {
  Fetched f;
  int rc = fetch(c, slot, sol != nullptr, prior != nullptr, &f);
  if (rc) return rc;
  bool pass = false;
  if (sol && (rc = check_solution(c, slot, f))) return rc;
  if (prior && (rc = check_prior(c, slot, f, &pass))) return rc;
  if (sol) unpack_solution(c, slot, f, sol);
  if (prior && pass && (rc = fetch_device_prior(c, slot, nullptr))) return rc;
  if (prior) unpack_prior(c, slot, f, pass, prior);
  return LFVIO_OK;
}
int download_solution(lfvio_ctx *c, int slot, LfvioSolution *out) { return download(c, slot, out, nullptr); }
int download_prior(lfvio_ctx *c, int slot, LfvioPrior *out) { return download(c, slot, nullptr, out); }

}  // namespace

extern "C" {

const char *lfvio_version(void) { return "lfvio-hip 0.1 (gfx950, FP64)"; }

lfvio_ctx *lfvio_create(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  lfvio_ctx *c = new lfvio_ctx();
  c->device = device;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return nullptr;
  }
  {
    // The stream of the feature steps must not share a hardware queue with `stream`, or its kernels wait behind the tail they
    // are meant to run beside: the runtime deals its (four, by default) hardware queues to streams round-robin PER PRIORITY
    // LEVEL, so with a second context in the process the two streams of one context can land on the same queue (measured:
    // lfvio_shift_depth 45 us -> 216 us behind a marginalization).  A stream of another priority comes from another pool.
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (hipStreamCreateWithPriority(&c->fstream, hipStreamNonBlocking, greatest) != hipSuccess) c->fstream = nullptr;  // (falls back to `stream`)
    // ... and the workers of the marginalization run ahead (kernels_spec.h) from the third pool; without three levels there are none
    for (int wk = 0; wk < lfvio_ctx::WORKERS; wk++)
      if (least == greatest || hipStreamCreateWithPriority(&c->sstream[wk], hipStreamNonBlocking, least) != hipSuccess) c->sstream[wk] = nullptr;
  }
  // the mailbox: fine-grained host memory mapped into the device's address space.  Without it begin() simply waits for the end.
  if (hipHostMalloc((void **)&c->h_mail, MAIL_BYTES, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
    std::memset(c->h_mail, 0, MAIL_BYTES);
    if (hipHostGetDevicePointer((void **)&c->d_mail, c->h_mail, 0) != hipSuccess) c->d_mail = nullptr;
  } else {
    c->h_mail = nullptr;
    (void)hipGetLastError();
  }
  // kernels that need more than the default 64 KiB of LDS
  (void)hipFuncSetAttribute((const void *)k_solve_dense<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SOLVE_LDS);
  (void)hipFuncSetAttribute((const void *)k_solve_dense<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SOLVE_LDS);
  (void)hipFuncSetAttribute((const void *)k_linw<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LW_LDS_BYTES);
  (void)hipFuncSetAttribute((const void *)k_linw<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LW_LDS_BYTES);
  (void)hipFuncSetAttribute((const void *)k_linb<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LW_LDS_BYTES);
  (void)hipFuncSetAttribute((const void *)k_linb<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LW_LDS_BYTES);
  {  // static table of k_linw's phase 3: where each packed camera entry of H_pp (then each camera-side gradient entry) sits in the LDS accumulators
    std::vector<int> tab(SUM_VIS);
    for (int e = 0; e < SUM_VIS; e++) {
      int r, cc;
      if (e < SUM_VIS_PACKED) {
        r = 0;
        while ((r + 1) * (r + 2) / 2 <= e) r++;
        cc = e - r * (r + 1) / 2;
      } else {
        r = cc = e - SUM_VIS_PACKED;
      }
      const int fr = r < 66 ? r / 6 : 11, fc = cc < 66 ? cc / 6 : 11, lr = r < 66 ? r - 6 * fr : r - 66, lc = cc < 66 ? cc - 6 * fc : cc - 66;
      int at;
      if (e >= SUM_VIS_PACKED) at = LW_G + cc;
      else if (fr == fc && fr < 11) at = LW_D + 21 * fr + lw_tri(6, lc, lr);
      else if (fr < 11) at = (LW_OFF0 + 36 * lw_pidx(fc, fr) + lc * 6 + lr) | LWT_ABS | (fc << 20) | (fr << 24);  // (frames of the block: k_linb)
      else if (fc < 11) at = LW_FX + 42 * fc + lc * 7 + lr;
      else at = LW_XX + lw_tri(7, lc, lr);
      auto is_ex = [](int q) { return q >= 66 && q < 72; };
      if (is_ex(r) || is_ex(cc)) at |= LWT_EX;
      if (r == 72 || cc == 72) at |= LWT_TD;
      tab[e] = at;
    }
    if (hipMalloc((void **)&c->d_lwt, sizeof(int) * tab.size()) != hipSuccess ||
        hipMemcpy(c->d_lwt, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
      lfvio_destroy(c);
      return nullptr;
    }
  }
  {  // static scatter table of the assembling solve: packed visual entry -> tile address; IMU block entries, even factors then odd
    std::vector<int> tab(ASM_LEN);
    for (int r = 0, e = 0; r < KC; r++)
      for (int cc = 0; cc <= r; cc++, e++) tab[e] = asm_lidx(r, cc);
    auto glob = [](int p, int f) { return p < 6 ? 6 * f + p : p < 15 ? KC + 9 * f + (p - 6) : p < 21 ? 6 * (f + 1) + (p - 15) : KC + 9 * (f + 1) + (p - 21); };
    for (int par = 0; par < 2; par++) {
      std::vector<std::pair<int, int>> ent;  // (tile address, source index)
      for (int f = par; f < LFVIO_WINDOW_SIZE; f += 2)
        for (int p = 0; p < 30; p++)
          for (int q = 0; q < 30; q++)
            if (glob(p, f) >= glob(q, f)) ent.push_back({asm_lidx(glob(p, f), glob(q, f)), f * IMU_OUT + p * 30 + q});
      std::sort(ent.begin(), ent.end());
      for (size_t k = 0; k < ent.size(); k++) tab[ASM_VIS + par * ASM_IMU_HALF + k] = ent[k].second | (ent[k].first << 16);
    }
    if (hipMalloc((void **)&c->d_asm, sizeof(int) * tab.size()) != hipSuccess ||
        hipMemcpy(c->d_asm, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) {
      lfvio_destroy(c);
      return nullptr;
    }
  }
  (void)hipFuncSetAttribute((const void *)k_marg_solve<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MARG_LDS);
  (void)hipFuncSetAttribute((const void *)k_marg_solve<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)MARG_LDS);
  return c;
}

void lfvio_destroy(lfvio_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  destroy_graph(c);
  if (c->d_base) (void)hipFree(c->d_base);
  if (c->d_feat) (void)hipFree(c->d_feat);
  if (c->h_feat) (void)hipHostFree(c->h_feat);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->h_down) (void)hipHostFree(c->h_down);
  if (c->d_asm) (void)hipFree(c->d_asm);
  if (c->d_lwt) (void)hipFree(c->d_lwt);
  if (c->d_pending) (void)hipFree(c->d_pending);
  if (c->h_pending) (void)hipHostFree(c->h_pending);
  if (c->h_flags) {
    (void)hipHostFree(c->h_flags);
    for (auto &e : c->flag_event) (void)hipEventDestroy(e);
  }
  if (c->stage_event) (void)hipEventDestroy(c->stage_event);
  if (c->h_mail) (void)hipHostFree(c->h_mail);
  if (c->fstream) (void)hipStreamDestroy(c->fstream);
  for (auto &st : c->sstream)
    if (st) (void)hipStreamDestroy(st);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *lfvio_last_error(const lfvio_ctx *c) { return c ? c->err.c_str() : "null context"; }
void *lfvio_stream(lfvio_ctx *c) { return c ? (void *)c->stream : nullptr; }

int lfvio_solve(lfvio_ctx *c, const LfvioWindow *in, LfvioSolution *out) {
  if (!c || !in || !out) return LFVIO_ERR_ARG;
  (void)hipSetDevice(c->device);
  int rc = reserve(c, 1, in->num_landmarks, in->num_observations);
  if (rc) return rc;
  if ((rc = upload_window(c, 0, in))) return rc;
  if ((rc = enqueue_solve(c, 1, in->max_num_iterations, true, -1, nullptr, in->max_solver_time_in_seconds))) return rc;
  return download_solution(c, 0, out);
}

int lfvio_marginalize(lfvio_ctx *c, const LfvioWindow *in, int flag, LfvioPrior *out) {
  if (!c || !in || !out || (flag != LFVIO_MARGIN_OLD && flag != LFVIO_MARGIN_SECOND_NEW)) return LFVIO_ERR_ARG;
  (void)hipSetDevice(c->device);
  int rc = reserve(c, 1, in->num_landmarks, in->num_observations);
  if (rc) return rc;
  if ((rc = upload_window(c, 0, in))) return rc;
  if ((rc = enqueue_marg(c, 1, flag, true))) return rc;
  return download_prior(c, 0, out);
}

int lfvio_batch_reserve(lfvio_ctx *c, int batch, int max_landmarks, int max_observations) {
  if (!c || batch <= 0) return LFVIO_ERR_ARG;
  (void)hipSetDevice(c->device);
  return reserve(c, batch, max_landmarks, max_observations);
}

int lfvio_batch_upload(lfvio_ctx *c, int slot, const LfvioWindow *in) {
  if (!c || !in || slot < 0 || slot >= c->batch) return LFVIO_ERR_ARG;
  if (in->num_landmarks > c->L.maxN || in->num_observations > c->L.maxM) {
    c->err = "window larger than the reserved capacity";
    return LFVIO_ERR_ARG;
  }
  (void)hipSetDevice(c->device);
  return upload_window(c, slot, in);
}

int lfvio_batch_upload_chained_device(lfvio_ctx *c, int slot, const LfvioWindow *in) {
  if (!c || !in || slot < 0 || slot >= c->batch) return LFVIO_ERR_ARG;
  if (in->num_landmarks > c->L.maxN || in->num_observations > c->L.maxM) {
    c->err = "window larger than the reserved capacity";
    return LFVIO_ERR_ARG;
  }
  (void)hipSetDevice(c->device);
  return upload_window(c, slot, in, 0, 1, nullptr, true);
}

int lfvio_batch_upload_chained(lfvio_ctx *c, int slot, const LfvioWindow *in, LfvioPrior *prior_io) {
  if (!c || !in || !prior_io || slot < 0 || slot >= c->batch) return LFVIO_ERR_ARG;
  if (in->num_landmarks > c->L.maxN || in->num_observations > c->L.maxM) {
    c->err = "window larger than the reserved capacity";
    return LFVIO_ERR_ARG;
  }
  (void)hipSetDevice(c->device);
  return upload_window(c, slot, in, 0, 1, prior_io);
}

static int batch_optimize_impl(lfvio_ctx *c, int count, int marg_flag, bool adaptive, bool early = false) {
  if (!c || count <= 0 || count > c->batch) return LFVIO_ERR_ARG;
  if (marg_flag != LFVIO_MARGIN_OLD && marg_flag != LFVIO_MARGIN_SECOND_NEW) {  // (the flag indexes the slot's two marginalization plans)
    c->err = "marg_flag is neither LFVIO_MARGIN_OLD nor LFVIO_MARGIN_SECOND_NEW";
    return LFVIO_ERR_ARG;
  }
  (void)hipSetDevice(c->device);
  if (int rc = join_inflight(c, early && count == 1)) return rc;
  c->has_held = false;  // (a prior nobody collected before the next optimization is dropped, like one left in the slot)
  c->inflight_flag = marg_flag;
  c->publish = early && adaptive && count == 1 && c->h_mail && c->info[0].resident && c->info[0].N <= MAIL_MAX_LM;
  struct PublishOff {
    lfvio_ctx *c;
    ~PublishOff() { c->publish = false; }
  } publish_off{c};
  // every slot carries its own max_iter on the device; the pass count follows the largest, the wall-clock cap
  // (synchronous driver only) the smallest positive one
  int max_iter = 0;
  double max_seconds = -1.0;
  for (int s = 0; s < count; s++) {
    if (!c->info[s].resident) {
      c->err = "slot not uploaded";
      return LFVIO_ERR_ARG;
    }
    max_iter = std::max(max_iter, c->info[s].max_iter);
    const double t = c->info[s].max_seconds;
    if (t > 0.0 && (max_seconds <= 0.0 || t < max_seconds)) max_seconds = t;
  }
  if (c->info[0].in_prior_device && (count != 1 || !adaptive || !c->use_graph)) {
    // (k_prior_chain's verdict on the prior travels with the graph of ONE window's synchronous call: include/lfvio.h)
    c->err = "a window uploaded with lfvio_batch_upload_chained_device is optimized by lfvio_batch_optimize_begin or lfvio_batch_optimize(ctx, 1, flag)";
    return LFVIO_ERR_ARG;
  }
  const bool fuse = adaptive && c->use_graph;
  bool tail_done = false;
  int rc = enqueue_solve(c, count, max_iter, adaptive, fuse ? marg_flag : -1, &tail_done, max_seconds, early);
  if (rc) return rc;
  if (fuse) {
    if (tail_done) return LFVIO_OK;  // the usual case: everything ran in the graph of the first chunk
    // some window needed more passes: gauge fix + marginalization for the slots that have not had theirs (gated)
    hipGraphExec_t &tail_graph = c->tail[c->publish ? 1 : 0][marg_flag];
    if (!tail_graph) {
      hipGraph_t graph;
      HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
      CaptureGuard guard(c->stream);
      hipLaunchKernelGGL(k_force_done, dim3((count + 63) / 64), dim3(64), 0, c->stream, c->d_base, c->L.total, count);
      hipLaunchKernelGGL(k_gauge, dim3(1 + (grid_for(c, count).lm + 1) / 2, count), dim3(128), 0, c->stream, c->d_base, c->L.total, 1);
      if (c->publish) hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, c->stream, c->d_base, c->L.total);
      rc = enqueue_marg(c, count, marg_flag, false, true);
      HIPCHK(c, guard.end(&graph));
      if (rc) {
        (void)hipGraphDestroy(graph);
        return rc;
      }
      HIPCHK(c, hipGraphInstantiate(&tail_graph, graph, nullptr, nullptr, 0));
      HIPCHK(c, hipGraphDestroy(graph));
    }
    const bool watch = c->publish;
    if (watch) __atomic_store_n((int *)c->h_mail + 1, 0, __ATOMIC_RELAXED), __atomic_store_n((int *)c->h_mail, 0, __ATOMIC_RELEASE);
    HIPCHK(c, hipGraphLaunch(tail_graph, c->stream));
    if (watch && wait_early(c)) c->inflight = true, c->inflight_first = false;
    return LFVIO_OK;
  }
  hipLaunchKernelGGL(k_gauge, dim3(1 + (grid_for(c, count).lm + 1) / 2, count), dim3(128), 0, c->stream, c->d_base, c->L.total, 0);
  return enqueue_marg(c, count, marg_flag, false);
}

int lfvio_batch_optimize_async(lfvio_ctx *c, int count, int marg_flag) { return batch_optimize_impl(c, count, marg_flag, false); }

int lfvio_batch_sync(lfvio_ctx *c) {
  if (!c) return LFVIO_ERR_ARG;
  if (int rc = join_inflight(c)) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return wait_side(c);
}

int lfvio_batch_optimize(lfvio_ctx *c, int count, int marg_flag) {
  int rc = batch_optimize_impl(c, count, marg_flag, true);
  if (rc) return rc;
  return lfvio_batch_sync(c);
}

int lfvio_batch_download(lfvio_ctx *c, int slot, LfvioSolution *sol, LfvioPrior *prior) {
  if (!c || slot < 0 || slot >= c->batch) return LFVIO_ERR_ARG;
  (void)hipSetDevice(c->device);
  return download(c, slot, sol, prior);
}

// The split form of lfvio_batch_optimize(ctx, 1, marg_flag) + lfvio_batch_download(ctx, 0, sol, prior) for ONE resident window:
// begin() returns with the solution as soon as solve + gauge fix are out — the gated gauge fix pushes the state into host
// memory (Slot::mail) and the host polls that word, no copy and no stream synchronization — while the marginalization of the
// same graph is still running; finish() waits for it and fetches the prior.  (estimator.cpp: the pose is published after
// optimization(), the prior is first read by the next optimization(): the ~0.19 ms of the marginalization overlap with
// whatever the caller does in between.)  Windows without a mailbox (more than MAIL_MAX_LM landmarks, no mapped host memory)
// and windows that were not done within the first graph take the synchronous route inside begin().
int lfvio_batch_optimize_begin(lfvio_ctx *c, int marg_flag, LfvioSolution *sol) {
  if (!c || !sol || c->batch < 1) return LFVIO_ERR_ARG;
  int rc = batch_optimize_impl(c, 1, marg_flag, true, true);
  if (rc) return rc;
  if (!c->inflight) return download(c, 0, sol, nullptr);  // (synchronizes)
  Fetched f;
  char *m = c->h_mail;
  // the loop is closed: its pass and iteration counts size the next call's first graph (a caller that pipelines — the next window
  // uploaded behind this call's marginalization — never joins this graph)
  c->last_passes = std::max(((const int *)m)[4], 1), c->last_iters = ((const TRState *)(m + MAIL_TR))->iteration;
  if (c->inflight_first) predict(c), c->predicted_early = true;
  if (((const int *)m)[5]) {
    c->err = CHAIN_ERR_TEXT;
    c->chain_err_told = true;
    return LFVIO_ERR_DEVICE;
  }
  f.xs = (const FrameState *)(m + MAIL_X), f.tr = (const TRState *)(m + MAIL_TR);
  f.lam[0] = (const double *)(m + MAIL_LAM), f.lam[1] = (const double *)(m + MAIL_LAM + MAIL_LAM_STRIDE), f.prior = nullptr;
  if ((rc = check_solution(c, 0, f))) return rc;
  unpack_solution(c, 0, f, sol);
  return LFVIO_OK;
}

int lfvio_batch_optimize_finish(lfvio_ctx *c, LfvioPrior *prior) {
  if (!c) return LFVIO_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (c->has_held) {
    c->has_held = false;
    if (prior) copy_prior(prior, c->held.get());
    return LFVIO_OK;
  }
  if (c->inflight && prior) {
    // the marginalization ends by pushing its prior into the mailbox (publish_prior): wait for that word, not for the stream
    int *flag = (int *)c->h_mail + 1;
    const int want = c->info[0].mail_seq;
    bool there = false;
    for (;;) {
      if ((there = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == want)) break;
      if (hipStreamQuery(c->stream) != hipErrorNotReady && !(c->side_launched && (hipStreamQuery(c->sstream[0]) == hipErrorNotReady || hipStreamQuery(c->sstream[1]) == hipErrorNotReady))) {
        there = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == want;  // (a worker of kernels_spec.h may still be delivering when stream 0 is idle)
        break;
      }
    }
    if (there) {
      c->inflight = false, c->unsynced = true;  // (what is left of the graph is a copy of two words: the next join waits for it)
      c->last_passes = std::max(((const int *)c->h_mail)[2], 1), c->last_iters = ((const int *)c->h_mail)[3];
      if (c->inflight_first && !c->predicted_early) predict(c);
      c->predicted_early = false;
      Fetched f{};
      f.prior = (LfvioPrior *)(c->h_mail + MAIL_PRIOR);
      bool pass = false;
      if (int rc = check_prior(c, 0, f, &pass)) return rc;
      if (pass)
        if (int rc = fetch_device_prior(c, 0, nullptr)) return rc;
      unpack_prior(c, 0, f, pass, prior);
      return LFVIO_OK;
    }
  }
  int rc = join_inflight(c);
  if (rc) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return prior ? download(c, 0, nullptr, prior) : LFVIO_OK;
}

int lfvio_batch_optimize_pending(const lfvio_ctx *c) { return c && (c->inflight || c->has_held) ? 1 : 0; }

// ---- the landmark-parallel steps either side of optimization()
static int feat_reserve(lfvio_ctx *c, size_t bytes) {
  if (bytes <= c->feat_bytes) return LFVIO_OK;
  if (c->d_feat) (void)hipFree(c->d_feat);
  if (c->h_feat) (void)hipHostFree(c->h_feat);
  c->d_feat = nullptr, c->h_feat = nullptr, c->feat_bytes = 0;
  bytes = align_up(bytes + bytes / 4, 4096);
  if (hipMalloc(&c->d_feat, bytes) != hipSuccess || hipHostMalloc((void **)&c->h_feat, bytes, hipHostMallocDefault) != hipSuccess) {
    c->err = "out of memory (feature scratch)";
    return LFVIO_ERR_DEVICE;
  }
  c->feat_bytes = bytes;
  return LFVIO_OK;
}

int lfvio_triangulate(lfvio_ctx *c, const LfvioTriangulateIn *in, double *estimated_depth) {
  if (!c || !in || in->num_landmarks < 0 || in->num_observations < 0) return LFVIO_ERR_ARG;
  const int N = in->num_landmarks, M = in->num_observations;
  if (N == 0) return LFVIO_OK;
  if (!estimated_depth || !in->start_frame || !in->obs_offset || !in->obs_point || in->obs_offset[0] != 0 || in->obs_offset[N] != M) {
    c->err = "triangulate: null arrays or obs_offset is not a CSR over num_observations";
    return LFVIO_ERR_ARG;
  }
  for (int l = 0; l < N; l++) {
    const int k = in->obs_offset[l + 1] - in->obs_offset[l], s = in->start_frame[l];
    if (k < 2 || s < 0 || s + k > LFVIO_NUM_FRAMES) {
      c->err = "triangulate: landmark with fewer than 2 observations or a track leaving the window";
      return LFVIO_ERR_ARG;
    }
  }
  (void)hipSetDevice(c->device);
  hipStream_t fs = c->fstream ? c->fstream : c->stream;  // not behind the tail of an optimization still in flight
  const size_t oF = 0, oS = align_up(sizeof(FeatFrames), 256), oO = align_up(oS + (size_t)N * 4, 256),
               oP = align_up(oO + (size_t)(N + 1) * 4, 256), oD = align_up(oP + (size_t)M * 24, 256), total = oD + (size_t)N * 8;
  int rc = feat_reserve(c, total);
  if (rc) return rc;
  FeatFrames F;
  std::memcpy(F.Ps, in->Ps, sizeof F.Ps), std::memcpy(F.Rs, in->Rs, sizeof F.Rs);
  std::memcpy(F.tic, in->tic, sizeof F.tic), std::memcpy(F.ric, in->ric, sizeof F.ric);
  F.init_depth = in->init_depth;
  char *d = c->d_feat, *h = c->h_feat;  // (a copy from pageable memory costs ~20 us each on this runtime: one packed pinned block instead of five)
  std::memcpy(h + oF, &F, sizeof F);
  std::memcpy(h + oS, in->start_frame, (size_t)N * 4);
  std::memcpy(h + oO, in->obs_offset, (size_t)(N + 1) * 4);
  std::memcpy(h + oP, in->obs_point, (size_t)M * 24);
  std::memcpy(h + oD, estimated_depth, (size_t)N * 8);
  HIPCHK(c, hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, fs));
  hipLaunchKernelGGL(k_triangulate, dim3((N + TRI_THREADS - 1) / TRI_THREADS), dim3(TRI_THREADS), 0, fs, (const FeatFrames *)(d + oF), N,
                     (const int *)(d + oS), (const int *)(d + oO), (const double *)(d + oP), (double *)(d + oD));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(h + oD, d + oD, (size_t)N * 8, hipMemcpyDeviceToHost, fs));
  HIPCHK(c, hipStreamSynchronize(fs));
  std::memcpy(estimated_depth, h + oD, (size_t)N * 8);
  return LFVIO_OK;
}

int lfvio_shift_depth(lfvio_ctx *c, int n, const double *uv_i, const double marg_R[9], const double marg_P[3], const double new_R[9],
                      const double new_P[3], double init_depth, double *estimated_depth) {
  if (!c || n < 0) return LFVIO_ERR_ARG;
  if (n == 0) return LFVIO_OK;
  if (!uv_i || !marg_R || !marg_P || !new_R || !new_P || !estimated_depth) return LFVIO_ERR_ARG;
  (void)hipSetDevice(c->device);
  hipStream_t fs = c->fstream ? c->fstream : c->stream;  // not behind the tail of an optimization still in flight
  const size_t oT = 0, oU = 256, oD = align_up(oU + (size_t)n * 24, 256), total = oD + (size_t)n * 8;
  int rc = feat_reserve(c, total);
  if (rc) return rc;
  double T[25];
  std::memcpy(T, marg_R, 72), std::memcpy(T + 9, marg_P, 24), std::memcpy(T + 12, new_R, 72), std::memcpy(T + 21, new_P, 24);
  T[24] = init_depth;
  char *d = c->d_feat, *h = c->h_feat;
  std::memcpy(h + oT, T, sizeof T);
  std::memcpy(h + oU, uv_i, (size_t)n * 24);
  std::memcpy(h + oD, estimated_depth, (size_t)n * 8);
  HIPCHK(c, hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, fs));
  hipLaunchKernelGGL(k_shift_depth, dim3((n + 255) / 256), dim3(256), 0, fs, n, (const double *)(d + oU), (const double *)(d + oT),
                     (double *)(d + oD));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(h + oD, d + oD, (size_t)n * 8, hipMemcpyDeviceToHost, fs));
  HIPCHK(c, hipStreamSynchronize(fs));
  std::memcpy(estimated_depth, h + oD, (size_t)n * 8);
  return LFVIO_OK;
}

int lfvio_preintegrate(lfvio_ctx *c, int num_intervals, const LfvioImuInterval *in, const double noise[4], LfvioPreintegration *out) {
  if (!c || num_intervals < 0) return LFVIO_ERR_ARG;
  if (num_intervals == 0) return LFVIO_OK;
  if (!in || !noise || !out) return LFVIO_ERR_ARG;
  size_t S = 0;
  for (int k = 0; k < num_intervals; k++) {
    if (in[k].num_samples < 0 || (in[k].num_samples > 0 && (!in[k].dt || !in[k].acc || !in[k].gyr))) {
      c->err = "preintegrate: interval with a negative sample count or null sample arrays";
      return LFVIO_ERR_ARG;
    }
    S += (size_t)in[k].num_samples;
  }
  (void)hipSetDevice(c->device);
  hipStream_t fs = c->fstream ? c->fstream : c->stream;  // not behind the tail of an optimization still in flight
  const size_t K = (size_t)num_intervals;
  const size_t oJ = 0, oN = align_up(K * sizeof(ImuJob), 256), oT = oN + 256, oA = align_up(oT + S * 8, 256), oG = align_up(oA + S * 24, 256),
               oO = align_up(oG + S * 24, 256), total = oO + K * sizeof(LfvioPreintegration);
  int rc = feat_reserve(c, total);
  if (rc) return rc;
  // one packed (pinned) staging block -> one host-to-device copy
  char *h = c->h_feat;
  size_t off = 0;
  for (int k = 0; k < num_intervals; k++) {
    ImuJob *jb = (ImuJob *)(h + oJ) + k;
    const size_t n = (size_t)in[k].num_samples;
    jb->n = (int)n, jb->off = (int)off;
    std::memcpy(jb->acc_0, in[k].acc_0, 24), std::memcpy(jb->gyr_0, in[k].gyr_0, 24);
    std::memcpy(jb->ba, in[k].linearized_ba, 24), std::memcpy(jb->bg, in[k].linearized_bg, 24);
    if (n) {
      std::memcpy(h + oT + off * 8, in[k].dt, n * 8);
      std::memcpy(h + oA + off * 24, in[k].acc, n * 24);
      std::memcpy(h + oG + off * 24, in[k].gyr, n * 24);
    }
    off += n;
  }
  std::memcpy(h + oN, noise, 32);
  char *d = c->d_feat;
  HIPCHK(c, hipMemcpyAsync(d, h, oO, hipMemcpyHostToDevice, fs));
  hipLaunchKernelGGL(k_preintegrate, dim3(num_intervals), dim3(PRE_THREADS), 0, fs, (const ImuJob *)(d + oJ), (const double *)(d + oT),
                     (const double *)(d + oA), (const double *)(d + oG), (const double *)(d + oN), (LfvioPreintegration *)(d + oO));
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(h + oO, d + oO, K * sizeof(LfvioPreintegration), hipMemcpyDeviceToHost, fs));
  HIPCHK(c, hipStreamSynchronize(fs));
  std::memcpy(out, h + oO, K * sizeof(LfvioPreintegration));
  return LFVIO_OK;
}

// ---- debug / parity hooks (include/lfvio_debug.h)
int lfvio_debug_linearize(lfvio_ctx *c, const LfvioWindow *in, double *Hpp, double *gp, double *a, double *b, double *W,
                          double *cost) {
  if (!c || !in) return LFVIO_ERR_ARG;
  (void)hipSetDevice(c->device);
  int rc = reserve(c, 1, in->num_landmarks, in->num_observations);
  if (rc) return rc;
  if ((rc = upload_window(c, 0, in))) return rc;
  const Grid g = grid_for(c, 1);
  { const SetupLaunch sl_ = setup_launch(c, 1, grid_for(c, 1).lm, false); hipLaunchKernelGGL(k_setup, dim3(sl_.gx, 1), dim3(256), 0, c->stream, c->d_base, c->L.total, MODE_SOLVE, sl_.bits); }
  launch_lin(c, 1, g, MODE_SOLVE);
  launch_sum(c, 1, g, MODE_SOLVE);
  launch_solve(c, 1);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const Layout &L = c->L;
  char *d = c->d_base;
  const int N = in->num_landmarks;
  std::vector<int> woff(N + 1, 0), lst(N), lcn(N);
  if (N) {
    HIPCHK(c, hipMemcpy(woff.data(), d + L.lm_woff, sizeof(int) * (N + 1), hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(lst.data(), d + L.lm_start, sizeof(int) * N, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(lcn.data(), d + L.lm_cnt, sizeof(int) * N, hipMemcpyDeviceToHost));
  }
  std::vector<double> packed(PACKED), av(N), bv(N), Wv((size_t)woff[N] + 1);
  HIPCHK(c, hipMemcpy(packed.data(), d + L.xch + sizeof(double) * XOFF_H, sizeof(double) * PACKED, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(gp, d + L.xch + sizeof(double) * XOFF_G, sizeof(double) * KP, hipMemcpyDeviceToHost));
  if (N) {
    HIPCHK(c, hipMemcpy(av.data(), d + L.a, sizeof(double) * N, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(bv.data(), d + L.b, sizeof(double) * N, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(Wv.data(), d + L.W, sizeof(double) * woff[N], hipMemcpyDeviceToHost));
  }
  TRState tr;
  HIPCHK(c, hipMemcpy(&tr, d + offsetof(Slot, tr), sizeof tr, hipMemcpyDeviceToHost));
  for (int i = 0; i < KP; i++)
    for (int j = 0; j < KP; j++) Hpp[i * KP + j] = packed[pidx(i, j)];
  const std::vector<int> &perm = c->info[0].perm;
  for (int dl = 0; dl < N; dl++) {
    a[perm[dl]] = av[dl], b[perm[dl]] = bv[dl];
    for (int k = 0; k < KC; k++) W[(size_t)perm[dl] * KC + k] = 0.0;
    for (int ci = 0; ci < 6 * lcn[dl] + 7; ci++)  // stored span: the track's frames, then ex (6) and td
      W[(size_t)perm[dl] * KC + (ci < 6 * lcn[dl] ? 6 * lst[dl] + ci : 66 + (ci - 6 * lcn[dl]))] = Wv[(size_t)woff[dl] + ci];
  }
  *cost = tr.x_cost;
  return LFVIO_OK;
}

// The mu-retry path (do_schur without do_lin: k_lin's landmark blocks redo only the Schur SYRK from the stored W rows)
// against a full linearization at the same mu.  Returns the largest absolute difference of the Schur sums (expected 0).
int lfvio_debug_schur_repeat(lfvio_ctx *c, const LfvioWindow *in, double mu, double *max_abs_diff) {
  if (!c || !in || !max_abs_diff) return LFVIO_ERR_ARG;
  (void)hipSetDevice(c->device);
  int rc = reserve(c, 1, in->num_landmarks, in->num_observations);
  if (rc) return rc;
  if ((rc = upload_window(c, 0, in))) return rc;
  const Grid g = grid_for(c, 1);
  const bool lw = use_linw(c, 1, g, MODE_SOLVE);  // (lfvio_debug_configure "linw" 2: the window-resident sweep, for one window)
  const bool lb = !lw && use_linb(c, 1, g, MODE_SOLVE);
  char *d = c->d_base;
  const size_t o_tr = offsetof(Slot, tr);
  auto poke_int = [&](size_t off, int v) { return hipMemcpy(d + o_tr + off, &v, sizeof v, hipMemcpyHostToDevice); };
  auto run = [&](int do_lin, std::vector<double> &out) -> int {
    HIPCHK(c, hipMemcpy(d + o_tr + offsetof(TRState, mu), &mu, sizeof mu, hipMemcpyHostToDevice));
    HIPCHK(c, poke_int(offsetof(TRState, do_lin), do_lin));
    HIPCHK(c, poke_int(offsetof(TRState, do_schur), 1));
    if (lw) {
      launch_linw(c, 1);
    } else if (lb) {
      launch_linb(c, 1);
    } else {
      launch_lin(c, 1, g, MODE_SOLVE);
      launch_sum(c, 1, g, MODE_SOLVE);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    out.resize(SCHUR_LEN);
    HIPCHK(c, hipMemcpy(out.data(), d + c->L.xch + sizeof(double) * XOFF_S, sizeof(double) * SCHUR_LEN, hipMemcpyDeviceToHost));
    return LFVIO_OK;
  };
  { const SetupLaunch sl_ = setup_launch(c, 1, g.lm, lw); hipLaunchKernelGGL(k_setup, dim3(sl_.gx, 1), dim3(256), 0, c->stream, c->d_base, c->L.total, MODE_SOLVE, sl_.bits); }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<double> first, repeat, full;
  const double mu1 = mu;
  mu = 1e-8;
  if ((rc = run(1, first))) return rc;   // ordinary first pass (fixes the Jacobi scaling: k_solve does that, so run it)
  launch_solve(c, 1, lw);
  HIPCHK(c, hipStreamSynchronize(c->stream));
  mu = mu1;
  if ((rc = run(0, repeat))) return rc;  // Schur only, new mu
  if ((rc = run(1, full))) return rc;    // everything again at the new mu
  double m = 0, ref = 0;
  for (int i = 0; i < SCHUR_LEN; i++) m = std::max(m, std::fabs(repeat[i] - full[i])), ref = std::max(ref, std::fabs(full[i] - first[i]));
  *max_abs_diff = m;
  return ref > 0 ? LFVIO_OK : LFVIO_ERR_ARG;  // the new mu must have changed the sums, or the check proves nothing
}

// Post-Schur system A', b' of the last marginalization of slot 0 (n x n, n).
int lfvio_debug_marg_system(lfvio_ctx *c, int n, double *A, double *b) {
  if (!c || !c->d_base) return LFVIO_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (int rc = join_inflight(c)) return rc;
  char *d = c->d_base + c->L.mscr;
  HIPCHK(c, hipMemcpy(A, d + sizeof(double) * (92 * 92 + 96), sizeof(double) * n * n, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(b, d + sizeof(double) * (92 * 92 + 96 + n * n), sizeof(double) * n, hipMemcpyDeviceToHost));
  return LFVIO_OK;
}

int lfvio_debug_read_clocks(lfvio_ctx *c, long long *out32) {
  if (!c || !c->d_base) return LFVIO_ERR_ARG;
  if (int rc = join_inflight(c)) return rc;
  HIPCHK(c, hipMemcpy(out32, c->d_base + offsetof(Slot, dbg), sizeof(long long) * 32, hipMemcpyDeviceToHost));
  HIPCHK(c, hipMemcpy(out32 + 32, c->d_base + offsetof(Slot, jtrace), sizeof(double) * 32, hipMemcpyDeviceToHost));
  return LFVIO_OK;
}

// Average duration (ms) of `reps` launches of one pipeline kernel on slots [0, count), measured with
// HIP events on the context's own stream.  which: 0 k_lin (with the Schur SYRK of the landmark blocks), 2 k_sum, 3 k_solve.
// The slots must hold an uploaded window; the trust-region flags are re-armed by k_setup first.
int lfvio_debug_time_kernel(lfvio_ctx *c, int which, int count, int reps, double *avg_ms) {
  if (!c || !c->d_base || count <= 0 || count > c->batch || reps <= 0) return LFVIO_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (int rc = join_inflight(c)) return rc;
  const Grid g = grid_for(c, count);
  const size_t st = c->L.total;
  const bool lw = use_linw(c, count, g, MODE_SOLVE);
  const bool offs = (slots_offs(c, count) & 2) != 0;  // the instantiation the passes behind the first one run (launch_lin)
  // (a rank of a sharded window sweeps its share group by group under the same condition: shard.inc)
  const bool lb = !lw && (c->shard_active ? (c->linw_mode != 0 && count == 1 && c->info[0].linb_ok) : use_linb(c, count, g, MODE_SOLVE));
  // which kernel the launch would take is settled before anything is created or enqueued
  if ((which == 11 || which == 12) && !lw) {
    c->err = "the resident windows are not linearized by k_linw";
    return LFVIO_ERR_ARG;
  }
  if (which >= 15 && which <= 17 && !lb) {
    c->err = "the resident window is not linearized by k_linb";
    return LFVIO_ERR_ARG;
  }
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  if (hipError_t e = hipEventCreate(&e1); e != hipSuccess) {
    (void)hipEventDestroy(e0);
    HIPCHK(c, e);
  }
  { const SetupLaunch sl_ = setup_launch(c, count, grid_for(c, count).lm, lw); hipLaunchKernelGGL(k_setup, dim3(sl_.gx, count), dim3(256), 0, c->stream, c->d_base, st, MODE_SOLVE, sl_.bits); }
  // one full linearization so that every kernel has valid inputs
  if (lw) {
    launch_linw(c, count, MODE_SOLVE, offs);
  } else if (lb && which >= 15 && which <= 17) {
    launch_linb(c, count, offs);
  } else {
    launch_lin(c, count, g, MODE_SOLVE, offs);
    launch_sum(c, count, g, MODE_SOLVE);
  }
  if (which == 14) launch_solve(c, count, lw);
  if (which == 17) launch_solve(c, count, false);

  HIPCHK(c, hipEventRecord(e0, c->stream));
  for (int r = 0; r < reps; r++) {
    switch (which) {
      case 0: launch_lin(c, count, g, MODE_SOLVE, offs); break;
      case 2: launch_sum(c, count, g, MODE_SOLVE); break;  // k_presum + k_sum for large windows
      case 8: case 9: case 10: case 18: {  // k_lin by role: landmark blocks | Gram chunks | IMU factors + prior | IMU factors alone
        const int gram_wgs = (g.ch + 3) / 4;
        const int gx = which == 8 ? g.lw : which == 9 ? gram_wgs : which == 18 ? LFVIO_WINDOW_SIZE : LFVIO_WINDOW_SIZE + 1;
        hipLaunchKernelGGL((k_lin<LIN_ROLE_ALL, false>), dim3(gx, count), dim3(LIN_THREADS), 0, c->stream, c->d_base, st, MODE_SOLVE, which == 8 ? g.lw : 0,
                           which == 9 ? gram_wgs : 0);
      } break;
      case 4: case 5: case 6: case 7: {  // k_setup by role: state + table | + IMU sqrt_info | + prior J0^T J0 | + inverse depths
        const SetupLaunch sl = setup_launch(c, count, g.lm, false);  // (the grid the product launches: compact for a resident batch)
        const int wgs = (sl.bits & 4) ? 3 : SETUP_WGS, gx = which == 4 ? 1 : which == 5 ? 2 : which == 6 ? wgs : sl.gx;
        hipLaunchKernelGGL(k_setup, dim3(gx, count), dim3(256), 0, c->stream, c->d_base, st, MODE_SOLVE, sl.bits);
      } break;
      case 11: case 12: launch_linw(c, count, MODE_SOLVE, offs); break;  // the window-resident sweep of a batch (k_linw: pose-side factors, visual sweep, Schur)
      case 13: launch_solve(c, count, use_linw(c, count, g, MODE_SOLVE)); break;
      case 14: hipLaunchKernelGGL(k_stepw, dim3(1, count), dim3(STEPW_LAUNCH_THREADS), 0, c->stream, c->d_base, st); break;  // (not idempotent: a few reps only)
      // a large single window, group by group: 15 the strip sweep (k_linb), 16 the sum of its partials (k_sumb), 17 the landmark
      // back-substitution from the transposed rows (k_backsub_wt)
      case 15: hipLaunchKernelGGL(k_linb<false>, dim3(linb_grid(c, count), count), dim3(LW_THREADS), LW_LDS_BYTES, c->stream, c->d_base, st, linw_args(c)); break;
      case 16: hipLaunchKernelGGL(k_sumb, dim3(LINB_SUM_GRID, count), dim3(LINB_SUM_THREADS), 0, c->stream, c->d_base, st, linw_args(c)); break;
      case 17: hipLaunchKernelGGL(k_backsub_wt, dim3(g.lm, count), dim3(64), 0, c->stream, c->d_base, st, c->L.capLmBlocks * LM_BLOCK); break;
      default: launch_solve(c, count); break;
    }
  }
  HIPCHK(c, hipEventRecord(e1, c->stream));
  HIPCHK(c, hipEventSynchronize(e1));
  float ms = 0;
  HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
  *avg_ms = (double)ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return LFVIO_OK;
}

// Which kernel linearizes a launch over the resident slots [0, count): 0 k_lin (+ k_sum), 1 k_linw, 2 k_linb (+ k_sumb).
static int sweep_kernel_of(lfvio_ctx *c, int count) {
  if (!c || !c->d_base || count <= 0 || count > c->batch) return LFVIO_ERR_ARG;
  const Grid g = grid_for(c, count);
  if (use_linw(c, count, g, MODE_SOLVE)) return 1;
  const bool lb = c->shard_active ? (c->linw_mode != 0 && count == 1 && c->info[0].linb_ok) : use_linb(c, count, g, MODE_SOLVE);
  return lb ? 2 : 0;
}

// One linearization + dense solve of the resident slots [0, count) from their uploaded state, by whichever path the launch
// takes (k_linw or k_lin roles + k_sum: lfvio_debug_configure "linw"), then what the pass left in slot `slot`:
// gp[172], schur[15 * 256] (tile layout), lm_sum[5], a[N], b[N], gn_p[172] (pose-side Gauss-Newton step), q[16] (the quadratic
// forms of the dogleg model), x_cost.  tests/test_linw.py holds the two paths against each other with it.
int lfvio_debug_resident_pass(lfvio_ctx *c, int count, int slot, double *gp, double *schur, double *lm_sum, double *a, double *b, double *gn_p, double *q,
                              double *x_cost) {
  if (!c || !c->d_base || count <= 0 || count > c->batch || slot < 0 || slot >= count) return LFVIO_ERR_ARG;
  (void)hipSetDevice(c->device);
  if (int rc = join_inflight(c)) return rc;
  for (int s = 0; s < count; s++)
    if (!c->info[s].resident) return LFVIO_ERR_ARG;
  const Grid g = grid_for(c, count);
  const size_t st = c->L.total;
  const bool lw = use_linw(c, count, g, MODE_SOLVE), lb = !lw && use_linb(c, count, g, MODE_SOLVE);
  { const SetupLaunch sl_ = setup_launch(c, count, g.lm, lw); hipLaunchKernelGGL(k_setup, dim3(sl_.gx, count), dim3(256), 0, c->stream, c->d_base, st, MODE_SOLVE, sl_.bits); }
  if (lw) {
    launch_linw(c, count);
  } else if (lb) {
    launch_linb(c, count);
  } else {
    launch_lin(c, count, g, MODE_SOLVE);
    launch_sum(c, count, g, MODE_SOLVE);
  }
  launch_solve(c, count, lw);
  HIPCHK(c, hipGetLastError());  // (a launch that was refused — resources, LDS — must not pass as a result left by an earlier one)
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const char *d = c->d_base + (size_t)slot * st;
  const int N = c->info[slot].N;
  auto get = [&](double *dst, size_t off, size_t n) { return !dst || !n ? hipSuccess : hipMemcpy(dst, d + off, n * 8, hipMemcpyDeviceToHost); };
  HIPCHK(c, get(gp, c->L.xch + (size_t)XOFF_G * 8, KP));
  HIPCHK(c, get(schur, c->L.xch + (size_t)XOFF_S * 8, SCHUR_LEN));
  HIPCHK(c, get(lm_sum, offsetof(Slot, lm_sum), 5));
  HIPCHK(c, get(a, c->L.a, N));
  HIPCHK(c, get(b, c->L.b, N));
  HIPCHK(c, get(gn_p, offsetof(Slot, gn_p), KP));
  HIPCHK(c, get(q, offsetof(Slot, tr) + offsetof(TRState, q), Q_COUNT));
  HIPCHK(c, get(x_cost, offsetof(Slot, tr) + offsetof(TRState, x_cost), 1));
  return lw ? 1 : lb ? 2 : 0;
}

// Every switch of the debug interface in one call (include/lfvio_debug.h).  The product entry points read no environment: a tool
// that wants LFVIO_DEBUG="key=value,key=value" honoured asks for it with the key "env".
int lfvio_debug_configure(lfvio_ctx *c, const char *key, double value) {
  if (!c || !key) return LFVIO_ERR_ARG;
  const std::string k = key;
  const int iv = (int)value;
  if (k == "env") {
    const char *e = getenv("LFVIO_DEBUG");
    if (!e) return LFVIO_OK;
    std::string all = e;
    for (size_t p = 0; p < all.size();) {
      const size_t q = std::min(all.find(',', p), all.size()), eq = all.find('=', p);
      if (eq != std::string::npos && eq < q) {
        const int rc = lfvio_debug_configure(c, all.substr(p, eq - p).c_str(), atof(all.substr(eq + 1, q - eq - 1).c_str()));
        if (rc) return rc;
      }
      p = q + 1;
    }
    return LFVIO_OK;
  }
  // (switches that change what the captured graphs hold or what an upload builds: the call in flight is joined, the graphs go)
  const bool structural = k == "graph" || k == "linw" || k == "force_eig";
  if (structural) {
    if (int rc = join_inflight(c)) return rc;
  }
  if (k == "graph") c->use_graph = iv != 0;
  else if (k == "first_passes") {
    if (iv < 0) return LFVIO_ERR_ARG;
    c->fixed_passes = iv;
  } else if (k == "spec_count") c->fixed_spec = iv > 0, c->spec_count = iv > 0 ? std::max(1, std::min(1 + SPEC_EXTRA, iv)) : 3;
  else if (k == "function_tolerance") {
    if (!(value >= 0.0)) return LFVIO_ERR_ARG;
    c->fn_tol = value;
  } else if (k == "initial_radius") c->init_radius = value > 0.0 ? value : 1e4;
  else if (k == "linw") {
    if (iv < 0 || iv > 2) return LFVIO_ERR_ARG;
    c->linw_mode = iv;  // (takes effect with the next upload: the plan and its arrays are built there)
  } else if (k == "lm_half") c->lm_half = iv != 0;  // (with the next upload)
  else if (k == "force_eig") c->force_eig = iv != 0;  // (a kernel argument of the captured launches)
  else if (k == "marg_ahead") c->marg_ahead = iv != 0;  // (a flag of the upload: Slot::spec_on)
  else if (k == "break_next_chain") c->debug_break_chain = iv != 0;
  else {
    c->err = "lfvio_debug_configure: unknown key '" + k + "'";
    return LFVIO_ERR_ARG;
  }
  if (structural) destroy_graph(c);
  return LFVIO_OK;
}

int lfvio_debug_query(lfvio_ctx *c, const char *key, double *out, int n) {
  if (!c || !key || !out || n <= 0) return LFVIO_ERR_ARG;
  const std::string k = key;
  auto put = [&](std::initializer_list<double> v) {
    int i = 0;
    for (double x : v)
      if (i < n) out[i++] = x;
    return LFVIO_OK;
  };
  if (k == "last_call") return put({(double)c->last_passes, (double)c->last_iters, (double)c->stat_chunks, (double)c->spec_count});
  if (k == "marg_ahead") return put({(double)c->stat_ahead_calls, (double)c->stat_ahead_hits});
  if (k == "upload_times") return put({c->up_us[0], c->up_us[1], c->up_us[2], c->up_us[3]});
  if (k == "sweep_kernel") {  // out[0] in: the number of resident slots the launch would cover
    const int r = sweep_kernel_of(c, (int)out[0]);
    if (r < 0) return r;
    return put({(double)r});
  }
  c->err = "lfvio_debug_query: unknown key '" + k + "'";
  return LFVIO_ERR_ARG;
}

// ---- landmark-sharded API: declared in lfvio.h, implemented in shard.inc
#include "shard.inc"
// ---- multi-GPU groups (RCCL): declared in lfvio.h, implemented in group.inc
#include "group.inc"

}  // extern "C"
