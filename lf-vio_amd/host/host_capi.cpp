// host_capi.cpp — extern "C" driver of the host side (window_estimator.h), so that the Python tests and tools can
// exercise it the way estimator_node.cpp drives the reference's Estimator.  Frames are addressed by their LOGICAL
// index 0..10 (oldest..newest); the ring underneath is not visible here.
#include <cstring>

#include "replay.h"
#include "window_estimator.h"

using namespace lfvio;

namespace {
inline WindowEstimator *E(void *h) { return (WindowEstimator *)h; }
inline Vector3d v3(const double *a) { return Vector3d(a[0], a[1], a[2]); }
inline void setM(Matrix3d &m, const double *a) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m(i, j) = a[i * 3 + j];
}
inline void getM(const Matrix3d &m, double *a) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) a[i * 3 + j] = m(i, j);
}
}  // namespace

extern "C" {

void *lfvio_host_create(void) { return new WindowEstimator(); }
void lfvio_host_destroy(void *h) { delete E(h); }

// the YAML / parameters.cpp values; p = {ACC_N, GYR_N, ACC_W, GYR_W, g_norm, TR, ROW, SOLVER_TIME, TD}
void lfvio_host_set_params(const double *p, int estimate_extrinsic, int estimate_td, int num_iterations) {
  Config &c = config();
  c.acc_n = p[0], c.gyr_n = p[1], c.acc_w = p[2], c.gyr_w = p[3];
  c.gravity[0] = 0, c.gravity[1] = 0, c.gravity[2] = p[4];
  c.tr = p[5], c.row = p[6], c.solver_time = p[7], c.td = p[8];
  c.estimate_extrinsic = estimate_extrinsic, c.estimate_td = estimate_td, c.num_iterations = num_iterations;
}
// the configured extrinsic (TIC / RIC of the YAML file): what a reset restores
void lfvio_host_set_extrinsic(const double *tic, const double *ric) {
  std::memcpy(config().tic, tic, sizeof config().tic);
  std::memcpy(config().ric, ric, sizeof config().ric);
}
// SOLVER_TIME on its own (<= 0: no wall-clock cap — what a test on a loaded machine wants: Ceres' max_solver_time_in_seconds makes the
// number of iterations depend on the clock)
void lfvio_host_set_solver_time(double seconds) { config().solver_time = seconds; }
void lfvio_host_set_min_parallax(double keyframe_parallax_px) { config().min_parallax = keyframe_parallax_px / FOCAL_LENGTH; }

void lfvio_host_set_state(void *h, const double *Ps, const double *Rs, const double *Vs, const double *Bas, const double *Bgs,
                          const double *tic, const double *ric, double td) {
  WindowEstimator *e = E(h);
  for (int i = 0; i < FRAMES; i++) {
    Keyframe &f = e->kf(i);
    f.P = v3(Ps + 3 * i), f.V = v3(Vs + 3 * i), f.Ba = v3(Bas + 3 * i), f.Bg = v3(Bgs + 3 * i);
    setM(f.R, Rs + 9 * i);
  }
  e->tic = v3(tic);
  setM(e->ric, ric);
  e->td = td;
}

void lfvio_host_get_state(void *h, double *Ps, double *Rs, double *Vs, double *Bas, double *Bgs, double *tic, double *ric, double *td) {
  WindowEstimator *e = E(h);
  for (int i = 0; i < FRAMES; i++) {
    const Keyframe &f = e->kf(i);
    for (int k = 0; k < 3; k++) Ps[3 * i + k] = f.P(k), Vs[3 * i + k] = f.V(k), Bas[3 * i + k] = f.Ba(k), Bgs[3 * i + k] = f.Bg(k);
    getM(f.R, Rs + 9 * i);
  }
  for (int k = 0; k < 3; k++) tic[k] = e->tic(k);
  getM(e->ric, ric);
  *td = e->td;
}

void lfvio_host_clear_features(void *h) { E(h)->tracks.clear(); }

// a whole track: obs n x 8 = [bearing xyz, pixel uv, bearing-velocity xyz] like the 8-vector of estimator_node.cpp:308
void lfvio_host_add_feature(void *h, int id, int start_frame, int n, const double *obs, const double *cur_td, double estimated_depth) {
  TrackTable &t = E(h)->tracks;
  const int s = t.create(id, start_frame);
  for (int k = 0; k < n; k++) t.append(s, obs + 8 * k, cur_td[k]);
  t.setDepth(s, estimated_depth);
}

int lfvio_host_triangulate(void *h) {
  WindowEstimator *e = E(h);
  e->status = LFVIO_OK;
  e->triangulate();
  return e->status;
}
// the depth re-anchoring of a MARGIN_OLD slide with the marginalized frame's pose given (back_R0, back_P0) and the
// estimator's frame 0 as the new anchor frame (estimator.cpp:1120-1127)
int lfvio_host_remove_back_shift_depth(void *h, const double *back_R0, const double *back_P0) {
  WindowEstimator *e = E(h);
  Matrix3d bR;
  setM(bR, back_R0);
  const Vector3d bP = v3(back_P0);
  e->status = LFVIO_OK;
  std::vector<TrackTable::Shifted> moved;
  e->tracks.dropOldestFrame(&moved);
  e->reanchorDepths(bR * e->ric, bP + bR * e->tic, e->kf(0).R * e->ric, e->kf(0).P + e->kf(0).R * e->tic, moved);
  return e->status;
}
void lfvio_host_set_depths(void *h, const double *d, int n) {
  TrackTable &t = E(h)->tracks;
  int k = 0;
  for (int s : t.order())
    if (k < n) t.setDepth(s, d[k++]);
}
int lfvio_host_num_features(void *h) { return E(h)->tracks.live(); }
// (feature_id, start_frame, number of observations, estimated_depth) of every track in table order
void lfvio_host_list_features(void *h, int *ids, int *start, int *count, double *depth) {
  const TrackTable &t = E(h)->tracks;
  int k = 0;
  for (int s : t.order()) ids[k] = t.id(s), start[k] = t.start(s), count[k] = t.count(s), depth[k] = t.depth(s), k++;
}
int lfvio_host_feature_count(void *h) { return E(h)->tracks.solvableCount(); }
void lfvio_host_get_depths(void *h, double *out) {
  const TrackTable &t = E(h)->tracks;
  int k = 0;
  for (int s : t.order()) out[k++] = t.depth(s);
}

// span `frame` (the samples between keyframes frame - 1 and frame) := {acc0, gyr0, ba, bg} + n samples
void lfvio_host_set_imu(void *h, int frame, const double *acc0, const double *gyr0, const double *ba, const double *bg, int n,
                        const double *dt, const double *acc, const double *gyr) {
  ImuSpan &sp = E(h)->span(frame);
  sp.open(v3(acc0), v3(gyr0), v3(ba), v3(bg));
  for (int k = 0; k < n; k++) sp.push(dt[k], acc + 3 * k, gyr + 3 * k);
}

void lfvio_host_repropagate(void *h, int frame, const double *ba, const double *bg) {
  ImuSpan &sp = E(h)->span(frame);
  if (!sp.present) return;
  std::memcpy(sp.lin_ba, ba, sizeof sp.lin_ba), std::memcpy(sp.lin_bg, bg, sizeof sp.lin_bg);
  sp.dirty = true;
}

// ba, bg: [11][3]; every present span is integrated again in one device call; returns the status
int lfvio_host_repropagate_window(void *h, const double *ba, const double *bg) {
  WindowEstimator *e = E(h);
  Vector3d a[FRAMES], g[FRAMES];
  for (int i = 0; i < FRAMES; i++) a[i] = v3(ba + 3 * i), g[i] = v3(bg + 3 * i);
  e->status = LFVIO_OK;
  e->refreshSpans(true, a, g);
  return e->status;
}

void lfvio_host_process_imu(void *h, double dt, const double *acc, const double *gyr) { E(h)->pushImu(dt, acc, gyr); }

// ids[n], pts[n][8] = x y z u v vx vy vz (camera 0); returns the status of the device calls inside
int lfvio_host_process_image(void *h, double stamp, int n, const int *ids, const double *pts) {
  WindowEstimator *e = E(h);
  e->status = LFVIO_OK;
  e->pushImage(stamp, n, ids, pts);
  return e->status;
}

// only the keyframe decision of an image: appends the observations, returns 1 for MARGIN_OLD
int lfvio_host_add_feature_check_parallax(void *h, int frame_count, int n, const int *ids, const double *pts, double td) {
  return E(h)->keyframeTest(frame_count, n, ids, pts, td) ? 1 : 0;
}

// Ps Rs Vs Bas Bgs as lfvio_host_set_state, g[3]
void lfvio_host_set_bootstrap(void *h, const double *Ps, const double *Rs, const double *Vs, const double *Bas, const double *Bgs, const double *g) {
  WindowEstimator::Bootstrap &b = E(h)->bootstrap;
  for (int i = 0; i < FRAMES; i++) {
    b.kf[i].P = v3(Ps + 3 * i), b.kf[i].V = v3(Vs + 3 * i), b.kf[i].Ba = v3(Bas + 3 * i), b.kf[i].Bg = v3(Bgs + 3 * i);
    setM(b.kf[i].R, Rs + 9 * i);
  }
  b.g = v3(g);
  b.valid = true;
}

// the window is already filled from outside (set_state / add_feature / set_imu): continue in the running phase
void lfvio_host_set_running(void *h, const double *stamps, const double *acc_0, const double *gyr_0, const double *g) {
  WindowEstimator *e = E(h);
  e->phase = WindowEstimator::NON_LINEAR;
  e->frame_count = WINDOW_SIZE;
  e->first_imu = true;
  for (int i = 0; i < FRAMES; i++) e->kf(i).stamp = stamps[i];
  e->acc_prev = v3(acc_0), e->gyr_prev = v3(gyr_0), e->g = v3(g);
  e->last_R = e->kf(WINDOW_SIZE).R, e->last_P = e->kf(WINDOW_SIZE).P, e->last_R0 = e->kf(0).R, e->last_P0 = e->kf(0).P;
}

void lfvio_host_clear_state(void *h) { E(h)->reset(); }
void lfvio_host_slide_window(void *h) { E(h)->slide(); }
int lfvio_host_failure_detection(void *h) { return E(h)->diverged() ? 1 : 0; }

// out = {phase, marginalization flag, frame_count, old-frame slides, second-new slides, tracks continued by the last image, tracks, failure_occur}
void lfvio_host_get_flow(void *h, int *out) {
  WindowEstimator *e = E(h);
  out[0] = e->phase, out[1] = e->marg_flag, out[2] = e->frame_count, out[3] = e->slides_old, out[4] = e->slides_new;
  out[5] = e->tracked_last, out[6] = e->tracks.live(), out[7] = e->failure_occur ? 1 : 0;
}

// per frame: stamp, number of buffered IMU samples, span present, its sum_dt
void lfvio_host_get_buffers(void *h, double *stamps, int *num_samples, int *has_pre, double *sum_dt) {
  WindowEstimator *e = E(h);
  for (int i = 0; i < FRAMES; i++) {
    const ImuSpan &sp = e->span(i);
    stamps[i] = e->kf(i).stamp, num_samples[i] = sp.samples(), has_pre[i] = sp.present ? 1 : 0, sum_dt[i] = sp.present ? sp.sum_dt : 0.0;
  }
}

// Replays an LFVT trace (replay.h) and writes the trajectory file.
// stats (may be null) = {images, thrown, keyframes, non_keyframes, poses, failures, last_status, iterations, restarts, bootstraps}
int lfvio_host_replay(void *h, const char *trace_path, const char *traj_path, int max_images, int *stats) {
  Trace trace;
  if (!trace.load(trace_path)) return -3;
  ReplayStats st;
  int rc = replay(*E(h), trace, traj_path, max_images, &st);
  if (stats) std::memcpy(stats, &st, sizeof st);
  return rc;
}

// the same with the wall-clock milliseconds of every image handed over (ms[cap]); *n_ms = how many were written
int lfvio_host_replay_timed(void *h, const char *trace_path, const char *traj_path, int max_images, int *stats, double *ms, int cap, int *n_ms) {
  Trace trace;
  if (!trace.load(trace_path)) return -3;
  ReplayStats st;
  std::vector<double> t;
  int rc = replay(*E(h), trace, traj_path, max_images, &st, &t);
  if (stats) std::memcpy(stats, &st, sizeof st);
  const int n = std::min((int)t.size(), cap);
  if (ms) std::memcpy(ms, t.data(), sizeof(double) * n);
  if (n_ms) *n_ms = n;
  return rc;
}

// the decode of one feature record, for the wire-format test: fills ids[n] (ascending), pts[n][8]; returns n
int lfvio_host_decode_features(const char *trace_path, int image_index, int cap, int *ids, double *pts, double *stamp) {
  Trace trace;
  if (!trace.load(trace_path) || image_index < 0 || image_index >= (int)trace.images.size()) return -1;
  DecodedImage img;
  decodeFeatures(trace.images[image_index], &img);
  *stamp = trace.images[image_index].t;
  const int n = std::min((int)img.ids.size(), cap);
  std::memcpy(ids, img.ids.data(), sizeof(int) * n);
  std::memcpy(pts, img.pts.data(), sizeof(double) * 8 * n);
  return n;
}

void lfvio_host_vector2double(void *h) { E(h)->vector2double(); }
void lfvio_host_double2vector(void *h) { E(h)->double2vector(); }

void lfvio_host_get_para(void *h, double *pose, double *sb, double *ex, double *td, double *feature) {
  WindowEstimator *e = E(h);
  std::memcpy(pose, e->para_Pose, sizeof e->para_Pose);
  std::memcpy(sb, e->para_SpeedBias, sizeof e->para_SpeedBias);
  std::memcpy(ex, e->para_Ex_Pose[0], sizeof e->para_Ex_Pose[0]);
  *td = e->para_Td[0][0];
  std::copy(e->para_Feature.begin(), e->para_Feature.end(), feature);
}
void lfvio_host_set_para(void *h, const double *pose, const double *sb, const double *ex, double td, const double *feature, int n) {
  WindowEstimator *e = E(h);
  std::memcpy(e->para_Pose, pose, sizeof e->para_Pose);
  std::memcpy(e->para_SpeedBias, sb, sizeof e->para_SpeedBias);
  std::memcpy(e->para_Ex_Pose[0], ex, sizeof e->para_Ex_Pose[0]);
  e->para_Td[0][0] = td;
  e->para_Feature.assign(feature, feature + n);
}

// LfvioWindow exactly as optimization() hands it to the C-ABI (pointers stay valid until the next pack); returns the
// status of the device pre-integration of the spans that needed it
int lfvio_host_pack(void *h, LfvioWindow *out) {
  WindowEstimator *e = E(h);
  e->status = LFVIO_OK;
  e->refreshSpans(false);
  e->vector2double();
  e->pack(out);
  return e->status;
}

void lfvio_host_set_flag(void *h, int flag) { E(h)->marg_flag = flag; }
void lfvio_host_set_prior(void *h, const LfvioPrior *p) {
  WindowEstimator *e = E(h);
  (void)e->collectPrior();
  e->has_prior = p && p->valid;
  if (e->has_prior) e->prior = *p;
}
int lfvio_host_get_prior(void *h, LfvioPrior *out) {
  WindowEstimator *e = E(h);
  (void)e->collectPrior();
  if (!e->has_prior) {
    out->valid = 0;
    return 0;
  }
  *out = e->prior;
  return 1;
}

// 1 (default): one upload, everything on the device; 0: the literal lfvio_solve / double2vector / lfvio_marginalize flow
void lfvio_host_set_fused(void *h, int on) { E(h)->fused = on != 0; }
// HIP devices of the estimator (bit d = device d); before the first call that needs the device.  More than one bit: the
// optimization() of every frame runs landmark-sharded through an lfvio_group over them.
void lfvio_host_set_device_mask(unsigned mask) { config().device_mask = mask ? mask : 1u; }
void lfvio_host_set_local_shards(int n) { config().local_shards = n; }
void lfvio_host_set_split_call(int on) { config().split_call = on != 0; }
void lfvio_host_set_device_chain(int on) { config().device_chain = on != 0; }
// waits for the marginalization a split optimization() left running and adopts its prior (what the next pack() would do)
int lfvio_host_collect_prior(void *h) { return E(h)->collectPrior() ? 0 : E(h)->status; }
void lfvio_host_get_timers(void *h, double *out6, int reset) {
  WindowEstimator *e = E(h);
  for (int k = 0; k < 6; k++) {
    out6[k] = e->timers[k];
    if (reset) e->timers[k] = 0.0;
  }
}
int lfvio_host_uses_group(void *h) { return E(h)->group != nullptr; }

int lfvio_host_optimization(void *h) {
  WindowEstimator *e = E(h);
  e->optimization();
  return e->status;
}
int lfvio_host_last_iterations(void *h) { return E(h)->summary.num_iterations; }
double lfvio_host_last_cost(void *h) { return E(h)->summary.final_cost; }

}  // extern "C"
