// host_capi.cpp — extern "C" driver of the host mirror, so that the Python tests (and bench) can
// exercise Estimator::optimization() the way estimator_node.cpp would.
#include <cstring>

#include "estimator.h"
#include "replay.h"

using namespace lfvio;

extern "C" {

void *lfvio_host_create(void) { return new Estimator(); }
void lfvio_host_destroy(void *h) { delete (Estimator *)h; }

// globals of parameters.cpp (readParameters); p = {ACC_N, GYR_N, ACC_W, GYR_W, g_norm, TR, ROW, SOLVER_TIME, TD}
void lfvio_host_set_params(const double *p, int estimate_extrinsic, int estimate_td, int num_iterations) {
  ACC_N = p[0], GYR_N = p[1], ACC_W = p[2], GYR_W = p[3];
  G = Vector3d(0, 0, p[4]);
  TR = p[5], ROW = p[6], SOLVER_TIME = p[7], TD = p[8];
  ESTIMATE_EXTRINSIC = estimate_extrinsic, ESTIMATE_TD = estimate_td, NUM_ITERATIONS = num_iterations;
}

static void setM(Matrix3d &m, const double *a) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) m(i, j) = a[i * 3 + j];
}
static void getM(const Matrix3d &m, double *a) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) a[i * 3 + j] = m(i, j);
}

void lfvio_host_set_state(void *h, const double *Ps, const double *Rs, const double *Vs, const double *Bas, const double *Bgs,
                          const double *tic, const double *ric, double td) {
  Estimator *e = (Estimator *)h;
  for (int i = 0; i <= WINDOW_SIZE; i++) {
    e->Ps[i] = Vector3d(Ps[3 * i], Ps[3 * i + 1], Ps[3 * i + 2]);
    e->Vs[i] = Vector3d(Vs[3 * i], Vs[3 * i + 1], Vs[3 * i + 2]);
    e->Bas[i] = Vector3d(Bas[3 * i], Bas[3 * i + 1], Bas[3 * i + 2]);
    e->Bgs[i] = Vector3d(Bgs[3 * i], Bgs[3 * i + 1], Bgs[3 * i + 2]);
    setM(e->Rs[i], Rs + 9 * i);
  }
  e->tic[0] = Vector3d(tic[0], tic[1], tic[2]);
  setM(e->ric[0], ric);
  e->td = td;
}

void lfvio_host_get_state(void *h, double *Ps, double *Rs, double *Vs, double *Bas, double *Bgs, double *tic, double *ric, double *td) {
  Estimator *e = (Estimator *)h;
  for (int i = 0; i <= WINDOW_SIZE; i++) {
    for (int k = 0; k < 3; k++) {
      Ps[3 * i + k] = e->Ps[i](k), Vs[3 * i + k] = e->Vs[i](k), Bas[3 * i + k] = e->Bas[i](k), Bgs[3 * i + k] = e->Bgs[i](k);
    }
    getM(e->Rs[i], Rs + 9 * i);
  }
  for (int k = 0; k < 3; k++) tic[k] = e->tic[0](k);
  getM(e->ric[0], ric);
  *td = e->td;
}

void lfvio_host_clear_features(void *h) { ((Estimator *)h)->f_manager.clearState(); }

// obs: n x 8 = [bearing xyz, pixel uv, bearing-velocity xyz] like the 8-vector of estimator_node.cpp:308
void lfvio_host_add_feature(void *h, int id, int start_frame, int n, const double *obs, const double *cur_td, double estimated_depth) {
  Estimator *e = (Estimator *)h;
  FeaturePerId &f = e->f_manager.addFeature(id, start_frame);
  for (int k = 0; k < n; k++) f.feature_per_frame.push_back(FeaturePerFrame(obs + 8 * k, cur_td[k]));
  f.estimated_depth = estimated_depth;
}

// FeatureManager::triangulate with the estimator's own Ps / tic / ric (estimator.cpp:473)
int lfvio_host_triangulate(void *h) {
  Estimator *e = (Estimator *)h;
  e->f_manager.triangulate(e->Ps, e->tic, e->ric);
  return e->f_manager.last_status;
}
// FeatureManager::removeBackShiftDepth with the arguments slideWindowOld() forms (estimator.cpp:1120-1127): the
// marginalized frame's pose is passed in (back_R0, back_P0), the new frame 0 is the estimator's
int lfvio_host_remove_back_shift_depth(void *h, const double *back_R0, const double *back_P0) {
  Estimator *e = (Estimator *)h;
  Matrix3d bR;
  setM(bR, back_R0);
  const Vector3d bP(back_P0[0], back_P0[1], back_P0[2]);
  const Matrix3d R0 = bR * e->ric[0], R1 = e->Rs[0] * e->ric[0];
  const Vector3d P0 = bP + bR * e->tic[0], P1 = e->Ps[0] + e->Rs[0] * e->tic[0];
  e->f_manager.removeBackShiftDepth(R0, P0, R1, P1);
  return e->f_manager.last_status;
}
void lfvio_host_set_depths(void *h, const double *d, int n) {
  int k = 0;
  for (auto &f : ((Estimator *)h)->f_manager.feature)
    if (k < n) f.estimated_depth = d[k++];
}
int lfvio_host_num_features(void *h) { return (int)((Estimator *)h)->f_manager.feature.size(); }
// (feature_id, start_frame, number of observations, estimated_depth) of every feature in list order
void lfvio_host_list_features(void *h, int *ids, int *start, int *count, double *depth) {
  int k = 0;
  for (auto &f : ((Estimator *)h)->f_manager.feature) {
    ids[k] = f.feature_id, start[k] = f.start_frame, count[k] = (int)f.feature_per_frame.size(), depth[k] = f.estimated_depth;
    k++;
  }
}

int lfvio_host_feature_count(void *h) { return ((Estimator *)h)->f_manager.getFeatureCount(); }
void lfvio_host_get_depths(void *h, double *out) {
  Estimator *e = (Estimator *)h;
  int k = 0;
  for (auto &f : e->f_manager.feature) out[k++] = f.estimated_depth;
}

// pre_integrations[frame] := IntegrationBase{acc0, gyr0, ba, bg}; push_back() of the n samples (processIMU, estimator.cpp:86-120)
void lfvio_host_set_imu(void *h, int frame, const double *acc0, const double *gyr0, const double *ba, const double *bg, int n,
                        const double *dt, const double *acc, const double *gyr) {
  Estimator *e = (Estimator *)h;
  delete e->pre_integrations[frame];
  e->pre_integrations[frame] = new IntegrationBase(Vector3d(acc0[0], acc0[1], acc0[2]), Vector3d(gyr0[0], gyr0[1], gyr0[2]),
                                                   Vector3d(ba[0], ba[1], ba[2]), Vector3d(bg[0], bg[1], bg[2]));
  for (int k = 0; k < n; k++)
    e->pre_integrations[frame]->push_back(dt[k], Vector3d(acc[3 * k], acc[3 * k + 1], acc[3 * k + 2]),
                                          Vector3d(gyr[3 * k], gyr[3 * k + 1], gyr[3 * k + 2]));
}

void lfvio_host_repropagate(void *h, int frame, const double *ba, const double *bg) {
  Estimator *e = (Estimator *)h;
  if (e->pre_integrations[frame]) e->pre_integrations[frame]->repropagate(Vector3d(ba[0], ba[1], ba[2]), Vector3d(bg[0], bg[1], bg[2]));
}

// ba, bg: [WINDOW_SIZE + 1][3]; returns Estimator::last_status
int lfvio_host_repropagate_window(void *h, const double *ba, const double *bg) {
  Estimator *e = (Estimator *)h;
  Vector3d a[WINDOW_SIZE + 1], g[WINDOW_SIZE + 1];
  for (int i = 0; i <= WINDOW_SIZE; i++) a[i] = Vector3d(ba[3 * i], ba[3 * i + 1], ba[3 * i + 2]), g[i] = Vector3d(bg[3 * i], bg[3 * i + 1], bg[3 * i + 2]);
  e->repropagateWindow(a, g);
  return e->last_status;
}

// ---- the control flow around optimization() and the trace replay
void lfvio_host_set_min_parallax(double keyframe_parallax_px) { MIN_PARALLAX = keyframe_parallax_px / FOCAL_LENGTH; }

void lfvio_host_process_imu(void *h, double dt, const double *acc, const double *gyr) {
  ((Estimator *)h)->processIMU(dt, Vector3d(acc[0], acc[1], acc[2]), Vector3d(gyr[0], gyr[1], gyr[2]));
}

// ids[n], pts[n][8] = x y z u v vx vy vz (camera 0); returns Estimator::last_status (or FeatureManager's when that failed)
int lfvio_host_process_image(void *h, double stamp, int n, const int *ids, const double *pts) {
  Estimator *e = (Estimator *)h;
  ImageMap image;
  for (int i = 0; i < n; i++) {
    Vector8d v;
    for (int k = 0; k < 8; k++) v.a[k] = pts[8 * i + k];
    image[ids[i]].emplace_back(0, v);
  }
  e->last_status = LFVIO_OK, e->f_manager.last_status = LFVIO_OK;
  e->processImage(image, stamp);
  return e->last_status != LFVIO_OK ? e->last_status : e->f_manager.last_status;
}

// only the keyframe decision of processImage(): appends the observations, returns 1 for MARGIN_OLD
int lfvio_host_add_feature_check_parallax(void *h, int frame_count, int n, const int *ids, const double *pts, double td) {
  Estimator *e = (Estimator *)h;
  ImageMap image;
  for (int i = 0; i < n; i++) {
    Vector8d v;
    for (int k = 0; k < 8; k++) v.a[k] = pts[8 * i + k];
    image[ids[i]].emplace_back(0, v);
  }
  return e->f_manager.addFeatureCheckParallax(frame_count, image, td) ? 1 : 0;
}

// Ps Rs Vs Bas Bgs as lfvio_host_set_state, g[3]
void lfvio_host_set_bootstrap(void *h, const double *Ps, const double *Rs, const double *Vs, const double *Bas, const double *Bgs, const double *g) {
  Estimator *e = (Estimator *)h;
  for (int i = 0; i <= WINDOW_SIZE; i++) {
    e->bootstrap.Ps[i] = Vector3d(Ps[3 * i], Ps[3 * i + 1], Ps[3 * i + 2]);
    e->bootstrap.Vs[i] = Vector3d(Vs[3 * i], Vs[3 * i + 1], Vs[3 * i + 2]);
    e->bootstrap.Bas[i] = Vector3d(Bas[3 * i], Bas[3 * i + 1], Bas[3 * i + 2]);
    e->bootstrap.Bgs[i] = Vector3d(Bgs[3 * i], Bgs[3 * i + 1], Bgs[3 * i + 2]);
    setM(e->bootstrap.Rs[i], Rs + 9 * i);
  }
  e->bootstrap.g = Vector3d(g[0], g[1], g[2]);
  e->bootstrap.valid = true;
}

// the window is already filled from outside (lfvio_host_set_state / add_feature / set_imu): continue from NON_LINEAR
void lfvio_host_set_running(void *h, const double *stamps, const double *acc_0, const double *gyr_0, const double *g) {
  Estimator *e = (Estimator *)h;
  e->solver_flag = Estimator::NON_LINEAR;
  e->frame_count = WINDOW_SIZE;
  e->first_imu = true;
  for (int i = 0; i <= WINDOW_SIZE; i++) e->Headers[i] = stamps[i];
  e->acc_0 = Vector3d(acc_0[0], acc_0[1], acc_0[2]), e->gyr_0 = Vector3d(gyr_0[0], gyr_0[1], gyr_0[2]);
  e->g = Vector3d(g[0], g[1], g[2]);
  e->last_R = e->Rs[WINDOW_SIZE], e->last_P = e->Ps[WINDOW_SIZE], e->last_R0 = e->Rs[0], e->last_P0 = e->Ps[0];
}

void lfvio_host_clear_state(void *h) {
  Estimator *e = (Estimator *)h;
  e->clearState();
  e->setParameter();
  e->bootstrap.valid = false;
}

void lfvio_host_slide_window(void *h) { ((Estimator *)h)->slideWindow(); }
int lfvio_host_failure_detection(void *h) { return ((Estimator *)h)->failureDetection() ? 1 : 0; }

// out = {solver_flag, marginalization_flag, frame_count, sum_of_back, sum_of_front, last_track_num, feature count, failure_occur}
void lfvio_host_get_flow(void *h, int *out) {
  Estimator *e = (Estimator *)h;
  out[0] = e->solver_flag, out[1] = e->marginalization_flag, out[2] = e->frame_count, out[3] = e->sum_of_back, out[4] = e->sum_of_front;
  out[5] = e->f_manager.last_track_num, out[6] = (int)e->f_manager.feature.size(), out[7] = e->failure_occur ? 1 : 0;
}

// per frame: Headers, number of buffered IMU samples, pre_integrations[i] != nullptr, its sum_dt
void lfvio_host_get_buffers(void *h, double *stamps, int *num_samples, int *has_pre, double *sum_dt) {
  Estimator *e = (Estimator *)h;
  for (int i = 0; i <= WINDOW_SIZE; i++) {
    stamps[i] = e->Headers[i], num_samples[i] = (int)e->dt_buf[i].size(), has_pre[i] = e->pre_integrations[i] ? 1 : 0;
    sum_dt[i] = e->pre_integrations[i] ? e->pre_integrations[i]->sum_dt : 0.0;
  }
}

// Replays an LFVT trace (host/replay.h) through processIMU / processImage and writes the trajectory file.
// stats (may be null) = {images, thrown, keyframes, non_keyframes, poses, failures, last_status, iterations}
int lfvio_host_replay(void *h, const char *trace_path, const char *traj_path, int max_images, int *stats) {
  Trace trace;
  if (!trace.load(trace_path)) return -3;
  ReplayStats st;
  int rc = replay(*(Estimator *)h, trace, traj_path, max_images, &st);
  if (stats) std::memcpy(stats, &st, sizeof st);
  return rc;
}

// the decode of one feature record, for the wire-format test: fills ids[n], pts[n][8]; returns n
int lfvio_host_decode_features(const char *trace_path, int image_index, int cap, int *ids, double *pts, double *stamp) {
  Trace trace;
  if (!trace.load(trace_path) || image_index < 0 || image_index >= (int)trace.images.size()) return -1;
  ImageMap m = decodeFeatures(trace.images[image_index]);
  *stamp = trace.images[image_index].t;
  int n = 0;
  for (auto &kv : m) {
    if (n >= cap) break;
    ids[n] = kv.first;
    for (int k = 0; k < 8; k++) pts[8 * n + k] = kv.second[0].second.a[k];
    n++;
  }
  return n;
}

void lfvio_host_vector2double(void *h) { ((Estimator *)h)->vector2double(); }
void lfvio_host_double2vector(void *h) { ((Estimator *)h)->double2vector(); }

void lfvio_host_get_para(void *h, double *pose, double *sb, double *ex, double *td, double *feature) {
  Estimator *e = (Estimator *)h;
  std::memcpy(pose, e->para_Pose, sizeof e->para_Pose);
  std::memcpy(sb, e->para_SpeedBias, sizeof e->para_SpeedBias);
  std::memcpy(ex, e->para_Ex_Pose[0], sizeof e->para_Ex_Pose[0]);
  *td = e->para_Td[0][0];
  for (size_t i = 0; i < e->para_Feature.size(); i++) feature[i] = e->para_Feature[i];
}
void lfvio_host_set_para(void *h, const double *pose, const double *sb, const double *ex, double td, const double *feature, int n) {
  Estimator *e = (Estimator *)h;
  std::memcpy(e->para_Pose, pose, sizeof e->para_Pose);
  std::memcpy(e->para_SpeedBias, sb, sizeof e->para_SpeedBias);
  std::memcpy(e->para_Ex_Pose[0], ex, sizeof e->para_Ex_Pose[0]);
  e->para_Td[0][0] = td;
  e->para_Feature.assign(feature, feature + n);
}

// LfvioWindow exactly as optimization() hands it to the C-ABI (pointers stay valid until the next pack)
void lfvio_host_pack(void *h, LfvioWindow *out) {
  Estimator *e = (Estimator *)h;
  e->vector2double();
  e->packWindow(out);
}

void lfvio_host_set_flag(void *h, int flag) { ((Estimator *)h)->marginalization_flag = (Estimator::MarginalizationFlag)flag; }
void lfvio_host_set_prior(void *h, const LfvioPrior *p) {
  Estimator *e = (Estimator *)h;
  delete e->last_marginalization_info;
  e->last_marginalization_info = (p && p->valid) ? new LfvioPrior(*p) : nullptr;
}
int lfvio_host_get_prior(void *h, LfvioPrior *out) {
  Estimator *e = (Estimator *)h;
  if (!e->last_marginalization_info) {
    out->valid = 0;
    return 0;
  }
  *out = *e->last_marginalization_info;
  return 1;
}

// 1 (default): one upload, everything on the device; 0: the literal lfvio_solve / double2vector / lfvio_marginalize flow
void lfvio_host_set_fused(void *h, int on) { ((Estimator *)h)->fused = on != 0; }

int lfvio_host_optimization(void *h) {
  Estimator *e = (Estimator *)h;
  e->optimization();
  return e->last_status;
}
int lfvio_host_last_iterations(void *h) { return ((Estimator *)h)->last_summary.num_iterations; }
double lfvio_host_last_cost(void *h) { return ((Estimator *)h)->last_summary.final_cost; }

}  // extern "C"
