// window_estimator.cpp — see window_estimator.h.  Reference lines are relative to /root/reference/vins_estimator/src.
#include "window_estimator.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

namespace lfvio {

Config &config() {
  static Config c;
  return c;
}

Vector3d yawPitchRollDeg(const Matrix3d &R) {  // utility.h:66-83
  const Vector3d n = R.col(0), o = R.col(1), a = R.col(2);
  const double y = atan2(n(1), n(0));
  const double p = atan2(-n(2), n(0) * cos(y) + n(1) * sin(y));
  const double r = atan2(a(0) * sin(y) - a(1) * cos(y), -o(0) * sin(y) + o(1) * cos(y));
  return Vector3d(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);
}

Matrix3d fromYawPitchRollDeg(const Vector3d &ypr) {  // utility.h:85-113
  const double y = ypr(0) / 180.0 * M_PI, p = ypr(1) / 180.0 * M_PI, r = ypr(2) / 180.0 * M_PI;
  Matrix3d Rz, Ry, Rx;
  Rz(0, 0) = cos(y), Rz(0, 1) = -sin(y), Rz(1, 0) = sin(y), Rz(1, 1) = cos(y), Rz(2, 2) = 1;
  Ry(0, 0) = cos(p), Ry(0, 2) = sin(p), Ry(1, 1) = 1, Ry(2, 0) = -sin(p), Ry(2, 2) = cos(p);
  Rx(0, 0) = 1, Rx(1, 1) = cos(r), Rx(1, 2) = -sin(r), Rx(2, 1) = sin(r), Rx(2, 2) = cos(r);
  return Rz * Ry * Rx;
}

// ---------------------------------------------------------------------------------------------------- ImuSpan
void ImuSpan::open(const Vector3d &a0, const Vector3d &g0, const Vector3d &ba, const Vector3d &bg) {
  close();
  present = true;
  for (int k = 0; k < 3; k++) acc0[k] = a0(k), gyr0[k] = g0(k), lin_ba[k] = ba(k), lin_bg[k] = bg(k);
  // an IntegrationBase without samples: identity Jacobian, zero covariance (integration_base.h:13-28)
  std::memset(&pre, 0, sizeof pre);
  pre.delta_q[3] = 1.0;
  for (int k = 0; k < 15; k++) pre.jacobian[16 * k] = 1.0;
  for (int k = 0; k < 3; k++) pre.linearized_ba[k] = lin_ba[k], pre.linearized_bg[k] = lin_bg[k];
}

void ImuSpan::push(double d, const double *a, const double *g) {
  dt.push_back(d);
  acc.insert(acc.end(), a, a + 3);
  gyr.insert(gyr.end(), g, g + 3);
  sum_dt += d;
  dirty = true;
}

// ---------------------------------------------------------------------------------------------------- TrackTable
void TrackTable::clear() {
  id_.clear(), start_.clear(), count_.clear(), first_.clear(), flag_.clear(), dead_.clear();
  depth_.clear(), rows_.clear(), order_.clear(), free_.clear(), slot_of_.clear();
  holes_ = false;
}

int TrackTable::solvableCount() const {
  int n = 0;
  for (int s : order_) n += solvable(s);
  return n;
}

int TrackTable::find(int feature_id) const {
  auto it = slot_of_.find(feature_id);
  return it == slot_of_.end() ? -1 : it->second;
}

int TrackTable::create(int feature_id, int start_frame) {
  int s;
  if (!free_.empty()) {
    s = free_.back();
    free_.pop_back();
  } else {
    s = (int)id_.size();
    id_.push_back(0), start_.push_back(0), count_.push_back(0), first_.push_back(0), flag_.push_back(0), dead_.push_back(0);
    depth_.push_back(0.0);
    rows_.resize(rows_.size() + (size_t)FRAMES * OBS_W);
  }
  id_[s] = feature_id, start_[s] = start_frame, count_[s] = 0, first_[s] = 0, flag_[s] = 0, dead_[s] = 0;
  depth_[s] = -1.0;  // FeaturePerId: estimated_depth(-1.0), solve_flag(0) (feature_manager.h:60-63)
  slot_of_[feature_id] = s;
  order_.push_back(s);
  return s;
}

void TrackTable::append(int s, const double *pt8, double cur_td) {
  if (count_[s] >= FRAMES) return;  // a track cannot be longer than the window
  double *r = &rows_[((size_t)s * FRAMES + (first_[s] + count_[s]) % FRAMES) * OBS_W];
  std::memcpy(r, pt8, 8 * sizeof(double));
  r[8] = cur_td;
  count_[s]++;
}

int TrackTable::appendFrame(int frame_count, int n, const int *ids, const double *pts8, double td) {
  // the reference walks a std::map keyed by feature id (estimator_node.cpp:292-312): ascending ids, and of several
  // entries with one id only the first is used (id_pts.second[0])
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return ids[a] < ids[b]; });
  int continued = 0;
  for (int k = 0; k < n; k++) {
    const int i = idx[k];
    if (k > 0 && ids[idx[k - 1]] == ids[i]) continue;
    int s = find(ids[i]);
    if (s < 0) {
      s = create(ids[i], frame_count);
    } else {
      continued++;
    }
    append(s, pts8 + 8 * (size_t)i, td);
  }
  return continued;
}

void TrackTable::parallax(int frame_count, double *sum, int *num) const {
  double acc = 0;
  int cnt = 0;
  for (int s : order_) {
    if (!(start_[s] <= frame_count - 2 && start_[s] + count_[s] - 1 >= frame_count - 1)) continue;
    const double *pi = obs(s, frame_count - 2 - start_[s]), *pj = obs(s, frame_count - 1 - start_[s]);
    acc += acos(pi[0] * pj[0] + pi[1] * pj[1] + pi[2] * pj[2]) * 10;  // the angle between the two bearings, uncompensated
    cnt++;
  }
  *sum = acc, *num = cnt;
}

void TrackTable::erase(int s) {
  dead_[s] = 1;
  slot_of_.erase(id_[s]);
  free_.push_back(s);
  holes_ = true;
}

void TrackTable::compact() {
  if (!holes_) return;
  order_.erase(std::remove_if(order_.begin(), order_.end(), [&](int s) { return dead_[s] != 0; }), order_.end());
  holes_ = false;
}

void TrackTable::dropOldestFrame(std::vector<Shifted> *survivors) {
  for (int s : order_) {
    if (start_[s] != 0) {
      start_[s]--;
      continue;
    }
    const double *b = obs(s, 0);
    Shifted sh{s, {b[0], b[1], b[2]}};
    first_[s] = (first_[s] + 1) % FRAMES;  // the row stays where it is; the track just starts one row later
    count_[s]--;
    if (survivors) {
      if (count_[s] < 2) erase(s);
      else survivors->push_back(sh);
    } else if (count_[s] == 0) {
      erase(s);
    }
  }
  compact();
}

void TrackTable::dropSecondNewestFrame(int frame_count) {
  for (int s : order_) {
    if (start_[s] == frame_count) {
      start_[s]--;
      continue;
    }
    if (start_[s] + count_[s] - 1 < frame_count - 1) continue;
    const int j = WINDOW_SIZE - 1 - start_[s];
    // at most the newest frame's observation sits behind row j
    for (int k = j; k + 1 < count_[s]; k++)
      std::memcpy(const_cast<double *>(obs(s, k)), obs(s, k + 1), OBS_W * sizeof(double));
    if (--count_[s] == 0) erase(s);
  }
  compact();
}

void TrackTable::dropFailed() {
  for (int s : order_)
    if (flag_[s] == 2) erase(s);
  compact();
}

// ---------------------------------------------------------------------------------------------------- WindowEstimator
WindowEstimator::WindowEstimator() {
  std::memset(&summary, 0, sizeof summary);
  std::memset(&prior, 0, sizeof prior);
  std::memset(&next_, 0, sizeof next_);
  reset();
}

WindowEstimator::~WindowEstimator() {
  if (group) lfvio_group_destroy(group);  // owns its contexts, `gpu` among them
  else if (gpu) lfvio_destroy(gpu);
}

namespace {
// what is there of a prior: header, n x n of the Jacobian slots, n residuals (46 KB instead of 240 KB for n = 76)
void assign_prior(LfvioPrior *dst, const LfvioPrior *src) {
  std::memcpy(dst, src, offsetof(LfvioPrior, linearized_jacobians));
  if (src->valid && src->n > 0 && src->n <= LFVIO_MAX_PRIOR_DIM) {
    std::memcpy(dst->linearized_jacobians, src->linearized_jacobians, sizeof(double) * src->n * src->n);
    std::memcpy(dst->linearized_residuals, src->linearized_residuals, sizeof(double) * src->n);
  }
}
}  // namespace

// The marginalization of the last optimization() may still be running on the device (Config::split_call): wait for it and
// adopt its prior.  Everything that reads `prior` starts here.
namespace {
struct Stopwatch {  // adds its lifetime to *acc
  double *acc;
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  explicit Stopwatch(double *a) : acc(a) {}
  ~Stopwatch() { *acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

bool WindowEstimator::collectPrior() {
  if (!prior_pending_) return true;
  Stopwatch sw(&timers[1]);
  prior_pending_ = false;
  const int rc = lfvio_batch_optimize_finish(gpu, &next_);
  if (rc != LFVIO_OK) {
    status = rc;
    return false;
  }
  assign_prior(&prior, &next_), has_prior = next_.valid != 0;
  return true;
}

bool WindowEstimator::device() {
  if (!gpu) {
    const unsigned mask = config().device_mask ? config().device_mask : 1u;
    if (config().local_shards > 1) {
      int d = 0;
      while (!(mask & (1u << d))) d++;
      group = lfvio_group_create_local(d, config().local_shards);
      gpu = group ? lfvio_group_ctx(group, 0) : nullptr;
    } else if (mask & (mask - 1)) {  // several devices: one group, whose first context also serves the single-device calls
      group = lfvio_group_create(mask);
      gpu = group ? lfvio_group_ctx(group, 0) : nullptr;
    } else {
      int d = 0;
      while (!(mask & (1u << d))) d++;
      gpu = lfvio_create(d);
    }
  }
  if (!gpu) status = LFVIO_ERR_DEVICE;  // there is no host fallback: the caller sees the failure
  return gpu != nullptr;
}

void WindowEstimator::reset() {
  // clearState() (estimator.cpp:23-84) followed by setParameter() (:10-21): the CONFIGURED extrinsic and td come back,
  // and a bootstrap record that described the old window is not reused
  const Config &c = config();
  if (prior_pending_) {  // the prior of a window that is being dropped: wait for the device, keep nothing
    prior_pending_ = false;
    (void)lfvio_batch_optimize_finish(gpu, nullptr);
  }
  ring_.head = 0;
  for (int i = 0; i < FRAMES; i++) {
    frames_[i] = Keyframe();
    frames_[i].R.setIdentity();
    spans_[i].close();
  }
  for (int k = 0; k < 3; k++) tic(k) = c.tic[k];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) ric(i, j) = c.ric[3 * i + j];
  td = c.td;
  phase = INITIAL;
  first_imu = false;
  slides_old = slides_new = 0;
  frame_count = 0;
  initial_timestamp = 0;
  has_prior = false;
  tracks.clear();
  failure_occur = false;
  bootstrap.valid = false;
}

void WindowEstimator::pushImu(double dt, const double acc[3], const double gyr[3]) {  // estimator.cpp:86-120
  const Vector3d a(acc[0], acc[1], acc[2]), w(gyr[0], gyr[1], gyr[2]);
  if (!first_imu) {
    first_imu = true;
    acc_prev = a, gyr_prev = w;
  }
  Keyframe &f = kf(frame_count);
  ImuSpan &sp = span(frame_count);
  if (!sp.present) sp.open(acc_prev, gyr_prev, f.Ba, f.Bg);
  if (frame_count != 0) {
    sp.push(dt, acc, gyr);
    // mid-point dead reckoning of the newest keyframe
    const Vector3d a0 = f.R * (acc_prev - f.Ba) - g;
    const Vector3d wm = 0.5 * (gyr_prev + w) - f.Bg;
    f.R = f.R * Quaterniond(1.0, wm(0) * dt / 2.0, wm(1) * dt / 2.0, wm(2) * dt / 2.0).toRotationMatrix();  // Utility::deltaQ, unnormalised
    const Vector3d a1 = f.R * (a - f.Ba) - g;
    const Vector3d am = 0.5 * (a0 + a1);
    f.P += dt * f.V + 0.5 * dt * dt * am;
    f.V += dt * am;
  }
  acc_prev = a, gyr_prev = w;
}

bool WindowEstimator::keyframeTest(int fc, int n, const int *ids, const double *pts8, double cur_td) {  // feature_manager.cpp:45-95
  tracked_last = tracks.appendFrame(fc, n, ids, pts8, cur_td);
  if (fc < 2 || tracked_last < 20) return true;
  double sum;
  int num;
  tracks.parallax(fc, &sum, &num);
  if (num == 0) return true;
  return sum / num >= config().min_parallax;
}

bool WindowEstimator::applyBootstrap() {
  // stand-in for initialStructure() + visualInitialAlign() (estimator.cpp:222-473): the aligned window state comes from the
  // record; what the reference does with it afterwards is kept: every span is integrated again with its new gyroscope bias
  // and a zero accelerometer bias (:403-406), gravity is taken over (:445), every depth is marked unknown (:386-390)
  if (!bootstrap.valid) return false;
  for (int i = 0; i < FRAMES; i++) {
    const double stamp = kf(i).stamp;
    kf(i) = bootstrap.kf[i];
    kf(i).stamp = stamp;
  }
  g = bootstrap.g;
  Vector3d zero[FRAMES], bg[FRAMES];
  for (int i = 0; i < FRAMES; i++) bg[i] = kf(i).Bg;
  if (!refreshSpans(true, zero, bg)) return false;
  for (int s : tracks.order()) tracks.setDepth(s, -1.0);
  return true;
}

void WindowEstimator::pushImage(double stamp, int n, const int *ids, const double *pts8) {  // estimator.cpp:122-220
  marg_flag = keyframeTest(frame_count, n, ids, pts8, td) ? LFVIO_MARGIN_OLD : LFVIO_MARGIN_SECOND_NEW;
  kf(frame_count).stamp = stamp;
  auto remember = [&] { last_R = kf(WINDOW_SIZE).R, last_P = kf(WINDOW_SIZE).P, last_R0 = kf(0).R, last_P0 = kf(0).P; };
  if (phase == INITIAL) {
    if (frame_count < WINDOW_SIZE) {
      frame_count++;
      return;
    }
    bool aligned = false;
    if (config().estimate_extrinsic != 2 && stamp - initial_timestamp > 0.1) {
      aligned = applyBootstrap();
      initial_timestamp = stamp;
    }
    if (!aligned) {
      slide();
      return;
    }
    phase = NON_LINEAR;
    triangulate();
    optimization();
    slide();
    tracks.dropFailed();
    remember();
    return;
  }
  if (frame_count == WINDOW_SIZE) {  // solveOdometry(), estimator.cpp:475-486
    triangulate();
    optimization();
  }
  if (diverged()) {
    failure_occur = true;
    reset();  // as shipped, clearState() ends with failure_occur = 0 (estimator.cpp:82), so the flag does not survive the reboot
    return;
  }
  slide();
  tracks.dropFailed();
  remember();
}

bool WindowEstimator::diverged() const {  // estimator.cpp:628-674 as shipped: the other tests only log
  const Keyframe &f = kf(WINDOW_SIZE);
  if (f.Bg.norm() > 1.0) return true;
  if ((f.P - last_P).norm() > 5) return true;
  return std::abs(f.P.z() - last_P.z()) > 1;
}

void WindowEstimator::slide() {  // estimator.cpp:1011-1131
  if (frame_count != WINDOW_SIZE) return;
  Keyframe &newest = kf(WINDOW_SIZE);
  if (marg_flag == LFVIO_MARGIN_OLD) {
    const Matrix3d old_R = kf(0).R;
    const Vector3d old_P = kf(0).P;
    // the ring turns: frame 1 becomes frame 0, ...; the slot of the old frame 0 becomes the new newest frame, which starts
    // as a copy of its predecessor with an empty span
    const Keyframe prev = newest;
    ring_.advance();
    kf(WINDOW_SIZE) = prev;
    span(WINDOW_SIZE).open(acc_prev, gyr_prev, prev.Ba, prev.Bg);
    slides_old++;
    if (phase == NON_LINEAR) {
      std::vector<TrackTable::Shifted> moved;
      tracks.dropOldestFrame(&moved);
      reanchorDepths(old_R * ric, old_P + old_R * tic, kf(0).R * ric, kf(0).P + kf(0).R * tic, moved);
    } else {
      tracks.dropOldestFrame(nullptr);
    }
    return;
  }
  // MARGIN_SECOND_NEW: the newest frame replaces the one before it, whose span takes the newest span's samples as well
  ImuSpan &keep = span(WINDOW_SIZE - 1), &gone = span(WINDOW_SIZE);
  for (int k = 0; k < gone.samples(); k++) keep.push(gone.dt[k], &gone.acc[3 * k], &gone.gyr[3 * k]);
  const double stamp = newest.stamp;
  kf(WINDOW_SIZE - 1) = newest;
  kf(WINDOW_SIZE - 1).stamp = stamp;
  gone.open(acc_prev, gyr_prev, newest.Ba, newest.Bg);
  slides_new++;
  tracks.dropSecondNewestFrame(frame_count);
}

void WindowEstimator::triangulate() {  // feature_manager.cpp:199-253: the 2k x 4 SVD per landmark runs on the device
  Stopwatch sw(&timers[2]);
  Staging &st = stage_;
  st.start_frame.clear(), st.obs_offset.assign(1, 0), st.point.clear(), st.lam_out.clear();
  std::vector<int> sel;
  for (int s : tracks.order()) {
    if (!tracks.solvable(s) || tracks.depth(s) > 0) continue;
    sel.push_back(s);
    st.start_frame.push_back(tracks.start(s));
    for (int k = 0; k < tracks.count(s); k++) {
      const double *o = tracks.obs(s, k);
      st.point.insert(st.point.end(), o, o + 3);
    }
    st.obs_offset.push_back((int)st.point.size() / 3);
    st.lam_out.push_back(tracks.depth(s));
  }
  if (sel.empty() || !device()) return;
  LfvioTriangulateIn in;
  in.num_landmarks = (int)sel.size(), in.num_observations = (int)st.point.size() / 3;
  in.start_frame = st.start_frame.data(), in.obs_offset = st.obs_offset.data(), in.obs_point = st.point.data();
  for (int f = 0; f < FRAMES; f++)
    for (int i = 0; i < 3; i++) {
      in.Ps[f][i] = kf(f).P(i);
      for (int j = 0; j < 3; j++) in.Rs[f][3 * i + j] = kf(f).R(i, j);
    }
  for (int i = 0; i < 3; i++) {
    in.tic[i] = tic(i);
    for (int j = 0; j < 3; j++) in.ric[3 * i + j] = ric(i, j);
  }
  in.init_depth = config().init_depth;
  status = lfvio_triangulate(gpu, &in, st.lam_out.data());
  if (status != LFVIO_OK) return;
  for (size_t k = 0; k < sel.size(); k++) tracks.setDepth(sel[k], st.lam_out[k]);
}

void WindowEstimator::reanchorDepths(const Matrix3d &old_R, const Vector3d &old_P, const Matrix3d &new_R, const Vector3d &new_P,
                                     const std::vector<TrackTable::Shifted> &moved) {  // feature_manager.cpp:291-299
  Stopwatch sw(&timers[3]);
  if (moved.empty() || !device()) return;
  std::vector<double> uv(3 * moved.size()), depth(moved.size());
  for (size_t k = 0; k < moved.size(); k++) {
    std::memcpy(&uv[3 * k], moved[k].bearing, 3 * sizeof(double));
    depth[k] = tracks.depth(moved[k].slot);
  }
  double mR[9], nR[9], mP[3], nP[3];
  for (int i = 0; i < 3; i++) {
    mP[i] = old_P(i), nP[i] = new_P(i);
    for (int j = 0; j < 3; j++) mR[3 * i + j] = old_R(i, j), nR[3 * i + j] = new_R(i, j);
  }
  status = lfvio_shift_depth(gpu, (int)moved.size(), uv.data(), mR, mP, nR, nP, config().init_depth, depth.data());
  if (status != LFVIO_OK) return;
  for (size_t k = 0; k < moved.size(); k++) tracks.setDepth(moved[k].slot, depth[k]);
}

bool WindowEstimator::refreshSpans(bool all, const Vector3d *ba, const Vector3d *bg) {
  Stopwatch sw(&timers[4]);
  // IntegrationBase::{push_back, propagate, repropagate} (integration_base.h:29-158) for every span whose samples or
  // linearization biases changed since its `pre` was made — all of them in ONE device call
  std::vector<LfvioImuInterval> in;
  std::vector<int> which;
  for (int i = 0; i < FRAMES; i++) {
    ImuSpan &sp = span(i);
    if (!sp.present) continue;
    if (ba)
      for (int k = 0; k < 3; k++) sp.lin_ba[k] = ba[i](k), sp.lin_bg[k] = bg[i](k);
    if (!(all || sp.dirty)) continue;
    LfvioImuInterval iv;
    iv.num_samples = sp.samples();
    iv.dt = sp.dt.data(), iv.acc = sp.acc.data(), iv.gyr = sp.gyr.data();
    std::memcpy(iv.acc_0, sp.acc0, sizeof iv.acc_0), std::memcpy(iv.gyr_0, sp.gyr0, sizeof iv.gyr_0);
    std::memcpy(iv.linearized_ba, sp.lin_ba, sizeof iv.linearized_ba), std::memcpy(iv.linearized_bg, sp.lin_bg, sizeof iv.linearized_bg);
    in.push_back(iv), which.push_back(i);
  }
  if (in.empty()) return true;
  if (!device()) return false;
  std::vector<LfvioPreintegration> out(in.size());
  const Config &c = config();
  const double noise[4] = {c.acc_n, c.gyr_n, c.acc_w, c.gyr_w};
  status = lfvio_preintegrate(gpu, (int)in.size(), in.data(), noise, out.data());
  if (status != LFVIO_OK) return false;
  for (size_t k = 0; k < in.size(); k++) {
    ImuSpan &sp = span(which[k]);
    sp.pre = out[k];
    sp.sum_dt = out[k].sum_dt;
    sp.dirty = false;
  }
  return true;
}

void WindowEstimator::vector2double() {  // estimator.cpp:488-530
  for (int i = 0; i < FRAMES; i++) {
    const Keyframe &f = kf(i);
    const Quaterniond q{f.R};
    const double pose[7] = {f.P.x(), f.P.y(), f.P.z(), q.x(), q.y(), q.z(), q.w()};
    const double sb[9] = {f.V.x(), f.V.y(), f.V.z(), f.Ba.x(), f.Ba.y(), f.Ba.z(), f.Bg.x(), f.Bg.y(), f.Bg.z()};
    std::memcpy(para_Pose[i], pose, sizeof pose);
    std::memcpy(para_SpeedBias[i], sb, sizeof sb);
  }
  const Quaterniond qe{ric};
  const double ex[7] = {tic.x(), tic.y(), tic.z(), qe.x(), qe.y(), qe.z(), qe.w()};
  std::memcpy(para_Ex_Pose[0], ex, sizeof ex);
  para_Feature.clear();
  for (int s : tracks.order())
    if (tracks.solvable(s)) para_Feature.push_back(1. / tracks.depth(s));  // getDepthVector(), feature_manager.cpp:181-197
  if (config().estimate_td) para_Td[0][0] = td;
}

void WindowEstimator::double2vector() {  // estimator.cpp:532-600 (the relocalization tail :603-625 is dead as shipped)
  Vector3d origin_R0 = yawPitchRollDeg(kf(0).R);
  Vector3d origin_P0 = kf(0).P;
  if (failure_occur) {
    origin_R0 = yawPitchRollDeg(last_R0);
    origin_P0 = last_P0;
    failure_occur = false;
  }
  auto quat = [](const double *p) { return Quaterniond(p[6], p[3], p[4], p[5]); };
  const Vector3d origin_R00 = yawPitchRollDeg(quat(para_Pose[0]).toRotationMatrix());
  const double y_diff = origin_R0.x() - origin_R00.x();
  Matrix3d rot_diff = fromYawPitchRollDeg(Vector3d(y_diff, 0, 0));
  if (std::abs(std::abs(origin_R0.y()) - 90) < 1.0 || std::abs(std::abs(origin_R00.y()) - 90) < 1.0)
    rot_diff = kf(0).R * quat(para_Pose[0]).toRotationMatrix().transpose();  // singular pitch
  for (int i = 0; i < FRAMES; i++) {
    Keyframe &f = kf(i);
    const double *p = para_Pose[i], *s = para_SpeedBias[i];
    f.R = rot_diff * quat(p).normalized().toRotationMatrix();
    f.P = rot_diff * Vector3d(p[0] - para_Pose[0][0], p[1] - para_Pose[0][1], p[2] - para_Pose[0][2]) + origin_P0;
    f.V = rot_diff * Vector3d(s[0], s[1], s[2]);
    f.Ba = Vector3d(s[3], s[4], s[5]);
    f.Bg = Vector3d(s[6], s[7], s[8]);
  }
  tic = Vector3d(para_Ex_Pose[0][0], para_Ex_Pose[0][1], para_Ex_Pose[0][2]);
  ric = quat(para_Ex_Pose[0]).toRotationMatrix();
  size_t k = 0;
  for (int s : tracks.order()) {
    if (!tracks.solvable(s)) continue;
    tracks.setDepth(s, 1.0 / para_Feature[k++]);  // setDepth(), feature_manager.cpp:139-156
    tracks.setSolveFlag(s, 1);                     // both branches of the reference set 1
  }
  if (config().estimate_td) td = para_Td[0][0];
}

void WindowEstimator::pack(LfvioWindow *w) {
  const Config &c = config();
  if (!chain_upload_) (void)collectPrior();  // (optimization() on one device lets the upload collect it: lfvio_batch_upload_chained)
  std::memset(w, 0, sizeof *w);
  std::memcpy(w->para_pose, para_Pose, sizeof para_Pose);
  std::memcpy(w->para_speed_bias, para_SpeedBias, sizeof para_SpeedBias);
  std::memcpy(w->para_ex_pose, para_Ex_Pose[0], sizeof para_Ex_Pose[0]);
  w->para_td = c.estimate_td ? para_Td[0][0] : td;
  w->estimate_extrinsic = c.estimate_extrinsic != 0;
  w->estimate_td = c.estimate_td != 0;
  w->max_num_iterations = c.num_iterations;
  // estimator.cpp:819-822; <= 0 disables the cap (parity / bench)
  w->max_solver_time_in_seconds = c.solver_time <= 0 ? -1.0 : (marg_flag == LFVIO_MARGIN_OLD ? c.solver_time * 4.0 / 5.0 : c.solver_time);
  std::memcpy(w->g, c.gravity, sizeof w->g);
  w->tr = c.tr, w->row = c.row;
  w->sqrt_info = FOCAL_LENGTH / 1.5;  // estimator.cpp:18-19
  // the solvable tracks in table order -> CSR: one pass, sized once
  Staging &st = stage_;
  const int N = tracks.solvableCount();
  int M = 0;
  for (int s : tracks.order())
    if (tracks.solvable(s)) M += tracks.count(s);
  st.start_frame.resize(N), st.obs_offset.resize(N + 1), st.inv_depth.resize(N);
  st.point.resize(3 * (size_t)M), st.velocity.resize(3 * (size_t)M), st.cur_td.resize(M), st.uv_y.resize(M);
  int l = 0, o = 0;
  st.obs_offset[0] = 0;
  for (int s : tracks.order()) {
    if (!tracks.solvable(s)) continue;
    st.start_frame[l] = tracks.start(s);
    st.inv_depth[l] = para_Feature[l];
    for (int k = 0; k < tracks.count(s); k++, o++) {
      const double *r = tracks.obs(s, k);
      std::memcpy(&st.point[3 * (size_t)o], r, 3 * sizeof(double));
      std::memcpy(&st.velocity[3 * (size_t)o], r + 5, 3 * sizeof(double));
      st.uv_y[o] = r[4];
      st.cur_td[o] = r[8];
    }
    st.obs_offset[++l] = o;
  }
  w->num_landmarks = N, w->num_observations = M;
  w->start_frame = st.start_frame.data(), w->obs_offset = st.obs_offset.data(), w->inv_depth = st.inv_depth.data();
  w->obs_point = st.point.data(), w->obs_velocity = st.velocity.data(), w->obs_cur_td = st.cur_td.data(), w->obs_uv_y = st.uv_y.data();
  for (int i = 0; i < WINDOW_SIZE; i++) {  // pre_integrations[1..10] (estimator.cpp:717-724)
    const ImuSpan &sp = span(i + 1);
    if (sp.present) w->imu[i] = sp.pre;
    else w->imu[i].sum_dt = 1e9;  // no factor
  }
  w->prior = has_prior ? &prior : nullptr;
}

// estimator.cpp:676-1009 over the C-ABI.  On any error the state is left as the caller had it — but for a prior that was lost on the
// device with the failed call, see below — and `status` says why (the
// reference has no error channel at all).
void WindowEstimator::optimization() {
  status = LFVIO_OK;
  if (!device() || !refreshSpans(false)) return;
  // The marginalization behind the previous call's early state may still be running.  On one device with one upload the
  // window is packed and its tables are built while it finishes: the upload collects the prior itself (chained); every other
  // route waits for it here.
  const bool chain = prior_pending_ && fused && !group && config().split_call;
  if (!chain && !collectPrior()) return;
  Stopwatch sw(&timers[0]);
  timers[5] += 1.0;
  vector2double();  // :707
  LfvioWindow w;
  chain_upload_ = chain;
  pack(&w);
  chain_upload_ = false;
  Staging &st = stage_;
  st.lam_out.assign(w.num_landmarks > 0 ? w.num_landmarks : 1, 0.0);
  bool second_new = marg_flag == LFVIO_MARGIN_SECOND_NEW && has_prior && prior.valid;  // (chained upload: decided again once the prior is there)
  bool marginalize = marg_flag == LFVIO_MARGIN_OLD || second_new;
  auto take_state = [&] {
    std::memcpy(para_Pose, summary.para_pose, sizeof para_Pose);
    std::memcpy(para_SpeedBias, summary.para_speed_bias, sizeof para_SpeedBias);
    std::memcpy(para_Ex_Pose[0], summary.para_ex_pose, sizeof para_Ex_Pose[0]);
    if (config().estimate_td) para_Td[0][0] = summary.para_td;
    std::copy(st.lam_out.begin(), st.lam_out.begin() + w.num_landmarks, para_Feature.begin());
    double2vector();  // :830
  };
  LfvioPrior &next = next_;
  summary.inv_depth = st.lam_out.data();
  if (group) {
    // several devices: the same three steps as below in ONE call — the landmarks sharded over the devices of the group,
    // every collective an ncclAllReduce the library issues itself (lf-vio_amd/csrc/group.inc)
    status = lfvio_group_solve(group, &w, marg_flag, &summary, &next);
    summary.inv_depth = nullptr;
    if (status != LFVIO_OK) return;
    take_state();
    if (marginalize) assign_prior(&prior, &next), has_prior = next.valid != 0;
    return;
  }
  if (fused) {
    // one upload; solve (:810-825), the gauge fix of double2vector() (:532-626) and the marginalization (:833-1005) run back
    // to back on the device.  The state that comes back is already re-anchored, so double2vector() below applies a zero
    // yaw and a zero shift to it.
    status = lfvio_batch_reserve(gpu, 1, w.num_landmarks, w.num_observations);
    if (status == LFVIO_OK && chain) {
      // `prior` is an output here first (the prior of the call in flight), then the window's input
      w.prior = nullptr;
      // first choice: the prior stays on the device and nothing is waited for.  Refused with LFVIO_ERR_ARG — and the call in flight
      // untouched — when that marginalization passes its input prior through or the window is malformed: the host-side hand-over
      // then decides (it reports a malformed window itself).
      status = config().device_chain ? lfvio_batch_upload_chained_device(gpu, 0, &w) : LFVIO_ERR_ARG;
      if (status == LFVIO_OK) {
        // the window's prior is on the device only; the host's copy `prior` is the one of two windows back from here on, until the
        // next marginalization is collected
        prior_pending_ = false, has_prior = true, prior_on_device_ = true;
      } else {
        status = lfvio_batch_upload_chained(gpu, 0, &w, &prior);
        prior_pending_ = lfvio_batch_optimize_pending(gpu) != 0;  // (an upload refused before it got to the prior leaves it where it was)
        if (!prior_pending_) has_prior = prior.valid != 0;
      }
      if (status == LFVIO_OK) {
        second_new = marg_flag == LFVIO_MARGIN_SECOND_NEW && has_prior;
        marginalize = marg_flag == LFVIO_MARGIN_OLD || second_new;
      }
    } else if (status == LFVIO_OK)
      status = lfvio_batch_upload(gpu, 0, &w);
    if (status == LFVIO_OK && config().split_call) {
      // The state comes back as soon as solve + gauge fix are out (the device pushes it into mapped host memory); the
      // marginalization runs on while the caller publishes the pose, slides the window, takes the next image — its prior is
      // collected by the first thing that needs it (collectPrior(): the next pack()).
      status = lfvio_batch_optimize_begin(gpu, marg_flag, &summary);
      summary.inv_depth = nullptr;
      if (status != LFVIO_OK) {
        // A window that took its prior over on the device and did not get through (the prior was not there, a device error): that
        // prior is gone with the call, and the host's copy is two windows old — its frames have been shifted twice since.  The
        // estimator goes on WITHOUT a prior rather than with that one.
        if (prior_on_device_) has_prior = false, prior.valid = 0, prior_on_device_ = false;
        return;
      }
      take_state();
      if (marginalize) prior_pending_ = true;
      else (void)lfvio_batch_optimize_finish(gpu, nullptr);
      if (marginalize) prior_on_device_ = false;  // (what is collected next is this call's own prior)
      return;
    }
    if (status == LFVIO_OK) status = lfvio_batch_optimize(gpu, 1, marg_flag);
    if (status == LFVIO_OK) status = lfvio_batch_download(gpu, 0, &summary, marginalize ? &next : nullptr);
    summary.inv_depth = nullptr;
    if (status != LFVIO_OK) return;
    take_state();
    if (marginalize) assign_prior(&prior, &next), has_prior = next.valid != 0;
    return;
  }
  // the reference's literal sequence: solve, double2vector() on the host, vector2double(), marginalize (two uploads)
  status = lfvio_solve(gpu, &w, &summary);
  summary.inv_depth = nullptr;
  if (status != LFVIO_OK) return;
  take_state();
  if (!marginalize) return;
  vector2double();
  pack(&w);
  status = lfvio_marginalize(gpu, &w, marg_flag, &next);
  if (status != LFVIO_OK) return;
  assign_prior(&prior, &next), has_prior = next.valid != 0;
}

}  // namespace lfvio
