// estimator.cpp — host mirror (see estimator.h).  Reference lines are relative to
// /root/reference/vins_estimator/src.
#include "estimator.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

namespace lfvio {

double ACC_N = 0.02, ACC_W = 0.04, GYR_N = 0.01, GYR_W = 0.001;  // config/mindvision/mindvision.yaml:138-141
Vector3d G{0.0, 0.0, 9.81007};                                     // :142
double INIT_DEPTH = 5.0;                                           // parameters.cpp:116
double SOLVER_TIME = 0.04;                                         // :133
int NUM_ITERATIONS = 8;                                            // :134
int ESTIMATE_EXTRINSIC = 1, ESTIMATE_TD = 1;                       // :83, :151
double TD = -0.008, TR = 0.0, ROW = 960, COL = 1280;
double MIN_PARALLAX = 10.0 / FOCAL_LENGTH;                         // keyframe_parallax: 10.0, parameters.cpp:56-57

// ---------------------------------------------------------------- Utility (utility.h:66-113)
Vector3d Utility::R2ypr(const Matrix3d &R) {
  Vector3d n = R.col(0), o = R.col(1), a = R.col(2);
  double y = atan2(n(1), n(0));
  double p = atan2(-n(2), n(0) * cos(y) + n(1) * sin(y));
  double r = atan2(a(0) * sin(y) - a(1) * cos(y), -o(0) * sin(y) + o(1) * cos(y));
  return Vector3d(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);
}
Matrix3d Utility::ypr2R(const Vector3d &ypr) {
  double y = ypr(0) / 180.0 * M_PI, p = ypr(1) / 180.0 * M_PI, r = ypr(2) / 180.0 * M_PI;
  Matrix3d Rz, Ry, Rx;
  Rz(0, 0) = cos(y), Rz(0, 1) = -sin(y), Rz(1, 0) = sin(y), Rz(1, 1) = cos(y), Rz(2, 2) = 1;
  Ry(0, 0) = cos(p), Ry(0, 2) = sin(p), Ry(1, 1) = 1, Ry(2, 0) = -sin(p), Ry(2, 2) = cos(p);
  Rx(0, 0) = 1, Rx(1, 1) = cos(r), Rx(1, 2) = -sin(r), Rx(2, 1) = sin(r), Rx(2, 2) = cos(r);
  return Rz * Ry * Rx;
}

// ---------------------------------------------------------------- IntegrationBase
IntegrationBase::IntegrationBase(const Vector3d &_acc_0, const Vector3d &_gyr_0, const Vector3d &_linearized_ba,
                                 const Vector3d &_linearized_bg)
    : acc_0{_acc_0}, gyr_0{_gyr_0}, linearized_acc{_acc_0}, linearized_gyr{_gyr_0}, linearized_ba{_linearized_ba},
      linearized_bg{_linearized_bg}, sum_dt{0.0}, delta_q{Quaterniond::Identity()} {
  // integration_base.h:13-28
  jacobian.setIdentity();
  covariance.setZero();
  noise.setZero();
  const double nn[6] = {ACC_N * ACC_N, GYR_N * GYR_N, ACC_N * ACC_N, GYR_N * GYR_N, ACC_W * ACC_W, GYR_W * GYR_W};
  for (int b = 0; b < 6; b++)
    for (int i = 0; i < 3; i++) noise(3 * b + i, 3 * b + i) = nn[b];
}

void IntegrationBase::push_back(double _dt, const Vector3d &acc, const Vector3d &gyr) {  // :30-36
  dt_buf.push_back(_dt);
  acc_buf.push_back(acc);
  gyr_buf.push_back(gyr);
  propagate(_dt, acc, gyr);
}

void IntegrationBase::repropagate(const Vector3d &_linearized_ba, const Vector3d &_linearized_bg) {  // :38-52
  sum_dt = 0.0;
  acc_0 = linearized_acc;
  gyr_0 = linearized_gyr;
  delta_p.setZero();
  delta_q.setIdentity();
  delta_v.setZero();
  linearized_ba = _linearized_ba;
  linearized_bg = _linearized_bg;
  jacobian.setIdentity();
  covariance.setZero();
  for (int i = 0; i < static_cast<int>(dt_buf.size()); i++) propagate(dt_buf[i], acc_buf[i], gyr_buf[i]);
}

void IntegrationBase::midPointIntegration(double _dt, const Vector3d &_acc_0, const Vector3d &_gyr_0, const Vector3d &_acc_1,
                                          const Vector3d &_gyr_1, const Vector3d &delta_p, const Quaterniond &delta_q,
                                          const Vector3d &delta_v, const Vector3d &linearized_ba, const Vector3d &linearized_bg,
                                          Vector3d &result_delta_p, Quaterniond &result_delta_q, Vector3d &result_delta_v,
                                          Vector3d &result_linearized_ba, Vector3d &result_linearized_bg, bool update_jacobian) {
  // :63-71 — note: result_delta_q is NOT normalised here and is used as is below
  Vector3d un_acc_0 = delta_q * (_acc_0 - linearized_ba);
  Vector3d un_gyr = 0.5 * (_gyr_0 + _gyr_1) - linearized_bg;
  result_delta_q = delta_q * Quaterniond(1, un_gyr(0) * _dt / 2, un_gyr(1) * _dt / 2, un_gyr(2) * _dt / 2);
  Vector3d un_acc_1 = result_delta_q * (_acc_1 - linearized_ba);
  Vector3d un_acc = 0.5 * (un_acc_0 + un_acc_1);
  result_delta_p = delta_p + delta_v * _dt + 0.5 * un_acc * _dt * _dt;
  result_delta_v = delta_v + un_acc * _dt;
  result_linearized_ba = linearized_ba;
  result_linearized_bg = linearized_bg;
  if (!update_jacobian) return;
  // :75-125
  Vector3d w_x = 0.5 * (_gyr_0 + _gyr_1) - linearized_bg;
  Vector3d a_0_x = _acc_0 - linearized_ba, a_1_x = _acc_1 - linearized_ba;
  Matrix3d R_w_x = Utility::skewSymmetric(w_x), R_a_0_x = Utility::skewSymmetric(a_0_x), R_a_1_x = Utility::skewSymmetric(a_1_x);
  Matrix3d Rdq = delta_q.toRotationMatrix(), Rrdq = result_delta_q.toRotationMatrix();
  Matrix3d I = Matrix3d::Identity();
  Mat<15, 15> F;
  F.setBlock3(0, 0, I);
  F.setBlock3(0, 3, (-0.25 * Rdq * R_a_0_x * _dt * _dt) + (-0.25 * Rrdq * R_a_1_x * (I - R_w_x * _dt) * _dt * _dt));
  F.setBlock3(0, 6, I * _dt);
  F.setBlock3(0, 9, -0.25 * (Rdq + Rrdq) * _dt * _dt);
  F.setBlock3(0, 12, -0.25 * Rrdq * R_a_1_x * _dt * _dt * -_dt);
  F.setBlock3(3, 3, I - R_w_x * _dt);
  F.setBlock3(3, 12, -1.0 * I * _dt);
  F.setBlock3(6, 3, (-0.5 * Rdq * R_a_0_x * _dt) + (-0.5 * Rrdq * R_a_1_x * (I - R_w_x * _dt) * _dt));
  F.setBlock3(6, 6, I);
  F.setBlock3(6, 9, -0.5 * (Rdq + Rrdq) * _dt);
  F.setBlock3(6, 12, -0.5 * Rrdq * R_a_1_x * _dt * -_dt);
  F.setBlock3(9, 9, I);
  F.setBlock3(12, 12, I);
  Mat<15, 18> V;
  Matrix3d V03 = 0.25 * (-Rrdq) * R_a_1_x * _dt * _dt * 0.5 * _dt;
  Matrix3d V63 = 0.5 * (-Rrdq) * R_a_1_x * _dt * 0.5 * _dt;
  V.setBlock3(0, 0, 0.25 * Rdq * _dt * _dt);
  V.setBlock3(0, 3, V03);
  V.setBlock3(0, 6, 0.25 * Rrdq * _dt * _dt);
  V.setBlock3(0, 9, V03);
  V.setBlock3(3, 3, 0.5 * I * _dt);
  V.setBlock3(3, 9, 0.5 * I * _dt);
  V.setBlock3(6, 0, 0.5 * Rdq * _dt);
  V.setBlock3(6, 3, V63);
  V.setBlock3(6, 6, 0.5 * Rrdq * _dt);
  V.setBlock3(6, 9, V63);
  V.setBlock3(9, 12, I * _dt);
  V.setBlock3(12, 15, I * _dt);
  // jacobian = F * jacobian; covariance = F * covariance * F^T + V * noise * V^T
  Mat<15, 15> FJ, FC, NC;
  Mat<15, 18> VN;
  for (int i = 0; i < 15; i++)
    for (int j = 0; j < 15; j++) {
      double s = 0, c = 0;
      for (int k = 0; k < 15; k++) s += F(i, k) * jacobian(k, j), c += F(i, k) * covariance(k, j);
      FJ(i, j) = s, FC(i, j) = c;
    }
  for (int i = 0; i < 15; i++)
    for (int j = 0; j < 18; j++) {
      double s = 0;
      for (int k = 0; k < 18; k++) s += V(i, k) * noise(k, j);
      VN(i, j) = s;
    }
  for (int i = 0; i < 15; i++)
    for (int j = 0; j < 15; j++) {
      double s = 0, t = 0;
      for (int k = 0; k < 15; k++) s += FC(i, k) * F(j, k);
      for (int k = 0; k < 18; k++) t += VN(i, k) * V(j, k);
      NC(i, j) = s + t;
    }
  jacobian = FJ;
  covariance = NC;
}

void IntegrationBase::propagate(double _dt, const Vector3d &_acc_1, const Vector3d &_gyr_1) {  // :130-158
  dt = _dt;
  acc_1 = _acc_1;
  gyr_1 = _gyr_1;
  Vector3d result_delta_p, result_delta_v, result_linearized_ba, result_linearized_bg;
  Quaterniond result_delta_q;
  midPointIntegration(_dt, acc_0, gyr_0, _acc_1, _gyr_1, delta_p, delta_q, delta_v, linearized_ba, linearized_bg, result_delta_p,
                      result_delta_q, result_delta_v, result_linearized_ba, result_linearized_bg, 1);
  delta_p = result_delta_p;
  delta_q = result_delta_q;
  delta_v = result_delta_v;
  linearized_ba = result_linearized_ba;
  linearized_bg = result_linearized_bg;
  delta_q.normalize();
  sum_dt += dt;
  acc_0 = acc_1;
  gyr_0 = gyr_1;
}

// ---------------------------------------------------------------- FeatureManager (feature_manager.cpp:28-42,139-197)
int FeatureManager::getFeatureCount() {
  int cnt = 0;
  for (auto &it : feature) {
    it.used_num = it.feature_per_frame.size();
    if (it.used_num >= 2 && it.start_frame < WINDOW_SIZE - 2) cnt++;
  }
  return cnt;
}
void FeatureManager::setDepth(const VectorXd &x) {
  int feature_index = -1;
  for (auto &it_per_id : feature) {
    it_per_id.used_num = it_per_id.feature_per_frame.size();
    if (!(it_per_id.used_num >= 2 && it_per_id.start_frame < WINDOW_SIZE - 2)) continue;
    it_per_id.estimated_depth = 1.0 / x(++feature_index);
    it_per_id.solve_flag = 1;  // both branches of the reference set 1 (feature_manager.cpp:149-154)
  }
}
void FeatureManager::clearDepth(const VectorXd &x) {
  int feature_index = -1;
  for (auto &it_per_id : feature) {
    it_per_id.used_num = it_per_id.feature_per_frame.size();
    if (!(it_per_id.used_num >= 2 && it_per_id.start_frame < WINDOW_SIZE - 2)) continue;
    it_per_id.estimated_depth = 1.0 / x(++feature_index);
  }
}
void FeatureManager::removeFailures() {
  for (auto it = feature.begin(), it_next = feature.begin(); it != feature.end(); it = it_next) {
    it_next++;
    if (it->solve_flag == 2) feature.erase(it);
  }
}
VectorXd FeatureManager::getDepthVector() {
  VectorXd dep_vec(getFeatureCount());
  int feature_index = -1;
  for (auto &it_per_id : feature) {
    it_per_id.used_num = it_per_id.feature_per_frame.size();
    if (!(it_per_id.used_num >= 2 && it_per_id.start_frame < WINDOW_SIZE - 2)) continue;
    dep_vec(++feature_index) = 1. / it_per_id.estimated_depth;
  }
  return dep_vec;
}
// feature_manager.cpp:199-253.  The per-landmark 2k x 4 SVD runs on the device; the inclusion rule and the in-place
// update of estimated_depth are the reference's.
void FeatureManager::triangulate(Vector3d Ps[], Vector3d tic[], Matrix3d ric[]) {
  last_status = LFVIO_OK;
  std::vector<FeaturePerId *> sel;
  std::vector<int> start, off(1, 0);
  std::vector<double> pts, depth;
  for (auto &it_per_id : feature) {
    it_per_id.used_num = (int)it_per_id.feature_per_frame.size();
    if (!(it_per_id.used_num >= 2 && it_per_id.start_frame < WINDOW_SIZE - 2)) continue;
    if (it_per_id.estimated_depth > 0) continue;
    sel.push_back(&it_per_id);
    start.push_back(it_per_id.start_frame);
    for (auto &f : it_per_id.feature_per_frame) pts.push_back(f.point.x()), pts.push_back(f.point.y()), pts.push_back(f.point.z());
    off.push_back((int)pts.size() / 3);
    depth.push_back(it_per_id.estimated_depth);
  }
  if (sel.empty()) return;
  if (!gpu || !Rs) {
    last_status = LFVIO_ERR_ARG;
    return;
  }
  if (!*gpu) *gpu = lfvio_create(0);
  if (!*gpu) {
    last_status = LFVIO_ERR_DEVICE;  // no fallback
    return;
  }
  LfvioTriangulateIn in;
  in.num_landmarks = (int)sel.size(), in.num_observations = (int)pts.size() / 3;
  in.start_frame = start.data(), in.obs_offset = off.data(), in.obs_point = pts.data();
  for (int f = 0; f <= WINDOW_SIZE; f++)
    for (int i = 0; i < 3; i++) {
      in.Ps[f][i] = Ps[f](i);
      for (int j = 0; j < 3; j++) in.Rs[f][3 * i + j] = Rs[f](i, j);
    }
  for (int i = 0; i < 3; i++) {
    in.tic[i] = tic[0](i);
    for (int j = 0; j < 3; j++) in.ric[3 * i + j] = ric[0](i, j);
  }
  in.init_depth = INIT_DEPTH;
  last_status = lfvio_triangulate(*gpu, &in, depth.data());
  if (last_status != LFVIO_OK) return;
  for (size_t k = 0; k < sel.size(); k++) sel[k]->estimated_depth = depth[k];
}

// feature_manager.cpp:271-310
void FeatureManager::removeBackShiftDepth(Matrix3d marg_R, Vector3d marg_P, Matrix3d new_R, Vector3d new_P) {
  last_status = LFVIO_OK;
  std::vector<FeaturePerId *> sel;
  std::vector<double> uv, depth;
  for (auto it = feature.begin(), it_next = feature.begin(); it != feature.end(); it = it_next) {
    it_next++;
    if (it->start_frame != 0) {
      it->start_frame--;
    } else {
      const Vector3d uv_i = it->feature_per_frame[0].point;
      it->feature_per_frame.erase(it->feature_per_frame.begin());
      if (it->feature_per_frame.size() < 2) {
        feature.erase(it);
        continue;
      }
      sel.push_back(&*it);
      uv.push_back(uv_i.x()), uv.push_back(uv_i.y()), uv.push_back(uv_i.z());
      depth.push_back(it->estimated_depth);
    }
  }
  if (sel.empty()) return;
  if (!gpu) {
    last_status = LFVIO_ERR_ARG;
    return;
  }
  if (!*gpu) *gpu = lfvio_create(0);
  if (!*gpu) {
    last_status = LFVIO_ERR_DEVICE;
    return;
  }
  double mR[9], nR[9], mP[3], nP[3];
  for (int i = 0; i < 3; i++) {
    mP[i] = marg_P(i), nP[i] = new_P(i);
    for (int j = 0; j < 3; j++) mR[3 * i + j] = marg_R(i, j), nR[3 * i + j] = new_R(i, j);
  }
  last_status = lfvio_shift_depth(*gpu, (int)sel.size(), uv.data(), mR, mP, nR, nP, INIT_DEPTH, depth.data());
  if (last_status != LFVIO_OK) return;
  for (size_t k = 0; k < sel.size(); k++) sel[k]->estimated_depth = depth[k];
}

// feature_manager.cpp:312-330
void FeatureManager::removeBack() {
  for (auto it = feature.begin(), it_next = feature.begin(); it != feature.end(); it = it_next) {
    it_next++;
    if (it->start_frame != 0) {
      it->start_frame--;
    } else {
      it->feature_per_frame.erase(it->feature_per_frame.begin());
      if (it->feature_per_frame.size() == 0) feature.erase(it);
    }
  }
}

// feature_manager.cpp:332-351
void FeatureManager::removeFront(int frame_count) {
  for (auto it = feature.begin(), it_next = feature.begin(); it != feature.end(); it = it_next) {
    it_next++;
    if (it->start_frame == frame_count) {
      it->start_frame--;
    } else {
      const int j = WINDOW_SIZE - 1 - it->start_frame;
      if (it->endFrame() < frame_count - 1) continue;
      it->feature_per_frame.erase(it->feature_per_frame.begin() + j);
      if (it->feature_per_frame.size() == 0) feature.erase(it);
    }
  }
}

// feature_manager.cpp:45-95 — appends the frame's observations and decides keyframe (true -> MARGIN_OLD) or not
bool FeatureManager::addFeatureCheckParallax(int frame_count, const ImageMap &image, double td) {
  double parallax_sum = 0;
  int parallax_num = 0;
  last_track_num = 0;
  for (auto &id_pts : image) {
    FeaturePerFrame f_per_fra(id_pts.second[0].second.a, td);
    int feature_id = id_pts.first;
    auto it = std::find_if(feature.begin(), feature.end(), [feature_id](const FeaturePerId &it) { return it.feature_id == feature_id; });
    if (it == feature.end()) {
      feature.push_back(FeaturePerId(feature_id, frame_count));
      feature.back().feature_per_frame.push_back(f_per_fra);
    } else if (it->feature_id == feature_id) {
      it->feature_per_frame.push_back(f_per_fra);
      last_track_num++;
    }
  }
  if (frame_count < 2 || last_track_num < 20) return true;
  for (auto &it_per_id : feature) {
    if (it_per_id.start_frame <= frame_count - 2 && it_per_id.start_frame + int(it_per_id.feature_per_frame.size()) - 1 >= frame_count - 1) {
      parallax_sum += compensatedParallax2(it_per_id, frame_count);
      parallax_num++;
    }
  }
  if (parallax_num == 0) return true;
  return parallax_sum / parallax_num >= MIN_PARALLAX;
}
// feature_manager.cpp:353-369 — the angle between the bearings of the second and third last frame (x 10), uncompensated
double FeatureManager::compensatedParallax2(const FeaturePerId &it_per_id, int frame_count) {
  const FeaturePerFrame &frame_i = it_per_id.feature_per_frame[frame_count - 2 - it_per_id.start_frame];
  const FeaturePerFrame &frame_j = it_per_id.feature_per_frame[frame_count - 1 - it_per_id.start_frame];
  Vector3d p_j = frame_j.point;
  Vector3d p_i = frame_i.point;
  double p_i_comp = p_i.dot(p_j);
  return acos(p_i_comp) * 10;
}

FeaturePerId &FeatureManager::addFeature(int feature_id, int start_frame) {
  feature.emplace_back(feature_id, start_frame);
  return feature.back();
}

// ---------------------------------------------------------------- Estimator
Estimator::Estimator() {
  for (int i = 0; i <= WINDOW_SIZE; i++) pre_integrations[i] = nullptr;
  f_manager.Rs = Rs, f_manager.gpu = &gpu;  // FeatureManager f_manager{Rs} (estimator.cpp:6)
  clearState();
}
Estimator::~Estimator() {
  for (int i = 0; i <= WINDOW_SIZE; i++) delete pre_integrations[i];
  delete last_marginalization_info;
  if (gpu) lfvio_destroy(gpu);
}
void Estimator::setParameter() {  // estimator.cpp:10-21 (the sqrt_info statics travel in LfvioWindow::sqrt_info)
  td = TD;
}
void Estimator::clearState() {  // estimator.cpp:23-84 (subset owned by the mirror)
  for (int i = 0; i < WINDOW_SIZE + 1; i++) {
    Rs[i].setIdentity();
    Ps[i].setZero();
    Vs[i].setZero();
    Bas[i].setZero();
    Bgs[i].setZero();
    delete pre_integrations[i];
    pre_integrations[i] = nullptr;
  }
  for (int i = 0; i < NUM_OF_CAM; i++) {
    tic[i] = Vector3d::Zero();
    ric[i] = Matrix3d::Identity();
  }
  td = TD;
  for (int i = 0; i < WINDOW_SIZE + 1; i++) {
    dt_buf[i].clear();
    linear_acceleration_buf[i].clear();
    angular_velocity_buf[i].clear();
    Headers[i] = 0;
  }
  solver_flag = INITIAL;
  first_imu = false;
  sum_of_back = 0;
  sum_of_front = 0;
  frame_count = 0;
  initial_timestamp = 0;
  delete last_marginalization_info;
  last_marginalization_info = nullptr;
  f_manager.clearState();
  failure_occur = 0;
}

// estimator.cpp:86-120
void Estimator::processIMU(double dt, const Vector3d &linear_acceleration, const Vector3d &angular_velocity) {
  if (!first_imu) {
    first_imu = true;
    acc_0 = linear_acceleration;
    gyr_0 = angular_velocity;
  }
  if (!pre_integrations[frame_count]) pre_integrations[frame_count] = new IntegrationBase{acc_0, gyr_0, Bas[frame_count], Bgs[frame_count]};
  if (frame_count != 0) {
    pre_integrations[frame_count]->push_back(dt, linear_acceleration, angular_velocity);
    // tmp_pre_integration (:100) only feeds all_image_frame, i.e. initialStructure(): not kept
    dt_buf[frame_count].push_back(dt);
    linear_acceleration_buf[frame_count].push_back(linear_acceleration);
    angular_velocity_buf[frame_count].push_back(angular_velocity);
    int j = frame_count;
    Vector3d un_acc_0 = Rs[j] * (acc_0 - Bas[j]) - g;
    Vector3d un_gyr = 0.5 * (gyr_0 + angular_velocity) - Bgs[j];
    Rs[j] = Rs[j] * Utility::deltaQ(un_gyr * dt).toRotationMatrix();
    Vector3d un_acc_1 = Rs[j] * (linear_acceleration - Bas[j]) - g;
    Vector3d un_acc = 0.5 * (un_acc_0 + un_acc_1);
    Ps[j] += dt * Vs[j] + 0.5 * dt * dt * un_acc;
    Vs[j] += dt * un_acc;
  }
  acc_0 = linear_acceleration;
  gyr_0 = angular_velocity;
}

// estimator.cpp:122-220 without the ROS logging, all_image_frame / tmp_pre_integration (inputs of initialStructure) and
// the ESTIMATE_EXTRINSIC == 2 rotation calibration (init-only)
void Estimator::processImage(const ImageMap &image, double header_stamp) {
  if (f_manager.addFeatureCheckParallax(frame_count, image, td))
    marginalization_flag = MARGIN_OLD;
  else
    marginalization_flag = MARGIN_SECOND_NEW;
  Headers[frame_count] = header_stamp;
  if (solver_flag == INITIAL) {
    if (frame_count == WINDOW_SIZE) {
      bool result = false;
      if (ESTIMATE_EXTRINSIC != 2 && (header_stamp - initial_timestamp) > 0.1) {
        result = initialStructure();
        initial_timestamp = header_stamp;
      }
      if (result) {
        solver_flag = NON_LINEAR;
        solveOdometry();
        slideWindow();
        f_manager.removeFailures();
        last_R = Rs[WINDOW_SIZE];
        last_P = Ps[WINDOW_SIZE];
        last_R0 = Rs[0];
        last_P0 = Ps[0];
      } else
        slideWindow();
    } else
      frame_count++;
  } else {
    solveOdometry();
    if (failureDetection()) {
      failure_occur = 1;
      clearState();
      setParameter();
      return;
    }
    slideWindow();
    f_manager.removeFailures();
    key_poses.clear();
    for (int i = 0; i <= WINDOW_SIZE; i++) key_poses.push_back(Ps[i]);
    last_R = Rs[WINDOW_SIZE];
    last_P = Ps[WINDOW_SIZE];
    last_R0 = Rs[0];
    last_P0 = Ps[0];
  }
}

// Stand-in for initialStructure() + visualInitialAlign() (estimator.cpp:222-473): the window state comes from the
// `bootstrap` record; what visualInitialAlign does with it afterwards is kept — every interval is re-propagated with its
// new gyroscope bias and a zero accelerometer bias (:403-406; here on the device, one call) and g is taken over (:445).
bool Estimator::initialStructure() {
  if (!bootstrap.valid) return false;
  for (int i = 0; i <= WINDOW_SIZE; i++) {
    Ps[i] = bootstrap.Ps[i], Rs[i] = bootstrap.Rs[i], Vs[i] = bootstrap.Vs[i];
    Bas[i] = bootstrap.Bas[i], Bgs[i] = bootstrap.Bgs[i];
  }
  g = bootstrap.g;
  Vector3d zero[(WINDOW_SIZE + 1)];
  repropagateWindow(zero, Bgs);
  if (last_status != LFVIO_OK) return false;
  for (auto &it : f_manager.feature) it.estimated_depth = -1;  // clearDepth(-1), :386-390; triangulated in solveOdometry()
  return true;
}

// estimator.cpp:475-486
void Estimator::solveOdometry() {
  if (frame_count < WINDOW_SIZE) return;
  if (solver_flag == NON_LINEAR) {
    f_manager.triangulate(Ps, tic, ric);
    optimization();
  }
}

// estimator.cpp:628-674 — as shipped: only the gyroscope-bias and the two translation tests return true
bool Estimator::failureDetection() {
  if (Bgs[WINDOW_SIZE].norm() > 1.0) return true;
  Vector3d tmp_P = Ps[WINDOW_SIZE];
  if ((tmp_P - last_P).norm() > 5) return true;
  if (std::abs(tmp_P.z() - last_P.z()) > 1) return true;
  return false;
}

// estimator.cpp:1011-1131
void Estimator::slideWindow() {
  if (marginalization_flag == MARGIN_OLD) {
    back_R0 = Rs[0];
    back_P0 = Ps[0];
    if (frame_count == WINDOW_SIZE) {
      for (int i = 0; i < WINDOW_SIZE; i++) {
        std::swap(Rs[i], Rs[i + 1]);
        std::swap(pre_integrations[i], pre_integrations[i + 1]);
        dt_buf[i].swap(dt_buf[i + 1]);
        linear_acceleration_buf[i].swap(linear_acceleration_buf[i + 1]);
        angular_velocity_buf[i].swap(angular_velocity_buf[i + 1]);
        Headers[i] = Headers[i + 1];
        std::swap(Ps[i], Ps[i + 1]);
        std::swap(Vs[i], Vs[i + 1]);
        std::swap(Bas[i], Bas[i + 1]);
        std::swap(Bgs[i], Bgs[i + 1]);
      }
      Headers[WINDOW_SIZE] = Headers[WINDOW_SIZE - 1];
      Ps[WINDOW_SIZE] = Ps[WINDOW_SIZE - 1];
      Vs[WINDOW_SIZE] = Vs[WINDOW_SIZE - 1];
      Rs[WINDOW_SIZE] = Rs[WINDOW_SIZE - 1];
      Bas[WINDOW_SIZE] = Bas[WINDOW_SIZE - 1];
      Bgs[WINDOW_SIZE] = Bgs[WINDOW_SIZE - 1];
      delete pre_integrations[WINDOW_SIZE];
      pre_integrations[WINDOW_SIZE] = new IntegrationBase{acc_0, gyr_0, Bas[WINDOW_SIZE], Bgs[WINDOW_SIZE]};
      dt_buf[WINDOW_SIZE].clear();
      linear_acceleration_buf[WINDOW_SIZE].clear();
      angular_velocity_buf[WINDOW_SIZE].clear();
      slideWindowOld();
    }
  } else {
    if (frame_count == WINDOW_SIZE) {
      for (unsigned int i = 0; i < dt_buf[frame_count].size(); i++) {
        double tmp_dt = dt_buf[frame_count][i];
        Vector3d tmp_linear_acceleration = linear_acceleration_buf[frame_count][i];
        Vector3d tmp_angular_velocity = angular_velocity_buf[frame_count][i];
        pre_integrations[frame_count - 1]->push_back(tmp_dt, tmp_linear_acceleration, tmp_angular_velocity);
        dt_buf[frame_count - 1].push_back(tmp_dt);
        linear_acceleration_buf[frame_count - 1].push_back(tmp_linear_acceleration);
        angular_velocity_buf[frame_count - 1].push_back(tmp_angular_velocity);
      }
      Headers[frame_count - 1] = Headers[frame_count];
      Ps[frame_count - 1] = Ps[frame_count];
      Vs[frame_count - 1] = Vs[frame_count];
      Rs[frame_count - 1] = Rs[frame_count];
      Bas[frame_count - 1] = Bas[frame_count];
      Bgs[frame_count - 1] = Bgs[frame_count];
      delete pre_integrations[WINDOW_SIZE];
      pre_integrations[WINDOW_SIZE] = new IntegrationBase{acc_0, gyr_0, Bas[WINDOW_SIZE], Bgs[WINDOW_SIZE]};
      dt_buf[WINDOW_SIZE].clear();
      linear_acceleration_buf[WINDOW_SIZE].clear();
      angular_velocity_buf[WINDOW_SIZE].clear();
      slideWindowNew();
    }
  }
}
void Estimator::slideWindowNew() {
  sum_of_front++;
  f_manager.removeFront(frame_count);
}
void Estimator::slideWindowOld() {
  sum_of_back++;
  bool shift_depth = solver_flag == NON_LINEAR ? true : false;
  if (shift_depth) {
    Matrix3d R0, R1;
    Vector3d P0, P1;
    R0 = back_R0 * ric[0];
    R1 = Rs[0] * ric[0];
    P0 = back_P0 + back_R0 * tic[0];
    P1 = Ps[0] + Rs[0] * tic[0];
    f_manager.removeBackShiftDepth(R0, P0, R1, P1);
  } else
    f_manager.removeBack();
}

void Estimator::vector2double() {  // estimator.cpp:488-530
  for (int i = 0; i <= WINDOW_SIZE; i++) {
    para_Pose[i][0] = Ps[i].x();
    para_Pose[i][1] = Ps[i].y();
    para_Pose[i][2] = Ps[i].z();
    Quaterniond q{Rs[i]};
    para_Pose[i][3] = q.x();
    para_Pose[i][4] = q.y();
    para_Pose[i][5] = q.z();
    para_Pose[i][6] = q.w();
    para_SpeedBias[i][0] = Vs[i].x();
    para_SpeedBias[i][1] = Vs[i].y();
    para_SpeedBias[i][2] = Vs[i].z();
    para_SpeedBias[i][3] = Bas[i].x();
    para_SpeedBias[i][4] = Bas[i].y();
    para_SpeedBias[i][5] = Bas[i].z();
    para_SpeedBias[i][6] = Bgs[i].x();
    para_SpeedBias[i][7] = Bgs[i].y();
    para_SpeedBias[i][8] = Bgs[i].z();
  }
  for (int i = 0; i < NUM_OF_CAM; i++) {
    para_Ex_Pose[i][0] = tic[i].x();
    para_Ex_Pose[i][1] = tic[i].y();
    para_Ex_Pose[i][2] = tic[i].z();
    Quaterniond q{ric[i]};
    para_Ex_Pose[i][3] = q.x();
    para_Ex_Pose[i][4] = q.y();
    para_Ex_Pose[i][5] = q.z();
    para_Ex_Pose[i][6] = q.w();
  }
  VectorXd dep = f_manager.getDepthVector();
  para_Feature.resize(f_manager.getFeatureCount());
  for (int i = 0; i < f_manager.getFeatureCount(); i++) para_Feature[i] = dep(i);
  if (ESTIMATE_TD) para_Td[0][0] = td;
}

void Estimator::double2vector() {  // estimator.cpp:532-600 (relocalization tail :603-625 is a dead branch as shipped)
  Vector3d origin_R0 = Utility::R2ypr(Rs[0]);
  Vector3d origin_P0 = Ps[0];
  if (failure_occur) {
    origin_R0 = Utility::R2ypr(last_R0);
    origin_P0 = last_P0;
    failure_occur = 0;
  }
  Vector3d origin_R00 =
      Utility::R2ypr(Quaterniond(para_Pose[0][6], para_Pose[0][3], para_Pose[0][4], para_Pose[0][5]).toRotationMatrix());
  double y_diff = origin_R0.x() - origin_R00.x();
  Matrix3d rot_diff = Utility::ypr2R(Vector3d(y_diff, 0, 0));
  if (std::abs(std::abs(origin_R0.y()) - 90) < 1.0 || std::abs(std::abs(origin_R00.y()) - 90) < 1.0) {
    rot_diff = Rs[0] * Quaterniond(para_Pose[0][6], para_Pose[0][3], para_Pose[0][4], para_Pose[0][5]).toRotationMatrix().transpose();
  }
  for (int i = 0; i <= WINDOW_SIZE; i++) {
    Rs[i] = rot_diff * Quaterniond(para_Pose[i][6], para_Pose[i][3], para_Pose[i][4], para_Pose[i][5]).normalized().toRotationMatrix();
    Ps[i] = rot_diff * Vector3d(para_Pose[i][0] - para_Pose[0][0], para_Pose[i][1] - para_Pose[0][1], para_Pose[i][2] - para_Pose[0][2]) +
            origin_P0;
    Vs[i] = rot_diff * Vector3d(para_SpeedBias[i][0], para_SpeedBias[i][1], para_SpeedBias[i][2]);
    Bas[i] = Vector3d(para_SpeedBias[i][3], para_SpeedBias[i][4], para_SpeedBias[i][5]);
    Bgs[i] = Vector3d(para_SpeedBias[i][6], para_SpeedBias[i][7], para_SpeedBias[i][8]);
  }
  for (int i = 0; i < NUM_OF_CAM; i++) {
    tic[i] = Vector3d(para_Ex_Pose[i][0], para_Ex_Pose[i][1], para_Ex_Pose[i][2]);
    ric[i] = Quaterniond(para_Ex_Pose[i][6], para_Ex_Pose[i][3], para_Ex_Pose[i][4], para_Ex_Pose[i][5]).toRotationMatrix();
  }
  VectorXd dep = f_manager.getDepthVector();
  for (int i = 0; i < f_manager.getFeatureCount(); i++) dep(i) = para_Feature[i];
  f_manager.setDepth(dep);
  if (ESTIMATE_TD) td = para_Td[0][0];
}

void Estimator::packWindow(LfvioWindow *w) {
  std::memset(w, 0, sizeof *w);
  std::memcpy(w->para_pose, para_Pose, sizeof para_Pose);
  std::memcpy(w->para_speed_bias, para_SpeedBias, sizeof para_SpeedBias);
  std::memcpy(w->para_ex_pose, para_Ex_Pose[0], sizeof para_Ex_Pose[0]);
  w->para_td = ESTIMATE_TD ? para_Td[0][0] : td;
  w->estimate_extrinsic = ESTIMATE_EXTRINSIC != 0;
  w->estimate_td = ESTIMATE_TD != 0;
  w->max_num_iterations = NUM_ITERATIONS;
  // estimator.cpp:819-822; <= 0 disables the cap (parity / bench)
  w->max_solver_time_in_seconds = SOLVER_TIME <= 0 ? -1.0 : (marginalization_flag == MARGIN_OLD ? SOLVER_TIME * 4.0 / 5.0 : SOLVER_TIME);
  w->g[0] = G.x(), w->g[1] = G.y(), w->g[2] = G.z();
  w->tr = TR, w->row = ROW;
  w->sqrt_info = FOCAL_LENGTH / 1.5;  // estimator.cpp:18-19
  // features: same filter and order as the loops at estimator.cpp:727-772
  Scratch &s = scratch_;
  s.start_frame.clear(), s.obs_offset.assign(1, 0), s.inv_depth.clear();
  s.point.clear(), s.velocity.clear(), s.cur_td.clear(), s.uv_y.clear();
  int feature_index = -1;
  for (auto &it_per_id : f_manager.feature) {
    it_per_id.used_num = it_per_id.feature_per_frame.size();
    if (!(it_per_id.used_num >= 2 && it_per_id.start_frame < WINDOW_SIZE - 2)) continue;
    ++feature_index;
    s.start_frame.push_back(it_per_id.start_frame);
    s.inv_depth.push_back(para_Feature[feature_index]);
    for (auto &it_per_frame : it_per_id.feature_per_frame) {
      for (int k = 0; k < 3; k++) s.point.push_back(it_per_frame.point(k)), s.velocity.push_back(it_per_frame.velocity(k));
      s.cur_td.push_back(it_per_frame.cur_td);
      s.uv_y.push_back(it_per_frame.uv.y());
    }
    s.obs_offset.push_back((int)s.cur_td.size());
  }
  w->num_landmarks = (int)s.start_frame.size();
  w->num_observations = (int)s.cur_td.size();
  w->start_frame = s.start_frame.data(), w->obs_offset = s.obs_offset.data(), w->inv_depth = s.inv_depth.data();
  w->obs_point = s.point.data(), w->obs_velocity = s.velocity.data(), w->obs_cur_td = s.cur_td.data(), w->obs_uv_y = s.uv_y.data();
  // pre_integrations[1..10] (estimator.cpp:717-724)
  for (int i = 0; i < WINDOW_SIZE; i++) {
    const IntegrationBase *p = pre_integrations[i + 1];
    LfvioPreintegration &o = w->imu[i];
    if (!p) {
      o.sum_dt = 1e9;  // no factor
      continue;
    }
    o.sum_dt = p->sum_dt;
    for (int k = 0; k < 3; k++) {
      o.delta_p[k] = p->delta_p(k), o.delta_v[k] = p->delta_v(k);
      o.linearized_ba[k] = p->linearized_ba(k), o.linearized_bg[k] = p->linearized_bg(k);
    }
    o.delta_q[0] = p->delta_q.x(), o.delta_q[1] = p->delta_q.y(), o.delta_q[2] = p->delta_q.z(), o.delta_q[3] = p->delta_q.w();
    std::memcpy(o.jacobian, p->jacobian.a, sizeof o.jacobian);
    std::memcpy(o.covariance, p->covariance.a, sizeof o.covariance);
  }
  w->prior = last_marginalization_info;
}

// `for (i <= WINDOW_SIZE) pre_integrations[i]->repropagate(ba[i], bg[i])` (estimator.cpp:403-406; IntegrationBase::repropagate,
// integration_base.h:38-52) with the propagation of all intervals on the device: each IntegrationBase ends in the state
// its own repropagate() would leave (delta_*, jacobian, covariance, sum_dt, acc_0 / gyr_0 = the last buffered sample).
void Estimator::repropagateWindow(const Vector3d ba[(WINDOW_SIZE + 1)], const Vector3d bg[(WINDOW_SIZE + 1)]) {
  if (!gpu) gpu = lfvio_create(0);
  if (!gpu) {
    last_status = LFVIO_ERR_DEVICE;  // no fallback
    return;
  }
  std::vector<LfvioImuInterval> in;
  std::vector<int> which;
  std::vector<std::vector<double>> acc, gyr;
  for (int i = 0; i <= WINDOW_SIZE; i++) {
    IntegrationBase *p = pre_integrations[i];
    if (!p) continue;
    const size_t n = p->dt_buf.size();
    acc.emplace_back(3 * n), gyr.emplace_back(3 * n);
    for (size_t k = 0; k < n; k++)
      for (int d = 0; d < 3; d++) acc.back()[3 * k + d] = p->acc_buf[k](d), gyr.back()[3 * k + d] = p->gyr_buf[k](d);
    LfvioImuInterval iv;
    iv.num_samples = (int)n;
    iv.dt = p->dt_buf.data();
    for (int d = 0; d < 3; d++)
      iv.acc_0[d] = p->linearized_acc(d), iv.gyr_0[d] = p->linearized_gyr(d), iv.linearized_ba[d] = ba[i](d), iv.linearized_bg[d] = bg[i](d);
    in.push_back(iv), which.push_back(i);
  }
  for (size_t k = 0; k < in.size(); k++) in[k].acc = acc[k].data(), in[k].gyr = gyr[k].data();
  std::vector<LfvioPreintegration> out(in.size());
  const double noise[4] = {ACC_N, GYR_N, ACC_W, GYR_W};
  last_status = lfvio_preintegrate(gpu, (int)in.size(), in.data(), noise, out.data());
  if (last_status != LFVIO_OK) return;
  for (size_t k = 0; k < in.size(); k++) {
    IntegrationBase *p = pre_integrations[which[k]];
    const LfvioPreintegration &o = out[k];
    p->sum_dt = o.sum_dt;
    p->delta_p = Vector3d(o.delta_p[0], o.delta_p[1], o.delta_p[2]);
    p->delta_q = Quaterniond(o.delta_q[3], o.delta_q[0], o.delta_q[1], o.delta_q[2]);
    p->delta_v = Vector3d(o.delta_v[0], o.delta_v[1], o.delta_v[2]);
    p->linearized_ba = ba[which[k]], p->linearized_bg = bg[which[k]];
    for (int r = 0; r < 15; r++)
      for (int c = 0; c < 15; c++) p->jacobian(r, c) = o.jacobian[15 * r + c], p->covariance(r, c) = o.covariance[15 * r + c];
    const size_t n = p->dt_buf.size();
    p->acc_0 = n ? p->acc_buf[n - 1] : p->linearized_acc, p->gyr_0 = n ? p->gyr_buf[n - 1] : p->linearized_gyr;
    if (n) p->dt = p->dt_buf[n - 1], p->acc_1 = p->acc_buf[n - 1], p->gyr_1 = p->gyr_buf[n - 1];
  }
}

// Estimator::optimization(), estimator.cpp:676-1009, over the C-ABI:
//   vector2double -> [ceres::Solve := lfvio_solve] -> double2vector
//   -> vector2double -> [MarginalizationInfo := lfvio_marginalize] -> new prior
// On any error the state is left as the caller had it (the reference has no error channel at all).
void Estimator::optimization() {
  if (!gpu) gpu = lfvio_create(0);
  if (!gpu) {
    last_status = LFVIO_ERR_DEVICE;  // no fallback: the caller sees the failure
    return;
  }
  vector2double();  // :707
  LfvioWindow w;
  packWindow(&w);
  std::vector<double> lam(w.num_landmarks > 0 ? w.num_landmarks : 1);
  const bool second_new_needed =
      marginalization_flag == MARGIN_SECOND_NEW && last_marginalization_info && last_marginalization_info->valid;
  const bool marginalize = marginalization_flag == MARGIN_OLD || second_new_needed;
  if (fused) {
    // One upload: solve (:810-825), the gauge fix of double2vector() (:532-626) and the marginalization (:833-1005) run
    // back to back on the device; the state that comes back is the one double2vector() would have produced, so the call
    // below re-applies it to an already re-anchored state (a rotation by zero yaw and a zero shift).
    const int flag = marginalization_flag == MARGIN_OLD ? LFVIO_MARGIN_OLD : LFVIO_MARGIN_SECOND_NEW;
    LfvioPrior *next = marginalize ? new LfvioPrior() : nullptr;
    last_summary.inv_depth = lam.data();
    last_status = lfvio_batch_reserve(gpu, 1, w.num_landmarks, w.num_observations);
    if (last_status == LFVIO_OK) last_status = lfvio_batch_upload(gpu, 0, &w);
    if (last_status == LFVIO_OK) last_status = lfvio_batch_optimize(gpu, 1, flag);
    if (last_status == LFVIO_OK) last_status = lfvio_batch_download(gpu, 0, &last_summary, next);
    last_summary.inv_depth = nullptr;
    if (last_status != LFVIO_OK) {
      delete next;
      return;
    }
    std::memcpy(para_Pose, last_summary.para_pose, sizeof para_Pose);
    std::memcpy(para_SpeedBias, last_summary.para_speed_bias, sizeof para_SpeedBias);
    std::memcpy(para_Ex_Pose[0], last_summary.para_ex_pose, sizeof para_Ex_Pose[0]);
    if (ESTIMATE_TD) para_Td[0][0] = last_summary.para_td;
    for (int i = 0; i < w.num_landmarks; i++) para_Feature[i] = lam[i];
    double2vector();  // :830
    if (next) {
      delete last_marginalization_info;  // :935-938
      last_marginalization_info = next;
    }
    return;
  }
  // The literal two-call flow of the reference: solve, double2vector() on the host, vector2double(), marginalize.
  last_summary.inv_depth = lam.data();
  last_status = lfvio_solve(gpu, &w, &last_summary);  // :810-825
  last_summary.inv_depth = nullptr;
  if (last_status != LFVIO_OK) return;
  std::memcpy(para_Pose, last_summary.para_pose, sizeof para_Pose);
  std::memcpy(para_SpeedBias, last_summary.para_speed_bias, sizeof para_SpeedBias);
  std::memcpy(para_Ex_Pose[0], last_summary.para_ex_pose, sizeof para_Ex_Pose[0]);
  if (ESTIMATE_TD) para_Td[0][0] = last_summary.para_td;
  for (int i = 0; i < w.num_landmarks; i++) para_Feature[i] = lam[i];
  double2vector();  // :830

  // :833-1005 — both branches start with vector2double() and differ only in the factor set, which the
  // library derives from the flag
  if (marginalize) {
    vector2double();
    packWindow(&w);
    LfvioPrior *next = new LfvioPrior();
    last_status = lfvio_marginalize(gpu, &w, marginalization_flag == MARGIN_OLD ? LFVIO_MARGIN_OLD : LFVIO_MARGIN_SECOND_NEW, next);
    if (last_status != LFVIO_OK) {
      delete next;
      return;
    }
    delete last_marginalization_info;  // :935-938
    last_marginalization_info = next;
  }
}

}  // namespace lfvio
