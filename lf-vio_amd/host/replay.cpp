// replay.cpp — see replay.h.  Reference lines are relative to /root/reference/vins_estimator/src.
#include "replay.h"

#include <cstdio>
#include <cstring>

namespace lfvio {

bool Trace::load(const char *path) {
  FILE *f = std::fopen(path, "rb");
  if (!f) {
    error = std::string("cannot open ") + path;
    return false;
  }
  char magic[4];
  uint32_t version = 0;
  if (std::fread(magic, 1, 4, f) != 4 || std::memcmp(magic, "LFVT", 4) != 0 || std::fread(&version, 4, 1, f) != 1 || version != 1) {
    error = "not an LFVT version 1 trace";
    std::fclose(f);
    return false;
  }
  std::vector<char> buf;
  for (;;) {
    uint32_t head[2];
    if (std::fread(head, 4, 2, f) != 2) break;  // end of file
    buf.resize(head[1]);
    if (head[1] && std::fread(buf.data(), 1, head[1], f) != head[1]) {
      error = "truncated record";
      std::fclose(f);
      return false;
    }
    const double *d = (const double *)buf.data();
    if (head[0] == 1 && head[1] == 7 * sizeof(double)) {
      imu.push_back({d[0], Vector3d(d[1], d[2], d[3]), Vector3d(d[4], d[5], d[6])});
    } else if (head[0] == 2 && head[1] >= 12) {
      uint32_t n;
      std::memcpy(&n, buf.data() + 8, 4);
      if (head[1] != 12 + (size_t)n * 36) {
        error = "feature record with a wrong length";
        std::fclose(f);
        return false;
      }
      TraceImage m;
      m.t = d[0];
      m.v.resize((size_t)n * 9);
      std::memcpy(m.v.data(), buf.data() + 12, (size_t)n * 36);
      images.push_back(std::move(m));
    } else if (head[0] == 3 && head[1] == (11 * 21 + 3 + 13) * sizeof(double)) {
      const int W = WINDOW_SIZE + 1;
      const double *Ps = d, *Rs = Ps + 3 * W, *Vs = Rs + 9 * W, *Bas = Vs + 3 * W, *Bgs = Bas + 3 * W, *gg = Bgs + 3 * W, *t = gg + 3, *r = t + 3;
      for (int i = 0; i < W; i++) {
        bootstrap.Ps[i] = Vector3d(Ps[3 * i], Ps[3 * i + 1], Ps[3 * i + 2]);
        bootstrap.Vs[i] = Vector3d(Vs[3 * i], Vs[3 * i + 1], Vs[3 * i + 2]);
        bootstrap.Bas[i] = Vector3d(Bas[3 * i], Bas[3 * i + 1], Bas[3 * i + 2]);
        bootstrap.Bgs[i] = Vector3d(Bgs[3 * i], Bgs[3 * i + 1], Bgs[3 * i + 2]);
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) bootstrap.Rs[i](a, b) = Rs[9 * i + 3 * a + b];
      }
      bootstrap.g = Vector3d(gg[0], gg[1], gg[2]);
      bootstrap.valid = true;
      tic = Vector3d(t[0], t[1], t[2]);
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) ric(a, b) = r[3 * a + b];
      td = r[9];
      has_bootstrap = true;
    }
  }
  std::fclose(f);
  return true;
}

ImageMap decodeFeatures(const TraceImage &msg) {  // estimator_node.cpp:292-312
  ImageMap image;
  for (size_t i = 0; i < msg.size(); i++) {
    const float *p = &msg.v[9 * i];
    int v = p[3] + 0.5;
    int feature_id = v / NUM_OF_CAM;
    int camera_id = v % NUM_OF_CAM;
    Vector8d xyz_uv_velocity;
    xyz_uv_velocity.a[0] = p[0], xyz_uv_velocity.a[1] = p[1], xyz_uv_velocity.a[2] = p[2];  // x y z
    xyz_uv_velocity.a[3] = p[4], xyz_uv_velocity.a[4] = p[5];                               // p_u p_v
    xyz_uv_velocity.a[5] = p[6], xyz_uv_velocity.a[6] = p[7], xyz_uv_velocity.a[7] = p[8];  // velocity x y z
    image[feature_id].emplace_back(camera_id, xyz_uv_velocity);
  }
  return image;
}

int replay(Estimator &estimator, const Trace &trace, const char *traj_path, int max_images, ReplayStats *stats) {
  ReplayStats st;
  std::memset(&st, 0, sizeof st);
  FILE *traj = nullptr;
  if (traj_path && traj_path[0]) {
    traj = std::fopen(traj_path, "w");
    if (!traj) return -1;
  }
  if (trace.has_bootstrap) {
    estimator.bootstrap = trace.bootstrap;
    estimator.tic[0] = trace.tic, estimator.ric[0] = trace.ric;
    estimator.td = trace.td;
  }
  const std::vector<TraceImu> &imu = trace.imu;
  size_t front = 0;  // imu_buf.front()
  double current_time = -1;
  double dx = 0, dy = 0, dz = 0, rx = 0, ry = 0, rz = 0;
  for (const TraceImage &img_msg : trace.images) {
    if (max_images > 0 && st.images >= max_images) break;
    // getMeasurements(), :96-134
    if (front >= imu.size()) break;
    if (!(imu.back().t > img_msg.t + estimator.td)) break;  // "wait for imu": nothing more will arrive in a recording
    if (!(imu[front].t < img_msg.t + estimator.td)) {       // "throw img, only should happen at the beginning"
      st.thrown++;
      continue;
    }
    size_t first = front;
    while (imu[front].t < img_msg.t + estimator.td) front++;
    // IMUs = [first, front] — imu_buf.front() is appended without being popped (:127) and comes again with the next image
    // process(), :218-262
    for (size_t k = first; k <= front; k++) {
      const TraceImu &imu_msg = imu[k];
      double t = imu_msg.t;
      double img_t = img_msg.t + estimator.td;
      if (t <= img_t) {
        if (current_time < 0) current_time = t;
        double dt = t - current_time;
        current_time = t;
        dx = imu_msg.acc.x(), dy = imu_msg.acc.y(), dz = imu_msg.acc.z();
        rx = imu_msg.gyr.x(), ry = imu_msg.gyr.y(), rz = imu_msg.gyr.z();
        estimator.processIMU(dt, Vector3d(dx, dy, dz), Vector3d(rx, ry, rz));
      } else {
        double dt_1 = img_t - current_time;
        double dt_2 = t - img_t;
        current_time = img_t;
        double w1 = dt_2 / (dt_1 + dt_2);
        double w2 = dt_1 / (dt_1 + dt_2);
        dx = w1 * dx + w2 * imu_msg.acc.x();
        dy = w1 * dy + w2 * imu_msg.acc.y();
        dz = w1 * dz + w2 * imu_msg.acc.z();
        rx = w1 * rx + w2 * imu_msg.gyr.x();
        ry = w1 * ry + w2 * imu_msg.gyr.y();
        rz = w1 * rz + w2 * imu_msg.gyr.z();
        estimator.processIMU(dt_1, Vector3d(dx, dy, dz), Vector3d(rx, ry, rz));
      }
    }
    ImageMap image = decodeFeatures(img_msg);
    const bool was_nonlinear = estimator.solver_flag == Estimator::NON_LINEAR;
    estimator.last_status = LFVIO_OK;
    estimator.processImage(image, img_msg.t);
    st.images++;
    if (estimator.last_status != LFVIO_OK || estimator.f_manager.last_status != LFVIO_OK) {
      st.last_status = estimator.last_status != LFVIO_OK ? estimator.last_status : estimator.f_manager.last_status;
      if (traj) std::fclose(traj);
      if (stats) *stats = st;
      return -2;  // the device call failed: no silent continuation
    }
    if (was_nonlinear && estimator.solver_flag == Estimator::INITIAL) st.failures++;  // failureDetection() -> clearState()
    if (estimator.solver_flag == Estimator::NON_LINEAR) {
      (estimator.marginalization_flag == Estimator::MARGIN_OLD ? st.keyframes : st.non_keyframes)++;
      st.iterations += estimator.last_summary.num_iterations;
      // pubOdometry(), utility/visualization.cpp:114-179: fixed, precision 12, "stamp x y z qx qy qz qw"
      Quaterniond tmp_Q = Quaterniond(estimator.Rs[WINDOW_SIZE]);
      const Vector3d &P = estimator.Ps[WINDOW_SIZE];
      if (traj)
        std::fprintf(traj, "%.12f %.12f %.12f %.12f %.12f %.12f %.12f %.12f\n", img_msg.t, P.x(), P.y(), P.z(), tmp_Q.x(), tmp_Q.y(), tmp_Q.z(),
                     tmp_Q.w());
      st.poses++;
    }
  }
  if (traj) std::fclose(traj);
  if (stats) *stats = st;
  return 0;
}

}  // namespace lfvio
