// replay.cpp — see replay.h.  Reference lines are relative to /root/reference/vins_estimator/src.
#include "replay.h"
#include <chrono>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <numeric>

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
      imu.push_back({d[0], {d[1], d[2], d[3]}, {d[4], d[5], d[6]}});
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
    } else if (head[0] == 3 && (head[1] == (11 * 21 + 3 + 13) * sizeof(double) || head[1] == (11 * 21 + 3 + 14) * sizeof(double))) {
      const int W = FRAMES;
      const double *Ps = d, *Rs = Ps + 3 * W, *Vs = Rs + 9 * W, *Bas = Vs + 3 * W, *Bgs = Bas + 3 * W, *gg = Bgs + 3 * W, *t = gg + 3, *r = t + 3;
      TraceBoot tb;
      for (int i = 0; i < W; i++) {
        Keyframe &k = tb.state.kf[i];
        k.P = Vector3d(Ps[3 * i], Ps[3 * i + 1], Ps[3 * i + 2]), k.V = Vector3d(Vs[3 * i], Vs[3 * i + 1], Vs[3 * i + 2]);
        k.Ba = Vector3d(Bas[3 * i], Bas[3 * i + 1], Bas[3 * i + 2]), k.Bg = Vector3d(Bgs[3 * i], Bgs[3 * i + 1], Bgs[3 * i + 2]);
        for (int a = 0; a < 3; a++)
          for (int b = 0; b < 3; b++) k.R(a, b) = Rs[9 * i + 3 * a + b];
      }
      tb.state.g = Vector3d(gg[0], gg[1], gg[2]);
      tb.state.valid = true;
      std::memcpy(tb.tic, t, sizeof tb.tic), std::memcpy(tb.ric, r, sizeof tb.ric);
      tb.td = r[9];
      tb.stamp = head[1] == (11 * 21 + 3 + 14) * sizeof(double) ? r[10] : std::nan("");
      tb.at_image = images.size();
      if (!has_bootstrap) {
        bootstrap = tb.state;
        std::memcpy(tic, tb.tic, sizeof tic), std::memcpy(ric, tb.ric, sizeof ric);
        td = tb.td;
        has_bootstrap = true;
      }
      boots.push_back(tb);
    } else if (head[0] == 5) {
      restarts.push_back(images.size());
    }
  }
  std::fclose(f);
  return true;
}

void decodeFeatures(const TraceImage &msg, DecodedImage *out) {  // estimator_node.cpp:292-312
  const size_t n = msg.size();
  std::vector<int> key(n), idx(n);
  for (size_t i = 0; i < n; i++) key[i] = (int)(msg.v[9 * i + 3] + 0.5);  // id * NUM_OF_CAM + cam, one camera
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return key[a] < key[b]; });
  out->ids.clear(), out->pts.clear();
  for (size_t k = 0; k < n; k++) {
    const int i = idx[k];
    if (k > 0 && key[idx[k - 1]] == key[i]) continue;
    const float *p = &msg.v[9 * (size_t)i];
    out->ids.push_back(key[i]);
    const double v[8] = {p[0], p[1], p[2], p[4], p[5], p[6], p[7], p[8]};  // x y z | u v | velocity
    out->pts.insert(out->pts.end(), v, v + 8);
  }
}

int replay(WindowEstimator &est, const Trace &trace, const char *traj_path, int max_images, ReplayStats *stats, std::vector<double> *image_ms) {
  ReplayStats st;
  std::memset(&st, 0, sizeof st);
  FILE *traj = nullptr;
  if (traj_path && traj_path[0]) {
    traj = std::fopen(traj_path, "w");
    if (!traj) return -1;
  }
  auto finish = [&](int rc) {
    if (traj) std::fclose(traj);
    if (stats) *stats = st;
    return rc;
  };
  if (trace.has_bootstrap) {
    // the recording's extrinsic is the CONFIGURED one: a reset after a divergence restores it (setParameter(), estimator.cpp:10-21)
    std::memcpy(config().tic, trace.tic, sizeof trace.tic), std::memcpy(config().ric, trace.ric, sizeof trace.ric);
    config().td = trace.td;
    for (int k = 0; k < 3; k++) est.tic(k) = trace.tic[k];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) est.ric(i, j) = trace.ric[3 * i + j];
    est.td = trace.td;
  }
  size_t next_boot = 0, next_restart = 0, image_index = 0;
  const std::vector<TraceImu> &imu = trace.imu;
  size_t front = 0;         // head of the IMU queue
  double clock = -1;        // time of the last sample handed over
  double a[3] = {0, 0, 0}, w[3] = {0, 0, 0};
  DecodedImage img;
  for (const TraceImage &msg : trace.images) {
    if (max_images > 0 && st.images >= max_images) break;
    // restart messages that arrived before this image: restart_callback (estimator_node.cpp:187-204) empties the queues
    // and reboots the estimator; integration starts again with the next IMU message
    for (; next_restart < trace.restarts.size() && trace.restarts[next_restart] <= image_index; next_restart++) {
      est.reset();
      clock = -1;
      st.restarts++;
    }
    image_index++;
    // an estimator that is (again) initializing takes the next bootstrap record of the file: records that describe a time
    // already passed are skipped, one with a stamp waits for its image (until then the full window slides, as after an
    // initialStructure() that failed)
    if (est.phase == WindowEstimator::INITIAL && !est.bootstrap.valid) {
      while (next_boot < trace.boots.size() && trace.boots[next_boot].stamp < msg.t - 1e-6) next_boot++;
      if (next_boot < trace.boots.size()) {
        const TraceBoot &tb = trace.boots[next_boot];
        if (!(tb.stamp > msg.t + 1e-6)) {  // (NaN: no stamp, take it now)
          est.bootstrap = tb.state;
          next_boot++;
          st.bootstraps++;
        }
      }
    }
    // getMeasurements(), estimator_node.cpp:96-134, against the CURRENT time-offset estimate
    const double img_t = msg.t + est.td;
    if (front >= imu.size() || !(imu.back().t > img_t)) break;  // "wait for imu": nothing more arrives in a recording
    if (!(imu[front].t < img_t)) {                              // "throw img, only should happen at the beginning"
      st.thrown++;
      continue;
    }
    const auto t_image = std::chrono::steady_clock::now();
    const size_t first = front;
    while (imu[front].t < img_t) front++;
    // samples [first, front]: the first one after the image is used (interpolated) and stays queued for the next image (:127)
    for (size_t k = first; k <= front; k++) {  // process(), :218-262
      const TraceImu &m = imu[k];
      if (m.t <= img_t) {
        if (clock < 0) clock = m.t;
        const double dt = m.t - clock;
        clock = m.t;
        std::memcpy(a, m.acc, sizeof a), std::memcpy(w, m.gyr, sizeof w);
        est.pushImu(dt, a, w);
      } else {  // linear interpolation of the IMU at image time
        const double dt_1 = img_t - clock, dt_2 = m.t - img_t;
        clock = img_t;
        const double w1 = dt_2 / (dt_1 + dt_2), w2 = dt_1 / (dt_1 + dt_2);
        for (int d = 0; d < 3; d++) a[d] = w1 * a[d] + w2 * m.acc[d], w[d] = w1 * w[d] + w2 * m.gyr[d];
        est.pushImu(dt_1, a, w);
      }
    }
    decodeFeatures(msg, &img);
    const bool was_running = est.phase == WindowEstimator::NON_LINEAR;
    est.status = LFVIO_OK;
    est.pushImage(msg.t, (int)img.ids.size(), img.ids.data(), img.pts.data());
    if (image_ms) image_ms->push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t_image).count() * 1e3);
    st.images++;
    if (est.status != LFVIO_OK) {
      st.last_status = est.status;
      return finish(-2);  // a device call failed: no silent continuation
    }
    if (was_running && est.phase == WindowEstimator::INITIAL) st.failures++;  // diverged -> reset
    if (est.phase != WindowEstimator::NON_LINEAR) continue;
    (est.marg_flag == LFVIO_MARGIN_OLD ? st.keyframes : st.non_keyframes)++;
    st.iterations += est.summary.num_iterations;
    // pubOdometry(), utility/visualization.cpp:114-179: fixed, precision 12, "stamp x y z qx qy qz qw"
    const Keyframe &f = est.kf(WINDOW_SIZE);
    const Quaterniond q(f.R);
    if (traj) std::fprintf(traj, "%.12f %.12f %.12f %.12f %.12f %.12f %.12f %.12f\n", msg.t, f.P.x(), f.P.y(), f.P.z(), q.x(), q.y(), q.z(), q.w());
    st.poses++;
  }
  return finish(0);
}

}  // namespace lfvio
