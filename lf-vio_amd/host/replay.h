// replay.h — ROS-free recording of what estimator_node.cpp consumes, and the loop that feeds it to the WindowEstimator
// (SURVEY §8f rank 1).  A trace is the two topics of the node in arrival order plus what stands in for initialStructure():
//
//   file   = "LFVT" u32 version(1)  { u32 type, u32 bytes, payload[bytes] }*          (little-endian, packed)
//   type 1 = sensor_msgs/Imu          f64 stamp, f64 linear_acceleration[3], f64 angular_velocity[3]   (imu_callback, estimator_node.cpp:136-161)
//   type 2 = sensor_msgs/PointCloud   f64 stamp, u32 n, n x f32[9]:
//                                     points[i].{x,y,z} (geometry_msgs/Point32: float32 — the bearing as the tracker
//                                     published it, feature_tracker_node.cpp:146-151), then channels[0..5].values[i]:
//                                     id * NUM_OF_CAM + cam, u, v, velocity x, y, z (:152-163; decoded at estimator_node.cpp:292-312)
//   type 3 = bootstrap                f64 Ps[11][3], Rs[11][9] (row-major), Vs[11][3], Bas[11][3], Bgs[11][3], g[3], tic[3], ric[9], td
//                                     [, f64 stamp] — the window state initialStructure() + visualInitialAlign() produced
//                                     (estimator.cpp:160-173), dumped where initialStructure() returns true; the optional
//                                     stamp is Headers[WINDOW_SIZE] of that moment: the record is then used for the image
//                                     with that stamp (earlier images of a filling window slide, as after a failed
//                                     initialStructure(), :177-178), without it at the first full window.  A recording may
//                                     hold several: after failureDetection() rebooted the estimator (:196-204) or a restart
//                                     message cleared it, the next one in the file is taken.
//   type 4 = ground truth (optional)  f64 stamp, p[3], q[4] (x y z w); skipped by the replay, read by tools/ate.py
//   type 5 = restart                  [f64 stamp] — std_msgs/Bool(true) on the tracker's restart topic: restart_callback
//                                     (estimator_node.cpp:187-204) clears the buffers, clearState(), setParameter()
//
// Unknown record types are skipped, so a recorder can add its own.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "window_estimator.h"

namespace lfvio {

struct TraceImu {
  double t;
  double acc[3], gyr[3];
};
struct TraceImage {
  double t;
  std::vector<float> v;  // n x 9
  size_t size() const { return v.size() / 9; }
};
struct TraceBoot {
  WindowEstimator::Bootstrap state;
  double tic[3], ric[9], td;
  double stamp;      // NaN: take it at the first full window
  size_t at_image;   // images recorded before it
};
struct Trace {
  std::vector<TraceImu> imu;
  std::vector<TraceImage> images;
  std::vector<TraceBoot> boots;       // in file order
  std::vector<size_t> restarts;       // restart messages: number of images recorded before each
  bool has_bootstrap = false;         // the first bootstrap record (what a recording without reboots has)
  WindowEstimator::Bootstrap bootstrap;
  double tic[3] = {0, 0, 0}, ric[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  double td = 0;
  std::string error;
  bool load(const char *path);
};

struct ReplayStats {
  int images, thrown, keyframes, non_keyframes, poses, failures, last_status, iterations, restarts, bootstraps;
};

// feature message -> what WindowEstimator::pushImage() takes (estimator_node.cpp:292-312): feature ids ascending (the
// reference keys a std::map by id), one 8-vector x y z u v vx vy vz per id (camera 0's, the first entry of an id)
struct DecodedImage {
  std::vector<int> ids;
  std::vector<double> pts;  // n x 8
};
void decodeFeatures(const TraceImage &msg, DecodedImage *out);

// getMeasurements() + process() of estimator_node.cpp (:96-134, :206-342) over a loaded trace, single-threaded;
// after every image in NON_LINEAR state one line of the trajectory file as pubOdometry() writes it
// (utility/visualization.cpp:173-179).  Returns 0 or a negative error; `stats` may be null.
// image_ms (optional): wall-clock milliseconds of every image the loop handed over — its IMU samples and processImage(), i.e. what
// process() spends per measurement (estimator_node.cpp:206-342) — in arrival order
int replay(WindowEstimator &estimator, const Trace &trace, const char *traj_path, int max_images, ReplayStats *stats, std::vector<double> *image_ms = nullptr);

}  // namespace lfvio
