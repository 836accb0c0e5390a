// window_estimator.h — host side of the drop-in, shaped for the device it feeds.
//
// The reference keeps the sliding window in `Estimator` as eleven parallel arrays that slideWindow() shuffles element by
// element (estimator.cpp:1011-1105), the landmarks as a std::list<FeaturePerId> of std::vector<FeaturePerFrame>
// (feature_manager.h:18-71) that every call walks and re-packs, and integrates every IMU sample on the host as it
// arrives (integration_base.h:29-158).  Here:
//   * FrameRing        — the eleven keyframes live in a ring: a MARGIN_OLD slide is one index increment, nothing moves;
//   * TrackTable       — landmarks are rows of ONE flat table, nine doubles per observation, eleven observation rows per
//                        track addressed through a per-track ring offset: dropping the oldest observation of every track
//                        (removeBack) moves no data either, lookups by feature id are a hash probe instead of the
//                        reference's linear std::find_if per feature, and the CSR arrays of LfvioWindow are produced by
//                        one linear pass of memcpy-sized copies into a staging block that is reused from call to call;
//   * ImuSpan          — raw samples are only buffered; the pre-integration of the spans that changed (normally the
//                        newest one) happens on the device in ONE lfvio_preintegrate call when optimization() needs it,
//                        so no 15x15 products run on the host at all.
// The functions the hot path's contract names stay what they are: vector2double() / double2vector() (a2, bit-exact on
// the host) and optimization() (a1) keep the reference's names and semantics; everything else is named for what it does,
// and INTEGRATION.md maps the reference's members onto it.
#pragma once
#include <chrono>
#include <unordered_map>
#include <vector>

#include "../../include/lfvio.h"
#include "small_eigen.h"

namespace lfvio {

constexpr int WINDOW_SIZE = LFVIO_WINDOW_SIZE;
constexpr int FRAMES = LFVIO_NUM_FRAMES;
constexpr double FOCAL_LENGTH = 160.0;  // parameters.h:11

// parameters.cpp / the YAML file: one process-wide record, like the reference's globals
struct Config {
  double acc_n = 0.02, gyr_n = 0.01, acc_w = 0.04, gyr_w = 0.001;  // mindvision.yaml:138-141
  double gravity[3] = {0.0, 0.0, 9.81007};                         // :142
  double solver_time = 0.04, init_depth = 5.0;                     // parameters.cpp:133, 116
  int num_iterations = 8, estimate_extrinsic = 1, estimate_td = 1;  // :134, :83, :151
  double td = -0.008, tr = 0.0, row = 960.0;
  double min_parallax = 10.0 / FOCAL_LENGTH;                       // keyframe_parallax / FOCAL_LENGTH, parameters.cpp:56-57
  // configured extrinsic (TIC / RIC, parameters.cpp:88-112): what setParameter() restores after every reset
  double tic[3] = {0, 0, 0};
  double ric[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  // HIP devices the estimator may use: bit d = device d.  One bit (default: device 0): a single context.  Several bits: an
  // lfvio_group over them — optimization() becomes ONE lfvio_group_solve(), the window's landmarks sharded over the
  // devices with RCCL all-reduces inside the library (include/lfvio.h; SURVEY §8b lfvio_create(device_mask), §8e).
  unsigned device_mask = 1u;
  // > 1: the group is `local_shards` ranks on the FIRST device of the mask (lfvio_group_create_local: the collective a
  // device-side sum) — how a one-GPU box runs the multi-rank path of the estimator end to end (tests)
  int local_shards = 0;
  // optimization() on one device returns with the state as soon as solve + gauge fix are out and lets the marginalization
  // finish in the background (lfvio_batch_optimize_begin / _finish): the pose is published ~0.2 ms earlier, the prior is
  // collected when the next window is packed.  false: the call returns when the prior is on the host as well.
  bool split_call = true;
  // ... and the next window takes that prior over ON THE DEVICE (lfvio_batch_upload_chained_device): the upload goes out behind the
  // marginalization without waiting for it, the prior never crosses PCIe.  The host's copy (`prior`) is then only refreshed when
  // something asks for it (collectPrior()).  false: the upload collects the prior and sends it back up (lfvio_batch_upload_chained).
  bool device_chain = true;
};
Config &config();

// yaw / pitch / roll in DEGREES (utility.h:66-113)
Vector3d yawPitchRollDeg(const Matrix3d &R);
Matrix3d fromYawPitchRollDeg(const Vector3d &ypr);

struct FrameRing {
  int head = 0;
  int phys(int logical) const { return (head + logical) % FRAMES; }
  void advance() { head = (head + 1) % FRAMES; }
};

struct Keyframe {
  Vector3d P, V, Ba, Bg;
  Matrix3d R;
  double stamp = 0;
};

// IMU samples between two keyframes + the IntegrationBase fields the device made of them
struct ImuSpan {
  bool present = false;  // the reference's pre_integrations[i] != nullptr
  bool dirty = false;    // `pre` is stale: samples or linearization biases changed since the last device call
  std::vector<double> dt, acc, gyr;  // n | 3n | 3n
  double acc0[3] = {0, 0, 0}, gyr0[3] = {0, 0, 0}, lin_ba[3] = {0, 0, 0}, lin_bg[3] = {0, 0, 0};
  double sum_dt = 0;
  LfvioPreintegration pre;
  void open(const Vector3d &a0, const Vector3d &g0, const Vector3d &ba, const Vector3d &bg);
  void push(double d, const double *a, const double *g);
  void close() {
    present = dirty = false;
    dt.clear(), acc.clear(), gyr.clear();
    sum_dt = 0;
  }
  int samples() const { return (int)dt.size(); }
};

// One row per observation: bearing xyz, pixel uv, bearing velocity xyz (the 8-vector of estimator_node.cpp:308), cur_td
constexpr int OBS_W = 9;

class TrackTable {
 public:
  void clear();
  int live() const { return (int)order_.size(); }
  // slots in insertion order (the reference's list order)
  const std::vector<int> &order() const { return order_; }
  int id(int s) const { return id_[s]; }
  int start(int s) const { return start_[s]; }
  int count(int s) const { return count_[s]; }
  double depth(int s) const { return depth_[s]; }
  void setDepth(int s, double d) { depth_[s] = d; }
  int solveFlag(int s) const { return flag_[s]; }
  void setSolveFlag(int s, int f) { flag_[s] = f; }
  const double *obs(int s, int k) const { return &rows_[((size_t)s * FRAMES + (first_[s] + k) % FRAMES) * OBS_W]; }
  bool solvable(int s) const { return count_[s] >= 2 && start_[s] < WINDOW_SIZE - 2; }  // feature_manager.cpp:36
  int solvableCount() const;

  int find(int feature_id) const;
  int create(int feature_id, int start_frame);
  void append(int s, const double *pt8, double cur_td);

  // feature_manager.cpp:45-95: the frame's observations (ids ascending, first occurrence of an id wins) are appended;
  // returns the number of continued tracks
  int appendFrame(int frame_count, int n, const int *ids, const double *pts8, double td);
  // feature_manager.cpp:353-369 summed over the tracks seen in both of the two frames before the newest one
  void parallax(int frame_count, double *sum, int *num) const;
  // feature_manager.cpp:312-330 / 271-310: every track loses its frame-0 observation and moves one frame down; with
  // `survivors` the tracks that started in frame 0 and keep >= 2 observations are reported (slot, erased bearing) for the
  // depth re-anchoring and shorter ones are dropped, without it a track is dropped when it becomes empty
  struct Shifted {
    int slot;
    double bearing[3];
  };
  void dropOldestFrame(std::vector<Shifted> *survivors);
  // feature_manager.cpp:332-351: the second newest frame is thrown away
  void dropSecondNewestFrame(int frame_count);
  void dropFailed();  // removeFailures(): solve_flag == 2

 private:
  void erase(int s);
  void compact();
  std::vector<int> id_, start_, count_, first_, flag_;
  std::vector<char> dead_;
  std::vector<double> depth_, rows_;
  std::vector<int> order_, free_;
  std::unordered_map<int, int> slot_of_;
  bool holes_ = false;
};

class WindowEstimator {
 public:
  WindowEstimator();
  ~WindowEstimator();
  WindowEstimator(const WindowEstimator &) = delete;

  // ---- the hot path's contract: reference names, reference semantics
  void vector2double();  // estimator.cpp:488-530
  void double2vector();  // estimator.cpp:532-600
  void optimization();   // estimator.cpp:676-1009 over include/lfvio.h
  bool collectPrior();   // waits for a marginalization still in flight (split_call) and adopts its prior; false + status on error

  // ---- the loop around it: processIMU / processImage / solveOdometry / slideWindow / failureDetection
  void reset();                                                     // clearState() + setParameter()
  void pushImu(double dt, const double acc[3], const double gyr[3]);
  void pushImage(double stamp, int n, const int *ids, const double *pts8);
  bool keyframeTest(int frame_count, int n, const int *ids, const double *pts8, double td);  // true: MARGIN_OLD
  void slide();
  bool diverged() const;
  void triangulate();
  void reanchorDepths(const Matrix3d &old_R, const Vector3d &old_P, const Matrix3d &new_R, const Vector3d &new_P,
                      const std::vector<TrackTable::Shifted> &tracks);
  bool refreshSpans(bool all, const Vector3d *ba = nullptr, const Vector3d *bg = nullptr);  // device pre-integration of dirty spans
  void pack(LfvioWindow *w);

  // keyframe i of the window (logical index through the ring)
  Keyframe &kf(int i) { return frames_[ring_.phys(i)]; }
  const Keyframe &kf(int i) const { return frames_[ring_.phys(i)]; }
  ImuSpan &span(int i) { return spans_[ring_.phys(i)]; }
  const ImuSpan &span(int i) const { return spans_[ring_.phys(i)]; }

  enum Phase { INITIAL = 0, NON_LINEAR = 1 };
  Phase phase = INITIAL;
  int frame_count = 0;
  int marg_flag = LFVIO_MARGIN_OLD;
  bool first_imu = false, failure_occur = false, fused = true;
  Vector3d acc_prev, gyr_prev, g;  // the previous IMU sample; gravity as aligned
  Matrix3d ric, last_R, last_R0;
  Vector3d tic, last_P, last_P0;
  double td = 0, initial_timestamp = 0;
  int slides_old = 0, slides_new = 0, tracked_last = 0;
  // wall clock spent in the device-backed steps since the last reset of the timers (seconds; tools/replay_stream.py):
  // [0] optimization() up to the state, [1] collectPrior(), [2] triangulate(), [3] reanchorDepths(), [4] refreshSpans(), [5] calls of [0]
  double timers[6] = {0, 0, 0, 0, 0, 0};
  TrackTable tracks;

  struct Bootstrap {  // what initialStructure() + visualInitialAlign() would leave (estimator.cpp:222-473), from outside
    bool valid = false;
    Keyframe kf[FRAMES];
    Vector3d g;
  } bootstrap;

  // the flat parameter arrays Ceres sees (estimator.h:107-113)
  double para_Pose[FRAMES][LFVIO_SIZE_POSE];
  double para_SpeedBias[FRAMES][LFVIO_SIZE_SPEEDBIAS];
  double para_Ex_Pose[1][LFVIO_SIZE_POSE];
  double para_Td[1][1];
  std::vector<double> para_Feature;

  bool has_prior = false;
  LfvioPrior prior;  // last_marginalization_info + its parameter blocks, (kind, frame)-tagged
  LfvioSolution summary;
  int status = LFVIO_OK;
  lfvio_ctx *gpu = nullptr;      // the context (of the first device of the mask): triangulation, depth shifts, preintegration
  lfvio_group *group = nullptr;  // more than one device in Config::device_mask: the sharded optimization()

 private:
  LfvioPrior next_;  // the prior being downloaded (240 KB: a member, not a stack object; only header + n x n + n are copied)
  bool chain_upload_ = false;   // pack() on behalf of an optimization() whose upload collects the pending prior itself
  bool prior_pending_ = false;  // the marginalization of the last optimization() has not been collected yet (collectPrior)
  bool prior_on_device_ = false;  // the resident window took its prior over on the device (lfvio_batch_upload_chained_device): `prior` is stale
  bool device();
  bool applyBootstrap();
  FrameRing ring_;
  Keyframe frames_[FRAMES];
  ImuSpan spans_[FRAMES];
  struct Staging {  // backing store of the pointers in LfvioWindow; grows, never shrinks
    std::vector<int> start_frame, obs_offset;
    std::vector<double> inv_depth, point, velocity, cur_td, uv_y, lam_out;
  } stage_;
};

}  // namespace lfvio
