// estimator.h — host-side mirror of the reference surface around the hot path:
//   parameters  (vins_estimator/src/parameters.h:11-65)
//   Utility     (utility/utility.h:15-140)
//   IntegrationBase (factor/integration_base.h:9-208)
//   FeaturePerFrame / FeaturePerId / FeatureManager (feature_manager.h:18-106; depth vector get/set)
//   Estimator::{vector2double, double2vector, optimization} and the public state arrays
//     (estimator.h:30-136, estimator.cpp:488-626, 676-1009)
// Same names, members and argument meaning as the reference, so that the re-implemented
// optimization() drops into the ROS node; the arithmetic of the hot path itself runs on the GPU
// behind include/lfvio.h.  No Eigen/Ceres/ROS here: see small_eigen.h.
#pragma once
#include <list>
#include <map>
#include <utility>
#include <vector>

#include "../../include/lfvio.h"
#include "small_eigen.h"

namespace lfvio {

// ---- parameters.h
const double FOCAL_LENGTH = 160.0;
const int WINDOW_SIZE = 10;
const int NUM_OF_CAM = 1;
const int NUM_OF_F = 1000;  // the reference's para_Feature capacity (unchecked there); not a limit here
extern double ACC_N, ACC_W, GYR_N, GYR_W;
extern Vector3d G;
extern double SOLVER_TIME;
extern double INIT_DEPTH;
extern int NUM_ITERATIONS;
extern int ESTIMATE_EXTRINSIC, ESTIMATE_TD;
extern double TD, TR, ROW, COL;
extern double MIN_PARALLAX;  // keyframe_parallax / FOCAL_LENGTH (parameters.cpp:56-57)
enum SIZE_PARAMETERIZATION { SIZE_POSE = 7, SIZE_SPEEDBIAS = 9, SIZE_FEATURE = 1 };
enum StateOrder { O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12 };

// ---- utility.h
struct Utility {
  static Quaterniond deltaQ(const Vector3d &theta) { return Quaterniond(1.0, theta.x() / 2.0, theta.y() / 2.0, theta.z() / 2.0); }
  static Matrix3d skewSymmetric(const Vector3d &q) {
    Matrix3d a;
    a(0, 1) = -q(2), a(0, 2) = q(1), a(1, 0) = q(2), a(1, 2) = -q(0), a(2, 0) = -q(1), a(2, 1) = q(0);
    return a;
  }
  static Vector3d R2ypr(const Matrix3d &R);          // degrees
  static Matrix3d ypr2R(const Vector3d &ypr);        // degrees
};

// ---- integration_base.h
class IntegrationBase {
 public:
  IntegrationBase() = delete;
  IntegrationBase(const Vector3d &_acc_0, const Vector3d &_gyr_0, const Vector3d &_linearized_ba, const Vector3d &_linearized_bg);
  void push_back(double dt, const Vector3d &acc, const Vector3d &gyr);
  void repropagate(const Vector3d &_linearized_ba, const Vector3d &_linearized_bg);
  void midPointIntegration(double _dt, const Vector3d &_acc_0, const Vector3d &_gyr_0, const Vector3d &_acc_1, const Vector3d &_gyr_1,
                           const Vector3d &delta_p, const Quaterniond &delta_q, const Vector3d &delta_v, const Vector3d &linearized_ba,
                           const Vector3d &linearized_bg, Vector3d &result_delta_p, Quaterniond &result_delta_q, Vector3d &result_delta_v,
                           Vector3d &result_linearized_ba, Vector3d &result_linearized_bg, bool update_jacobian);
  void propagate(double _dt, const Vector3d &_acc_1, const Vector3d &_gyr_1);
  // the 15-residual evaluate() of the reference runs on the device (dev_factors.h: imu_raw_residual)

  double dt = 0;
  Vector3d acc_0, gyr_0, acc_1, gyr_1;
  const Vector3d linearized_acc, linearized_gyr;
  Vector3d linearized_ba, linearized_bg;
  Mat<15, 15> jacobian, covariance;
  Mat<18, 18> noise;
  double sum_dt = 0;
  Vector3d delta_p;
  Quaterniond delta_q;
  Vector3d delta_v;
  std::vector<double> dt_buf;
  std::vector<Vector3d> acc_buf, gyr_buf;
};

// ---- feature_manager.h
class FeaturePerFrame {
 public:
  FeaturePerFrame(const double _point[8], double td) {
    point = Vector3d(_point[0], _point[1], _point[2]);
    uv.x() = _point[3], uv.y() = _point[4];
    velocity = Vector3d(_point[5], _point[6], _point[7]);
    cur_td = td;
  }
  double cur_td;
  Vector3d point;
  Vector2d uv;
  Vector3d velocity;
};

class FeaturePerId {
 public:
  const int feature_id;
  int start_frame;
  std::vector<FeaturePerFrame> feature_per_frame;
  int used_num;
  double estimated_depth;
  int solve_flag;  // 0 haven't solve yet; 1 solve succ; 2 solve fail
  FeaturePerId(int _feature_id, int _start_frame)
      : feature_id(_feature_id), start_frame(_start_frame), used_num(0), estimated_depth(-1.0), solve_flag(0) {}
  int endFrame() { return start_frame + (int)feature_per_frame.size() - 1; }
};

// the decoded feature message, estimator_node.cpp:292-312: feature_id -> [(camera_id, x y z u v vx vy vz)]
typedef Mat<8, 1> Vector8d;
typedef std::map<int, std::vector<std::pair<int, Vector8d>>> ImageMap;

class FeatureManager {
 public:
  void clearState() { feature.clear(); }
  int getFeatureCount();
  void setDepth(const VectorXd &x);
  void clearDepth(const VectorXd &x);
  void removeFailures();
  VectorXd getDepthVector();
  // SURVEY §8f rank 2 (feature_manager.cpp:199-253, 271-351): the landmark-parallel arithmetic runs behind
  // lfvio_triangulate / lfvio_shift_depth, the std::list surgery stays here
  void triangulate(Vector3d Ps[], Vector3d tic[], Matrix3d ric[]);
  void removeBackShiftDepth(Matrix3d marg_R, Vector3d marg_P, Matrix3d new_R, Vector3d new_P);
  void removeBack();
  void removeFront(int frame_count);
  // SURVEY §8f rank 4 (feature_manager.cpp:45-95, 353-369): the keyframe policy
  bool addFeatureCheckParallax(int frame_count, const ImageMap &image, double td);
  double compensatedParallax2(const FeaturePerId &it_per_id, int frame_count);
  int last_track_num = 0;
  FeaturePerId &addFeature(int feature_id, int start_frame);  // test / packing helper, not in the reference
  std::list<FeaturePerId> feature;
  const Matrix3d *Rs = nullptr;  // the estimator's Rs[] (FeatureManager(Matrix3d _Rs[]), feature_manager.cpp:13)
  lfvio_ctx **gpu = nullptr;     // the estimator's device context (created on first use)
  int last_status = 0;
};

// ---- estimator.h
class Estimator {
 public:
  Estimator();
  ~Estimator();
  void setParameter();
  void clearState();
  void optimization();
  // the batched repropagate of visualInitialAlign (estimator.cpp:403-406): every pre_integrations[i] redone from its
  // buffers with new linearization biases, all intervals in ONE lfvio_preintegrate call; status in last_status
  void repropagateWindow(const Vector3d ba[(WINDOW_SIZE + 1)], const Vector3d bg[(WINDOW_SIZE + 1)]);
  void vector2double();
  void double2vector();
  // the per-measurement control flow around optimization() (estimator.cpp:86-220, 475-486,
  // 628-674, 1011-1131).  initialStructure() (SfM + visual-inertial alignment, out of scope) is replaced by a state record
  // the caller supplies (`bootstrap`); everything after it is the reference's sequence.
  void processIMU(double dt, const Vector3d &linear_acceleration, const Vector3d &angular_velocity);
  void processImage(const ImageMap &image, double header_stamp);
  bool initialStructure();
  void solveOdometry();
  void slideWindow();
  void slideWindowNew();
  void slideWindowOld();
  bool failureDetection();

  enum SolverFlag { INITIAL, NON_LINEAR };
  SolverFlag solver_flag = INITIAL;
  int frame_count = 0;
  double Headers[(WINDOW_SIZE + 1)];  // header.stamp.toSec()
  bool first_imu = false;
  Vector3d acc_0, gyr_0, g;
  std::vector<double> dt_buf[(WINDOW_SIZE + 1)];
  std::vector<Vector3d> linear_acceleration_buf[(WINDOW_SIZE + 1)], angular_velocity_buf[(WINDOW_SIZE + 1)];
  Matrix3d back_R0, last_R;
  Vector3d back_P0, last_P;
  std::vector<Vector3d> key_poses;
  int sum_of_back = 0, sum_of_front = 0;
  double initial_timestamp = 0;
  struct Bootstrap {  // what initialStructure() + visualInitialAlign() leave behind (estimator.cpp:222-473), from outside
    bool valid = false;
    Vector3d Ps[(WINDOW_SIZE + 1)], Vs[(WINDOW_SIZE + 1)], Bas[(WINDOW_SIZE + 1)], Bgs[(WINDOW_SIZE + 1)], g;
    Matrix3d Rs[(WINDOW_SIZE + 1)];
  } bootstrap;

  enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };
  MarginalizationFlag marginalization_flag = MARGIN_OLD;

  Matrix3d ric[NUM_OF_CAM];
  Vector3d tic[NUM_OF_CAM];
  Vector3d Ps[(WINDOW_SIZE + 1)];
  Vector3d Vs[(WINDOW_SIZE + 1)];
  Matrix3d Rs[(WINDOW_SIZE + 1)];
  Vector3d Bas[(WINDOW_SIZE + 1)];
  Vector3d Bgs[(WINDOW_SIZE + 1)];
  double td = 0;
  Matrix3d last_R0;
  Vector3d last_P0;
  IntegrationBase *pre_integrations[(WINDOW_SIZE + 1)];
  FeatureManager f_manager;
  bool failure_occur = false;

  double para_Pose[WINDOW_SIZE + 1][SIZE_POSE];
  double para_SpeedBias[WINDOW_SIZE + 1][SIZE_SPEEDBIAS];
  std::vector<double> para_Feature;  // [feature][SIZE_FEATURE]; the reference's fixed NUM_OF_F array has no bounds check
  double para_Ex_Pose[NUM_OF_CAM][SIZE_POSE];
  double para_Td[1][1];

  // MarginalizationInfo *last_marginalization_info + last_marginalization_parameter_blocks of the reference,
  // carried in the ABI's (kind, frame)-tagged form
  LfvioPrior *last_marginalization_info = nullptr;

  // bookkeeping of the last call (the reference only logs these through ROS_DEBUG)
  LfvioSolution last_summary;
  int last_status = 0;
  lfvio_ctx *gpu = nullptr;
  // true: one upload, solve -> gauge fix -> marginalization on the device (lfvio_batch_* on one slot);
  // false: the literal flow, lfvio_solve / host double2vector() / lfvio_marginalize (two uploads)
  bool fused = true;

  // pack para_* + features + pre-integrations + prior into the ABI POD (buffers live in `scratch_`)
  void packWindow(LfvioWindow *w);

 private:
  struct Scratch {
    std::vector<int> start_frame, obs_offset;
    std::vector<double> inv_depth, point, velocity, cur_td, uv_y;
  } scratch_;
};

}  // namespace lfvio
