// estimator.h — the part of `class Estimator` (vins_estimator/src/estimator.h:26-139) that Estimator::optimization()
// reads and writes, with the same member names, plus the drop-in optimization() that packs them for the C-ABI
// (include/vinsgpu.h: vg_ba_optimize) instead of building a ceres::Problem.  The rest of the reference class
// (processIMU, processImage, initialisation, slideWindow ...) is host bookkeeping outside the hot path and compiles
// unchanged around this method in a real catkin build.
#pragma once
#include <list>
#include <string>
#include <vector>
#include "compat/eigen_compat.h"
#include "../../include/vinsgpu.h"

using namespace Eigen;
using namespace std;

const int WINDOW_SIZE = 10;                  // vins_estimator/src/parameters.h:12
const int NUM_OF_CAM = 1;
const int NUM_OF_F = 1000;
extern int ESTIMATE_EXTRINSIC, ESTIMATE_TD, NUM_ITERATIONS;
extern double TD, TR, ROW_D, FOCAL_LENGTH_D, G_NORM, INIT_DEPTH, SOLVER_TIME;
extern double ACC_N, ACC_W, GYR_N, GYR_W;          // vins_estimator/src/parameters.cpp:5-6 (the device pre-integration needs them)
extern double MIN_PARALLAX, BIAS_ACC_THRESHOLD, BIAS_GYR_THRESHOLD, COL_D;
extern int ROLLING_SHUTTER;
extern std::string IMU_TOPIC, VINS_RESULT_PATH, EX_CALIB_RESULT_PATH;
extern std::vector<Matrix3d> RIC;
extern std::vector<Vector3d> TIC;
// vins_estimator/src/parameters.cpp:42-137 without the ROS node handle and without touching the file system (the reference
// creates OUTPUT_PATH and truncates the result file there; here only the names are formed)
void readEstimatorParameters(const std::string& config_file);

struct FeaturePerFrame { Vector3d point; Vector2d uv; Vector2d velocity; double cur_td = 0; };   // feature_manager.h:19-42
struct FeaturePerId {                                                                             // feature_manager.h:44-66
    int feature_id = 0, start_frame = 0;
    vector<FeaturePerFrame> feature_per_frame;
    int used_num = 0;
    double estimated_depth = -1;
    int solve_flag = 0;            // 0 haven't solved yet; 1 solve succ; 2 solve fail
};
struct FeatureManager {
    list<FeaturePerId> feature;
    int getFeatureCount();
    // window-shift bookkeeping (feature_manager.cpp:275-341); host code, run between two optimization() calls
    void removeBackShiftDepth(Matrix3d marg_R, Vector3d marg_P, Matrix3d new_R, Vector3d new_P);
    void removeBack();
    void removeFront(int frame_count);
};

struct IntegrationBase {           // the members IMUFactor reads (factor/integration_base.h:188-208), same types
    double sum_dt = 0;
    Vector3d delta_p, delta_v, linearized_ba, linearized_bg;
    Quaterniond delta_q;
    Eigen::Matrix<double, 15, 15> jacobian, covariance;      // column-major (Eigen default), integration_base.h:195
    // the raw samples of the interval (integration_base.h:192, :205-208): slideWindow() folds the samples of a dropped
    // non-keyframe into the previous interval (estimator.cpp:1073-1085) and re-integrates it on the device
    Vector3d linearized_acc, linearized_gyr;                 // first measurement of the interval
    std::vector<double> dt_buf;
    std::vector<Vector3d> acc_buf, gyr_buf;
};

struct MarginalizationInfo {       // the members MarginalizationFactor reads (marginalization_factor.h:44-72), same names / types
    ~MarginalizationInfo() { for (double* p : keep_block_data) delete[] p; }
    int m = 0, n = 0;
    std::vector<int> keep_block_size;       // global size of every kept block (7 / 9 / 7 / 1)
    std::vector<int> keep_block_idx;        // its first column in the [m dropped | n kept] ordering (local, so >= m)
    std::vector<double*> keep_block_data;   // owned copy of its linearisation point
    Eigen::MatrixXd linearized_jacobians;   // n x n
    Eigen::VectorXd linearized_residuals;   // n
};

class Estimator {
  public:
    enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };
    Estimator();
    ~Estimator();
    void optimization();           // estimator.h:47
    void collectPrior();           // picks up the marginalization result of the last optimization() (it runs behind the state download)
    void vector2double();
    void double2vector();
    void slideWindow();            // estimator.cpp:1005-1126 (state / pre-integration shift + slideWindowOld / slideWindowNew)
    // IntegrationBase::repropagate (integration_base.h:38-52) on the device: integrates p's raw samples with its linearized biases
    void repropagate(IntegrationBase* p);

    MarginalizationFlag marginalization_flag = MARGIN_OLD;
    Vector3d Ps[(WINDOW_SIZE + 1)], Vs[(WINDOW_SIZE + 1)], Bas[(WINDOW_SIZE + 1)], Bgs[(WINDOW_SIZE + 1)];
    Matrix3d Rs[(WINDOW_SIZE + 1)];
    Matrix3d ric[NUM_OF_CAM];
    Vector3d tic[NUM_OF_CAM];
    double td = 0;
    IntegrationBase* pre_integrations[(WINDOW_SIZE + 1)];
    FeatureManager f_manager;
    double para_Pose[WINDOW_SIZE + 1][7], para_SpeedBias[WINDOW_SIZE + 1][9], para_Feature[NUM_OF_F][1], para_Ex_Pose[NUM_OF_CAM][7], para_Td[1][1];
    MarginalizationInfo* last_marginalization_info = nullptr;
    vector<double*> last_marginalization_parameter_blocks;      // para_* address of every kept block, already shifted (estimator.h:117)
    // relocalisation (estimator.h:125-138): setReloFrame() of the surrounding reference code fills the inputs, optimization()
    // adds the matched landmarks' factors against relo_Pose (:769-801), double2vector() leaves the by-products (:596-616)
    bool relocalization_info = false;
    int relo_frame_local_index = 0;
    vector<Vector3d> match_points;             // (x, y, feature_id), ascending feature_id
    double relo_Pose[7];
    Matrix3d drift_correct_r, prev_relo_r, relo_r_fixed;
    Vector3d drift_correct_t, prev_relo_t, relo_relative_t, relo_t_fixed;
    Quaterniond relo_relative_q;
    double relo_relative_yaw = 0;
    bool relo_in_problem = false;              // the last optimization() carried relocalisation factors
    Matrix3d back_R0;
    Vector3d back_P0;
    bool prior_pending = false;    // optimization() has returned, its marginalization result is still on the device
    bool solver_failed = false;    // the device reported a non-finite solve: the prior was dropped (see optimization())
    vg_ba_summary last_summary;    // trace of the last solve (the reference only logs Summary::BriefReport)

    // (kind, frame index) of a parameter-block address inside para_Pose / para_SpeedBias / para_Ex_Pose / para_Td
    bool block_of(const double* addr, int& kind, int& index) const;

  private:
    vg_handle* vg_ = nullptr;
};
