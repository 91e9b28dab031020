// estimator.h — the part of `class Estimator` (vins_estimator/src/estimator.h:26-139) that Estimator::optimization()
// reads and writes, with the same member names, plus the drop-in optimization() that packs them for the C-ABI
// (include/vinsgpu.h: vg_ba_optimize) instead of building a ceres::Problem.  The rest of the reference class
// (processIMU, processImage, initialisation, slideWindow ...) is host bookkeeping outside the hot path and compiles
// unchanged around this method in a real catkin build.
#pragma once
#include <list>
#include <vector>
#include "compat/eigen_compat.h"
#include "../../include/vinsgpu.h"

using namespace Eigen;
using namespace std;

const int WINDOW_SIZE = 10;                  // vins_estimator/src/parameters.h:12
const int NUM_OF_CAM = 1;
const int NUM_OF_F = 1000;
extern int ESTIMATE_EXTRINSIC, ESTIMATE_TD, NUM_ITERATIONS;
extern double TD, TR, ROW_D, FOCAL_LENGTH_D, G_NORM;

struct FeaturePerFrame { Vector3d point; Vector2d uv; Vector2d velocity; double cur_td = 0; };   // feature_manager.h:19-42
struct FeaturePerId {                                                                             // feature_manager.h:44-66
    int feature_id = 0, start_frame = 0;
    vector<FeaturePerFrame> feature_per_frame;
    int used_num = 0;
    double estimated_depth = -1;
    int solve_flag = 0;            // 0 haven't solved yet; 1 solve succ; 2 solve fail
};
struct FeatureManager { list<FeaturePerId> feature; int getFeatureCount(); };

struct IntegrationBase {           // the members IMUFactor reads (factor/integration_base.h:188-208)
    double sum_dt = 0;
    Vector3d delta_p, delta_v, linearized_ba, linearized_bg;
    Quaterniond delta_q;
    double jacobian[225], covariance[225];      // row-major 15x15
};

struct MarginalizationInfo {       // what getParameterBlocks() leaves behind (marginalization_factor.h:44-72)
    int n = 0, m = 0;
    vector<int> keep_block_kind, keep_block_index;      // re-labelled for the slid window
    vector<double> keep_block_data;                     // concatenated, global sizes
    vector<double> linearized_jacobians;                // n x n row-major
    vector<double> linearized_residuals;
};

class Estimator {
  public:
    enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };
    Estimator();
    ~Estimator();
    void optimization();           // estimator.h:47
    void vector2double();
    void double2vector();

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
    vg_ba_summary last_summary;    // trace of the last solve (the reference only logs Summary::BriefReport)

  private:
    vg_handle* vg_ = nullptr;
};
