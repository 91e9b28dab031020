// resident_estimator.h — N estimators in their NON_LINEAR phase whose sliding windows live on the device between frames
// (include/vinsgpu.h: vg_ba_seq_*; csrc/ba_seq.hip).  The two callbacks keep the reference's signatures
// (vins_estimator/src/estimator.h:32-33) with the estimator's index in front:
//     processIMU(i, dt, linear_acceleration, angular_velocity)        estimator.cpp:83-117
//     processImage(i, image)                                          estimator.cpp:120-215 (solver_flag == NON_LINEAR branch)
// processIMU does what the reference's does on the host: it keeps the raw samples of the running interval (dt_buf / acc_buf /
// gyr_buf of IntegrationBase) and propagates Ps / Rs / Vs of the newest frame.  processImage only stores the frame;
// solve() then runs ONE device step for all estimators (addFeatureCheckParallax, triangulate, optimization(), slideWindow(),
// removeFailures(): vg_ba_seq_step_async) and brings the states back.  f_manager, para_*, pre_integrations[] and
// last_marginalization_info have no host copy: that is the point.
#pragma once
#include <map>
#include <utility>
#include <vector>
#include "estimator.h"

class ResidentEstimators {
  public:
    typedef std::map<int, std::vector<std::pair<int, Eigen::Matrix<double, 7, 1>>>> Image;   // feature_id -> [(camera_id, x y z u v vx vy)]
    struct Interval {                         // what IntegrationBase keeps of one frame interval (integration_base.h:188-208)
        Vector3d linearized_acc, linearized_gyr, linearized_ba, linearized_bg;
        std::vector<double> samples;          // rows (dt, acc, gyr)
    };
    struct One {
        Vector3d Ps[WINDOW_SIZE + 1], Vs[WINDOW_SIZE + 1], Bas[WINDOW_SIZE + 1], Bgs[WINDOW_SIZE + 1];   // as the last solve + slide left them
        Matrix3d Rs[WINDOW_SIZE + 1];
        Matrix3d ric;
        Vector3d tic;
        double td = 0;
        Vector3d acc_0, gyr_0, g;
        bool first_imu = false;
        Estimator::MarginalizationFlag marginalization_flag = Estimator::MARGIN_OLD;   // of the last solve()
        vg_ba_summary last_summary;
        int n_features = 0, status = 0;       // tracks left after the slide; VG_OK or VG_ERR_UNSUPPORTED (a capacity was exceeded)
        int last_track_num = 0;               // f_manager.last_track_num of the last frame
        // Estimator::failureDetection() (estimator.cpp:621-667) on the window as solved, BEFORE its slide, against last_P / last_R
        // of the frame before (:205-208).  The reference reboots on failure (clearState + setParameter, :193-199).  Here the flag is
        // raised and the caller decides: the windows of a batch begin together (vg_ba_seq_begin), so re-initialising ONE estimator
        // means handing the batch over again (a per-window re-seed of a running sequence is not offered yet).
        bool failure_occur = false;
        Vector3d last_P, last_P0;
        Matrix3d last_R, last_R0;
        // ---- internal
        Interval cur, prev;                   // pre_integrations[WINDOW_SIZE] (running) and [WINDOW_SIZE - 1] (may take cur's samples, estimator.cpp:1069-1085)
        bool merge_pending = false;
        std::vector<int> ids;                 // the stored frame
        std::vector<double> rows;
        bool have_frame = false;
    };

    ResidentEstimators(int n, int max_features, int max_new_obs);
    ~ResidentEstimators();
    ResidentEstimators(const ResidentEstimators&) = delete;
    ResidentEstimators& operator=(const ResidentEstimators&) = delete;

    // Hand-over of estimator i between two frames (after slideWindow()): its states, pre_integrations[1 .. WINDOW_SIZE - 1] (the
    // last one with its raw samples), f_manager.feature and last_marginalization_info; acc_0 / gyr_0 = the newest IMU sample.
    void handOver(int i, Estimator& e, const Vector3d& acc_0, const Vector3d& gyr_0);
    void begin();                             // vg_ba_seq_begin once every estimator has been handed over
    // The way back, between two frames: window i of the running sequence into the members of a host Estimator (Ps / Rs / Vs / Bas /
    // Bgs, ric / tic / td, pre_integrations[1 .. WINDOW_SIZE - 1] -- the last one with its raw samples --, f_manager.feature,
    // last_marginalization_info + parameter blocks): vg_ba_seq_export + vg_ba_seq_get_tracks.  For a relocalisation frame (not
    // offered in a sequence), a checkpoint, a re-start after failureDetection().  The device keeps its copy; reseed() replaces it.
    void handBack(int i, Estimator& e);
    void reseed(int i, Estimator& e, const Vector3d& acc_0, const Vector3d& gyr_0);     // vg_ba_seq_import: estimator i only
    void processIMU(int i, double dt, const Vector3d& linear_acceleration, const Vector3d& angular_velocity);
    void processImage(int i, const Image& image);
    void solve();                             // one frame for every estimator (all must have received their image)
    One& operator[](int i) { return est_[i]; }
    int size() const { return (int)est_.size(); }

  private:
    struct Window;                            // the hand-over arrays of one estimator (alive until begin())
    Window* pack(int i, Estimator& e, const Vector3d& acc_0, const Vector3d& gyr_0);
    std::vector<One> est_;
    std::vector<Window*> win_;
    vg_handle* vg_ = nullptr;
    int max_features_, max_new_obs_;
    bool begun_ = false;
};
