/* vinsgpu.h — C-ABI of libvinsgpu.so: MI355X (gfx950) implementation of VINS-Mono's two compute
 * hot paths.  Plain C, POD structs, raw pointers, int status codes; no torch / Eigen / OpenCV / ROS
 * types cross this boundary.
 *
 * The reference (HKUST-Aerial-Robotics/VINS-Mono) has no FFI/plugin layer; the seam this ABI
 * replaces is two C++ member functions whose state lives in public members:
 *
 *   BA:  void Estimator::optimization()                 vins_estimator/src/estimator.h:47
 *        (body: estimator.cpp:670-1003; state crossing the seam: estimator.h:65-138)
 *   FE:  void FeatureTracker::readImage(const cv::Mat&, double)   feature_tracker/src/feature_tracker.h:33
 *        (body: feature_tracker.cpp:81-167; state: feature_tracker.h:49-62)
 *
 * Every function returns VG_OK (0) or a negative vg_status.  A handle is bound to the HIP device
 * that was current when it was created and must be used by one host thread at a time.
 * Host-pointer entry points are synchronous on return.  The *_async / *_dev entry points only
 * enqueue work on the handle's stream (or the stream given) — pair them with vg_sync().
 */
#ifndef VINSGPU_H
#define VINSGPU_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VG_ABI_VERSION 12   /* 2: vg_ba_problem::max_solver_time_s, large-window / all-reduce entry points; 3: vg_ba_summary::gauge_*;
                             * 4: VG_PRIOR_RESIDENT; 5: vg_ba_set_launch_mode; 6: vg_ba_reserve, vg_ba_seq_* (windows that stay on the device);
                             * 7: vg_ba_seq_export / vg_ba_seq_import; 8: vg_host_register, vg_ba_set_fused_min_windows, vg_ba_batch_is_fused;
                             * 9: vg_fe_keep_eig (the min-eigenvalue map is no longer written unless asked for);
                             * 10: vg_config / vg_create_config; vg_ba_batch_is_fused no longer returns 2; 11: vg_fe_read_image;
                             * 12: vg_config::device is 0 = current device / k + 1 = device k, vg_config::imu_info_mode, vg_ba_set_imu_info_mode */
#define VG_MAX_ITERS 32          /* capacity of the per-iteration trace in vg_ba_summary */

typedef enum {
    VG_OK = 0,
    VG_ERR_BAD_ARG = -1,         /* null pointer, negative size, inconsistent tables          */
    VG_ERR_HIP = -2,             /* a HIP runtime call failed (see vg_last_error)             */
    VG_ERR_UNSUPPORTED = -3,     /* problem does not fit the single-workgroup LDS fast path   */
    VG_ERR_NUMERIC = -4,         /* non-finite state / indefinite reduced system on device    */
    VG_ERR_NO_DEVICE = -5
} vg_status;

typedef struct vg_handle vg_handle;

/* ---- lifecycle ------------------------------------------------------------------------------ */
int vg_abi_version(void);
int vg_create(vg_handle** out);                 /* uses the current HIP device, owns one stream  */
/* The same with everything that shapes a handle's behaviour in ONE struct (ABI 10; SURVEY 8(b): vg_create(const vg_config*, ...)).
 * A handle created this way NEVER consults the environment: the development variables VG_BA_LAUNCH_MODE, VG_BA_FUSED,
 * VG_BA_FUSED_MIN, VG_BA_SOLVE_W8_BELOW and VG_PACK_THREADS are defaults of plain vg_create() only (a drop-in library must not change behaviour with
 * its host's environment).  Zero-initialise, set struct_size = sizeof(vg_config), fill what differs from the defaults. */
typedef struct vg_config {
    int struct_size;             /* sizeof(vg_config) of the caller: fields beyond it take their defaults                         */
    int device;                  /* 0 (or negative) = the CURRENT HIP device; k + 1 = device k (hipSetDevice(k) on the caller's
                                  * thread before the streams are created).  ABI 12: a zero-initialised struct no longer means device 0 */
    int launch_mode;             /* VG_LAUNCH_DIRECT + 1 / VG_LAUNCH_GRAPH + 1; 0 = the library's default (vg_ba_set_launch_mode)   */
    int marg_mode;               /* VG_MARG_SQRT (0, default) / VG_MARG_EIGEN: form of the prior factor (vg_ba_set_marg_mode)     */
    int fused_min_windows;       /* batches from this many windows on take the fused factor kernel; 0 = default (32), -1 = never  */
    int pack_threads;            /* host threads that share the packing / unpacking of a batch; 0 = default (8)                   */
    int imu_info_mode;           /* VG_IMU_INFO_FACTOR (0, default) / VG_IMU_INFO_REFERENCE: form of the IMU sqrt_info (vg_ba_set_imu_info_mode) */
} vg_config;
int vg_create_config(const vg_config* cfg, vg_handle** out);
/* What core clock does this box really run at?  out4 = { ns per dependent FP64 FMA with ONE wavefront on the chip, clock64() ticks per
 * microsecond of wall time during that, the same two with every CU loaded }.  The FMA latency is a constant number of core cycles, so
 * the first and third figures are inversely proportional to the clock (lone / under load).  bench.py reports them (`device.clock_probe`). */
int vg_probe_clocks(vg_handle* h, double* out4);
int vg_destroy(vg_handle* h);
int vg_sync(vg_handle* h);                      /* hipStreamSynchronize(handle stream)           */
const char* vg_last_error(vg_handle* h);        /* text of the last failure on this handle       */
/* ABI 8.  Page-lock a host buffer the caller will hand to the upload entry points again and again (an image ring buffer, a staging
 * area): hipHostRegister / hipHostUnregister.  Uploads from registered memory run at the PCIe rate instead of through the runtime's
 * pageable-copy staging (measured for 256 frames of 752x480: bench.py fe.upload_inclusive).  Optional: every entry point takes
 * pageable memory too. */
int vg_host_register(vg_handle* h, void* p, size_t bytes);
int vg_host_unregister(vg_handle* h, void* p);
void* vg_stream(vg_handle* h);                  /* the handle's hipStream_t (as void*)           */
/* HIP-event stopwatch on the handle's stream (used by bench.py for roofline.achieved) */
int vg_timer_start(vg_handle* h);
int vg_timer_stop(vg_handle* h, float* elapsed_ms);   /* records, synchronises, returns ms      */

/* =============================================================================================
 * BA — Estimator::optimization()  (estimator.cpp:670-1003)
 * ============================================================================================= */

/* IMU pre-integration constants of one frame pair, i.e. the members of IntegrationBase read by
 * IMUFactor::Evaluate (factor/imu_factor.h:19-179, factor/integration_base.h:160-186). */
typedef struct {
    double sum_dt;
    double delta_p[3];
    double delta_q[4];           /* x y z w */
    double delta_v[3];
    double linearized_ba[3];
    double linearized_bg[3];
    double jacobian[225];        /* 15x15 row-major, order [p, theta, v, ba, bg] (parameters.h:48-55) */
    double covariance[225];      /* 15x15 row-major */
    int valid;                   /* 0: no factor for this pair (estimator.cpp:714: sum_dt > 10) */
    int _pad;
} vg_imu_preint;

/* Batched IMU pre-integration — replaces IntegrationBase::push_back / propagate / midPointIntegration and, with new
 * linearisation biases, IntegrationBase::repropagate (factor/integration_base.h:30-158), as driven by
 * Estimator::processIMU (estimator.cpp:93-101) and the bias-change re-propagation of estimator.cpp:600-609.
 * Interval k integrates the samples [sample_off[k], sample_off[k+1]) — rows (dt, acc_x, acc_y, acc_z, gyr_x, gyr_y,
 * gyr_z) of `samples` — starting from the measurement first[k] = (acc_0, gyr_0) (IntegrationBase ctor,
 * integration_base.h:15-27) with the biases bias[k] = (linearized_ba, linearized_bg); noise = (ACC_N, GYR_N, ACC_W,
 * GYR_W) of the estimator configuration.  out[k].valid is set to 1 (the sum_dt > 10 s rule of estimator.cpp:714 is
 * applied where the factor is used).  SURVEY.md 8(f) row 2. */
int vg_imu_preintegrate(vg_handle* h, int n_intervals, const int* sample_off, const double* samples, const double* first,
                        const double* bias, const double* noise, vg_imu_preint* out);

/* FeatureManager::triangulate (feature_manager.cpp:202-257) for L landmarks whose depth is
 * not yet known (the used_num >= 2 && start_frame < WINDOW_SIZE - 2 && estimated_depth <= 0 filter of :206-211 is the
 * caller's).  Ps [K x 3], Rs [K x 9 row-major] = body poses of the window, tic / ric = camera extrinsics; landmark l is
 * observed in frames start[l] .. start[l] + nobs[l] - 1 with feature_per_frame.point = points[obs_off[l] + j] (x y z).
 * depth[l] = svd_V[2] / svd_V[3] of the 2n x 4 DLT system, or init_depth (INIT_DEPTH) when that is < 0.1 (:245-254). */
int vg_triangulate(vg_handle* h, int K, const double* Ps, const double* Rs, const double* tic, const double* ric, int L,
                   const int* start, const int* nobs, const int* obs_off, const double* points, double init_depth, double* depth);

/* block kinds of the marginalization prior (MarginalizationInfo::keep_block_*) */
enum { VG_BLK_POSE = 0, VG_BLK_SPEEDBIAS = 1, VG_BLK_EXPOSE = 2, VG_BLK_TD = 3 };
enum { VG_MARGIN_OLD = 0, VG_MARGIN_SECOND_NEW = 1, VG_MARGIN_NONE = 2 };

/* The optimisation problem exactly as Estimator::optimization() assembles it from
 * para_Pose / para_SpeedBias / para_Ex_Pose / para_Feature / para_Td (estimator.cpp:486-528),
 * f_manager.feature (:719-764), pre_integrations[] (:711-718) and last_marginalization_info (:703-709).
 * All arrays are caller-owned host memory, read-only. */
#define VG_PRIOR_RESIDENT (-1)
typedef struct {
    int K;                       /* frames in the window = WINDOW_SIZE + 1                        */
    int L;                       /* landmarks that pass used_num>=2 && start_frame<WINDOW_SIZE-2  */
    int n_obs;                   /* rows of `obs`                                                 */
    const double* pose;          /* K x 7  [px py pz qx qy qz qw]                                  */
    const double* speedbias;     /* K x 9  [v ba bg]                                               */
    const double* ex_pose;       /* 7      camera-in-IMU extrinsic                                 */
    double td;
    const double* inv_depth;     /* L      inverse depth in the landmark's first observing frame   */
    const int* lm_start;         /* L      start_frame                                             */
    const int* lm_nobs;          /* L      feature_per_frame.size()  (consecutive frames)          */
    const int* lm_obs_off;       /* L      first row of this landmark in `obs`                     */
    const double* obs;           /* n_obs x 7  [x y u v vx vy cur_td]; x,y normalised (z = 1)      */
    const vg_imu_preint* imu;    /* K-1    factor k links frame k -> k+1                           */
    /* prior (MarginalizationFactor, marginalization_factor.cpp:321-381); prior_n == 0: none.
     * prior_n == VG_PRIOR_RESIDENT: the prior this window slot (index in the batch) holds ON THE DEVICE -- the result of the
     * marginalization of the handle's last run for that slot if it produced one (flag != VG_MARGIN_NONE, vg_ba_prior.valid),
     * else the prior the slot was last given.  The reference keeps last_marginalization_info in place between two calls of
     * optimization() (estimator.cpp:703-709, :836-870); this is the same thing for a prior that lives in HBM: nothing of it
     * crosses the host boundary (download the states with out_priors = NULL).  The other prior_* fields are ignored.  Needs
     * the same K and the same batch slot as the run that produced it; VG_ERR_BAD_ARG if the slot holds nothing.          */
    int prior_n;                 /* rows of the prior = sum of local block sizes                   */
    int prior_nblocks;
    const int* prior_block_kind; /* VG_BLK_*                                                       */
    const int* prior_block_index;/* frame index for POSE / SPEEDBIAS, 0 otherwise                  */
    const double* prior_J0;      /* prior_n x prior_n row-major linearized_jacobians               */
    const double* prior_r0;      /* prior_n           linearized_residuals                          */
    const double* prior_x0;      /* concatenated keep_block_data (global sizes 7/9/7/1)            */
    /* relocalisation factors (estimator.cpp:769-801); relo_n == 0: none */
    int relo_n;
    const double* relo_pose;     /* 7                                                              */
    const int* relo_lm;          /* relo_n  landmark index                                         */
    const double* relo_xy;       /* relo_n x 2 matched normalised point in the loop frame          */
    /* options */
    int estimate_extrinsic;      /* 0: ex_pose constant (SetParameterBlockConstant, :686-689)      */
    int estimate_td;             /* 1: ProjectionTdFactor + td block (:694-698, :741-746)          */
    int max_iters;               /* NUM_ITERATIONS                                                  */
    double focal;                /* FOCAL_LENGTH: sqrt_info = focal/1.5 * I (estimator.cpp:17-18)   */
    double tr;                   /* TR  rolling-shutter read-out time                               */
    double row;                  /* ROW image height                                                */
    double g_norm;               /* G = (0,0,g_norm)                                                */
    double max_solver_time_s;    /* SOLVER_TIME (estimator.cpp:812-815: options.max_solver_time_in_seconds); 0 = no
                                  * cap.  Checked on the device clock before every trust-region iteration, like Ceres does
                                  * on the host clock: termination stays NO_CONVERGENCE.  A cap makes the result depend on
                                  * timing; it is ignored while an all-reduce hook is installed (ranks must agree).        */
} vg_ba_problem;

/* Optimised state, AFTER Estimator::double2vector()'s gauge fix and the vector2double() repack
 * (estimator.cpp:530-619, :486-528).  Caller-owned buffers sized as in vg_ba_problem. */
typedef struct {
    double* pose;                /* K x 7 */
    double* speedbias;           /* K x 9 */
    double* ex_pose;             /* 7     */
    double* td;                  /* 1     */
    double* inv_depth;           /* L     (a negative value => FeatureManager::setDepth marks solve_flag = 2) */
    double* relo_pose;           /* 7 or NULL */
} vg_ba_state;

enum { VG_TERM_NO_CONVERGENCE = 0, VG_TERM_CONVERGENCE = 1, VG_TERM_FAILURE = 2 };

typedef struct {
    int status;                  /* vg_status of this window                                      */
    int termination;             /* VG_TERM_*                                                     */
    int num_iterations;          /* trust-region iterations performed (accepted + rejected)       */
    int num_accepted;
    double initial_cost;
    double final_cost;
    double final_radius;
    /* per-iteration trace (first num_iterations entries) */
    double it_cost[VG_MAX_ITERS];       /* cost at the current point before the step             */
    double it_cost_cand[VG_MAX_ITERS];  /* cost at the candidate                                   */
    double it_model[VG_MAX_ITERS];      /* model cost change                                       */
    double it_radius[VG_MAX_ITERS];     /* trust-region radius used                                */
    double it_step_norm[VG_MAX_ITERS];  /* dogleg step norm (scaled space)                         */
    int it_flags[VG_MAX_ITERS];         /* bit0 valid, bit1 accepted                               */
    double prof[16];                    /* device phase timers (shader cycles of lane 0); zero unless the library
                                           was built with -DBA_PROFILE                                  */
    /* the gauge transform double2vector() applied (estimator.cpp:541-577): x_fixed = gauge_rot (x - gauge_p0) + Ps[0]_before,
     * R_fixed = gauge_rot R; gauge_rot = rot_diff (3x3 row-major), gauge_p0 = para_Pose[0] position right after the solve.
     * For quantities outside the window, e.g. relo_Pose when no matched landmark put it into the problem (:598-603).   */
    double gauge_rot[9];
    double gauge_p0[3];
} vg_ba_summary;

/* New marginalization prior (MarginalizationInfo after marginalize() + getParameterBlocks(),
 * marginalization_factor.cpp:174-319), blocks already re-labelled for the slid window
 * (addr_shift, estimator.cpp:913-930 / :969-996).  Caller allocates J0[cap*cap], r0[cap],
 * x0[9*cap_blocks] (global block sizes: 7 pose / 9 speed-bias / 7 extrinsic / 1 td), block_kind/index[cap_blocks] and sets
 * cap / cap_blocks. */
typedef struct {
    int cap;                     /* in: capacity (rows) of J0 / r0                                */
    int cap_blocks;              /* in: capacity of the block arrays                              */
    int n;                       /* out: kept dimension  (0: no prior produced)                   */
    int m;                       /* out: marginalised dimension                                   */
    int nblocks;                 /* out */
    int valid;                   /* out: 1 if a new prior was produced, 0 if the old one stays    */
    int* block_kind;
    int* block_index;
    double* J0;                  /* n x n row-major */
    double* r0;
    double* x0;                  /* concatenated, global sizes */
} vg_ba_prior;

/* One synchronous Estimator::optimization(): solve + gauge fix (+ marginalization if
 * margin_flag != VG_MARGIN_NONE; out_prior may be NULL then). */
int vg_ba_optimize(vg_handle* h, const vg_ba_problem* in, int margin_flag,
                   vg_ba_state* out_state, vg_ba_summary* out_summary, vg_ba_prior* out_prior);

/* Batch of independent windows (BASELINE.json configs[3]); staged so that benchmarks can time the
 * device work alone with inputs resident in HBM:
 *   upload   : pack + H2D (synchronous on return)
 *   run_async: enqueue the solve pipeline (+ marginalization kernel) for every uploaded window
 *   download : D2H + unpack (synchronises first)
 * All windows of a batch must share K, estimate_extrinsic, estimate_td and relo presence. */
int vg_ba_batch_upload(vg_handle* h, int nwin, const vg_ba_problem* const* in, const int* margin_flags);
int vg_ba_batch_run_async(vg_handle* h);
/* same as run_async but synchronous and timed with HIP events on the handle's stream: the solve pipeline (all its
 * launches) and the marginalization kernel */
int vg_ba_batch_run_timed(vg_handle* h, float* solve_ms, float* marg_ms);
int vg_ba_batch_download(vg_handle* h, int nwin, vg_ba_state* const* out_states,
                         vg_ba_summary* out_summaries, vg_ba_prior* const* out_priors);
/* The download in two parts.  The states and summaries are final as soon as the solve pipeline has finished; the
 * marginalization kernel that follows on the stream only reads them, and its result is consumed by the NEXT frame's
 * optimization (estimator.cpp:703-709).  download_state returns while the marginalization is still running (it waits for
 * an event recorded behind the solve pipeline and copies on a second stream); download_prior waits for the whole stream.
 * Collect the priors before the handle's next upload. */
int vg_ba_batch_download_state(vg_handle* h, int nwin, vg_ba_state* const* out_states, vg_ba_summary* out_summaries);
int vg_ba_batch_download_prior(vg_handle* h, int nwin, vg_ba_prior* const* out_priors);
/* vg_ba_optimize in the same two parts (one window): begin = upload + run + download_state */
int vg_ba_optimize_begin(vg_handle* h, const vg_ba_problem* in, int margin_flag, vg_ba_state* out_state, vg_ba_summary* out_summary);
int vg_ba_optimize_prior(vg_handle* h, vg_ba_prior* out_prior);
/* one synchronous run with a HIP event after every launch of the solve pipeline: ms[k] = summed duration and n[k] =
 * number of launches of kernel class k (both arrays VG_BA_KERNEL_COUNT long) */
enum { VG_BA_KERNEL_PROLOGUE = 0, VG_BA_KERNEL_LINEARIZE, VG_BA_KERNEL_ACCUMULATE, VG_BA_KERNEL_SOLVE, VG_BA_KERNEL_FINAL,
       VG_BA_KERNEL_MARG,
       /* large-window path: landmark Schur kernel; reduced solve (+ the preceding all-reduce); dogleg step (+ its all-reduce) */
       VG_BA_KERNEL_BIG_SCHUR, VG_BA_KERNEL_BIG_SOLVE, VG_BA_KERNEL_BIG_STEP, VG_BA_KERNEL_COUNT };
int vg_ba_batch_run_profiled(vg_handle* h, float* ms, int* n);
/* the SURVEY.md 8(d) flop model of one run of the uploaded batch, split per kernel class (VG_BA_KERNEL_COUNT doubles) */
int vg_ba_batch_flops_by_kernel(vg_handle* h, double* flops);
/* algorithmic work of the uploaded batch for roofline accounting (SURVEY.md 8(d) flop model) */
int vg_ba_batch_info(vg_handle* h, double* flops_per_run, double* bytes_in, double* bytes_out, int* lds_bytes);
/* the same flop model split per launch: ba_solve_kernel (solve + prior J^T J) and ba_marg_kernel (the marginalization term) */
int vg_ba_batch_flops(vg_handle* h, double* solve_flops, double* marg_flops);

/* ---- Large windows and landmark shards (BASELINE.json configs[4]) --------------------------------
 * The reference fixes WINDOW_SIZE = 10 and NUM_OF_F = 1000 at compile time (vins_estimator/src/parameters.h:12,14) and
 * walks every factor in one thread (estimator.cpp:719-764); here both are runtime sizes.  A window whose camera part is
 * wider than the single-workgroup pipeline's tiling (more than ~13 frames) automatically takes the LARGE-WINDOW PATH: the
 * landmark Schur complement is formed by a multi-workgroup kernel into a reduce buffer, the reduced system is factorised
 * out of HBM / LDS by one workgroup.  The same path shards a window over ranks: every rank uploads the SAME frames, IMU
 * factors and prior but only ITS contiguous share of the landmarks (with their observations), installs an all-reduce
 * hook, and calls vg_ba_batch_run_async as usual (on this path the call waits for the stream once or a few times near the end of
 * the solve: the rounds that only exist for retried factorisations are issued on demand, after a look at the windows' DONE flags);
 * after the run every rank holds the identical frame states and the inverse depths of its own landmarks.  Marginalization: a large
 * window on ONE rank is marginalized like any other (margin_flags); with an all-reduce hook installed margin_flags must be NONE --
 * a MARGIN_OLD marginalization only involves the tracks anchored at frame 0, so the ranks all-gather those (a few KB) and every rank
 * marginalizes the same reduced single-rank problem (all frames at the solved states, imu[], the old prior, those tracks,
 * max_iters = 0) on a second handle without a hook: identical result everywhere (vins-mono_amd/shard.py marginalize_sharded).
 *
 * The hook is called on the host, twice per trust-region round, between two kernel launches: it must enqueue on `stream`
 * (a hipStream_t) an in-place SUM over all ranks of `count` doubles at `device_buf` -- with RCCL:
 *     ncclAllReduce(device_buf, device_buf, count, ncclDouble, ncclSum, comm, (hipStream_t)stream)
 * -- and return 0.  No host synchronisation is needed or wanted.  count is the same on every rank (it depends only on the
 * window's frame count and flags): vg_ba_reduce_layout reports both counts (reduced system; step norms). */
typedef int (*vg_allreduce_fn)(void* user, double* device_buf, size_t count, void* stream);
int vg_ba_set_allreduce(vg_handle* h, vg_allreduce_fn fn, void* user);       /* fn == NULL: single rank */
int vg_ba_set_large_window(vg_handle* h, int force);     /* force != 0: take the large-window path whatever the size */

/* The same hook bound to RCCL over xGMI inside the library (csrc/vg_rccl.hip; librccl is opened with dlopen at the first call,
 * VG_ERR_UNSUPPORTED if it cannot be).  One process per GPU: rank 0 calls vg_rccl_unique_id and hands the 128 bytes to the
 * other ranks by whatever channel the application has (torch.distributed / MPI broadcast); every rank then calls
 * vg_ba_rccl_init(handle, nranks, rank, id) — collective, it creates the communicator on the handle's device and installs
 * ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, launch stream) as the hook.  vg_ba_rccl_finalize (also run by
 * vg_destroy) removes the hook and destroys the communicator. */
#define VG_RCCL_ID_BYTES 128
int vg_rccl_unique_id(char* id128);
int vg_ba_rccl_init(vg_handle* h, int nranks, int rank, const char* id128);
int vg_ba_rccl_finalize(vg_handle* h);

/* How vg_ba_batch_run_async / vg_ba_optimize* issue the solve pipeline (prologue, max_iters x {linearise IMU + prior,
 * linearise projections, accumulate, solve}, cost pass, final: 4 * max_iters + 4 launches):
 *   VG_LAUNCH_DIRECT  one hipLaunchKernelGGL per kernel;
 *   VG_LAUNCH_GRAPH   the sequence is captured from the handle's stream into a hipGraph the first time a given combination of
 *                     device buffers, grid sizes and round count is run, and replayed by ONE hipGraphLaunch afterwards (the
 *                     kernel arguments of a batch do not change from run to run; a re-allocation or a different size class
 *                     re-captures).  Same kernels, same results, bit for bit.
 * Default: environment VG_BA_LAUNCH_MODE = "graph" | "direct", else VG_LAUNCH_DEFAULT.  The large-window path (all-reduce hook
 * between launches) and the profiled run always launch directly.  vg_ba_launch_stats reports the mode in effect and how many
 * graph launches / captures the handle has made. */
enum { VG_LAUNCH_DIRECT = 0, VG_LAUNCH_GRAPH = 1 };
#ifndef VG_LAUNCH_DEFAULT
#define VG_LAUNCH_DEFAULT VG_LAUNCH_DIRECT
#endif
int vg_ba_set_launch_mode(vg_handle* h, int mode);
int vg_ba_launch_stats(vg_handle* h, int* mode, long long* graph_launches, long long* graph_captures);
/* ABI 8.  From how many windows per batch on the factors of a round are linearised AND accumulated by ONE kernel with one workgroup per
 * window (ba_linacc_proj_kernel: IMU factors, prior and projection factors; the projection records stay in LDS, J^T J of the camera
 * blocks on MFMA) instead of by ba_linearize_imu / ba_linearize_proj / ba_accumulate spread over many workgroups per window.  With few
 * windows the spread form is faster (the chip is empty), with a full batch the fused one (+18 % solves/s at 256 windows).  Default:
 * environment VG_BA_FUSED_MIN, else 32; 0 = never.  Windows with extrinsic / td columns or on the large-window path always take the
 * spread form.  Takes effect at the next upload. */
int vg_ba_set_fused_min_windows(vg_handle* h, int min_windows);
int vg_ba_batch_is_fused(vg_handle* h);      /* uploaded batch: 0 = spread kernels; 1 = fused factor kernel (counted as VG_BA_KERNEL_ACCUMULATE) + solve kernel;
                                              * < 0: error.  (2 = one merged launch per round was an experiment of ABI 8-9; removed) */

/* Capacities the layout of every later batch is built for at least (landmarks, projection factors and observation rows per window,
 * rows of the prior): a caller whose windows fluctuate from frame to frame keeps one layout -- no re-allocation, and in
 * VG_LAUNCH_GRAPH mode no re-capture.  Zero leaves a capacity to the data.  Takes effect at the next upload. */
int vg_ba_reserve(vg_handle* h, int max_landmarks, int max_factors, int max_obs, int max_prior_n);

/* ---- Windows that stay on the device from frame to frame ------------------------------------------------
 * The reference keeps the sliding window in host members between two calls of optimization(): f_manager.feature (every track,
 * also the ones not yet in the problem), Ps / Rs / Vs / Bas / Bgs, pre_integrations[], last_marginalization_info.  Around
 * optimization() it runs, per frame (Estimator::processImage, estimator.cpp:120-215):
 *     f_manager.addFeatureCheckParallax   (feature_manager.cpp:45-107)   new observations, key-frame decision
 *     f_manager.triangulate               (feature_manager.cpp:202-257)  depth of the tracks that enter the problem
 *     optimization()                      (estimator.cpp:670-1003)       solve + marginalization
 *     slideWindow()                       (estimator.cpp:1005-1126)      state / pre-integration shift,
 *                                         FeatureManager::removeBackShiftDepth / removeFront (feature_manager.cpp:275-351)
 *     f_manager.removeFailures            (feature_manager.cpp:161-171)
 * A sequence does all of that on the device for a batch of independent windows: the track tables, the states, the IMU constants
 * and the prior never leave HBM; per frame the host sends the new frame's observations (id + 7 doubles each), its state guess
 * (what processIMU propagated) and the pre-integration of the new interval, and reads back the states.  No per-frame packing,
 * no re-upload of the window.  Relocalisation factors and the large-window path are not offered in a sequence.              */
typedef struct {
    int n_features;              /* f_manager.feature.size(), list order                                                     */
    const int* feature_id;       /* FeaturePerId::feature_id                                                                 */
    const int* start_frame;
    const int* n_obs;            /* feature_per_frame.size() (consecutive frames from start_frame)                           */
    const int* solve_flag;       /* may be NULL (all 0)                                                                      */
    const double* depth;         /* estimated_depth (-1: not triangulated yet)                                               */
    const double* obs;           /* sum(n_obs) rows, feature-major: [x y z u v vx vy cur_td] (FeaturePerFrame)                */
} vg_ba_tracks;

typedef struct {
    int max_features;            /* capacity of a window's track table                                                       */
    int max_new_obs;             /* capacity of one frame's observation list                                                 */
    int max_landmarks;           /* tracks in the problem at once (0: max_features)                                           */
    int max_factors;             /* projection factors at once (0: 6 x max_landmarks)                                         */
    double init_depth;           /* INIT_DEPTH   (vins_estimator/src/parameters.cpp:3)                                       */
    double min_parallax;         /* MIN_PARALLAX (keyframe_parallax / FOCAL_LENGTH, parameters.cpp:79-80)                    */
} vg_ba_seq_config;

/* The new frame of one window: what Estimator::processImage receives (the `image` map in ascending feature id: one
 * observation per feature, rows [x y z u v vx vy]) plus what Estimator::processIMU has produced since the last frame: the
 * propagated state of the newest frame (Ps / Rs / Vs [WINDOW_SIZE], biases copied from the frame before) and
 * pre_integrations[WINDOW_SIZE].  imu_merged: after a frame that was NOT a key frame (margin flag VG_MARGIN_SECOND_NEW)
 * slideWindow folds the dropped interval's samples into pre_integrations[WINDOW_SIZE - 1] (estimator.cpp:1069-1085); the
 * caller re-integrates that interval (vg_imu_preintegrate over the concatenated samples) and hands it over here; NULL otherwise. */
typedef struct {
    double pose[7];
    double speedbias[9];
    const vg_imu_preint* imu_new;
    const vg_imu_preint* imu_merged;
    int n_obs;
    const int* feature_id;       /* ascending                                                                                */
    const double* obs;           /* n_obs x 7                                                                                */
} vg_ba_frame;

/* per-window record of the last step (vg_ba_seq_info) */
enum { VG_SEQ_FLAG = 0,          /* marginalization flag the key-frame test chose (VG_MARGIN_OLD: the frame before was a key frame) */
       VG_SEQ_N_FEATURES,        /* tracks in the table after the new observations were added                                 */
       VG_SEQ_N_TRACKED,         /* last_track_num                                                                            */
       VG_SEQ_N_PARALLAX,        /* parallax_num                                                                              */
       VG_SEQ_N_LANDMARKS,       /* tracks in the problem                                                                     */
       VG_SEQ_N_FACTORS,
       VG_SEQ_STATUS,            /* VG_OK or VG_ERR_UNSUPPORTED: a capacity of vg_ba_seq_config was exceeded (the step is void)  */
       VG_SEQ_N_AFTER,           /* tracks left after slideWindow + removeFailures                                            */
       VG_SEQ_INFO_INTS = 8 };

/* windows[w]: states of the K frames, imu[K-1], prior, options of window w as in vg_ba_optimize (its landmark tables are ignored;
 * frame K-1 and imu[K-2] are placeholders that the first step overwrites); tracks[w]: its track table.  This is the state
 * the reference is in between two frames (after slideWindow).  All windows share K and the estimate_* options. */
int vg_ba_seq_begin(vg_handle* h, int nwin, const vg_ba_seq_config* cfg, const vg_ba_problem* const* windows,
                    const vg_ba_tracks* const* tracks);
/* One frame for every window: add + key-frame test + triangulate + problem tables + solve + marginalization + slide, all
 * enqueued on the handle's stream. */
int vg_ba_seq_step_async(vg_handle* h, int nwin, const vg_ba_frame* const* frames);
/* States of the window as solved by the last step (before the slide) and its summaries: vg_ba_batch_download_state;
 * vg_ba_state::inv_depth, if not NULL, must hold max_landmarks rounded up to a multiple of 16 values (the first
 * info[VG_SEQ_N_LANDMARKS] are the problem's, in track-list order).  info: [nwin][VG_SEQ_INFO_INTS]. */
int vg_ba_seq_info(vg_handle* h, int nwin, int* info);
/* parity tap: the track table of one window as the last step left it; arrays of capacity cap (obs may be NULL; else
 * cap x K x 8 doubles, rows [x y u v vx vy cur_td z] of feature f at obs + (f * K + j) * 8) */
int vg_ba_seq_get_tracks(vg_handle* h, int window, int cap, int* n_features, int* feature_id, int* start_frame, int* n_obs,
                         int* solve_flag, double* depth, double* obs);
/* Hand-back and re-seed of ONE window of a running sequence, between two frames (after a step):
 *   export  the window of slot `window` in the form vg_ba_seq_begin takes it -- states (K x 7, K x 9, 7, 1), the K-1 pre-integration
 *           records (the newest one is the placeholder the next step fills: valid = 0), the prior (caller-allocated as for
 *           vg_ba_optimize; n = 0: none; after a step that chose VG_MARGIN_SECOND_NEW the record of interval K-3 is still the
 *           un-merged one -- the merged record reaches the device with the next frame's imu_merged) -- together with vg_ba_seq_get_tracks this is everything the reference keeps in
 *           Ps / Rs / Vs / Bas / Bgs, pre_integrations[], f_manager.feature and last_marginalization_info: a host Estimator can
 *           take the window back (a relocalisation frame, which a sequence does not offer; a checkpoint; a failure re-start);
 *   import  replaces what slot `window` holds (same K and estimate_* options as the sequence; prior from the host or none) while
 *           the other windows stay where they are.  Any output pointer of export may be NULL. */
int vg_ba_seq_export(vg_handle* h, int window, double* pose, double* speedbias, double* ex_pose, double* td, vg_imu_preint* imu,
                     vg_ba_prior* prior);
int vg_ba_seq_import(vg_handle* h, int window, const vg_ba_problem* in, const vg_ba_tracks* tracks);
int vg_ba_seq_end(vg_handle* h);

/* Form of the prior factor the marginalization hands back (marginalization_factor.cpp:285-296 builds J0 = S^1/2 V^T,
 * r0 = S^-1/2 V^T b' from the eigen-decomposition A' = V S V^T of the kept system).  Everything downstream uses the factor
 * only through J0^T J0, J0^T r0 and |r0|^2, which do not change under an orthogonal transformation from the left:
 *   VG_MARG_SQRT  (default)  J0 = L^T, r0 = L^-1 b' from the diagonally pivoted Cholesky factor of A', cut where no remaining
 *                            pivot exceeds eps = 1e-8 (the reference cuts eigenvalues at the same eps);
 *   VG_MARG_EIGEN            the reference's eigen form (rows ordered by ascending eigenvalue).
 * Takes effect at the next upload / vg_ba_optimize. */
enum { VG_MARG_SQRT = 0, VG_MARG_EIGEN = 1 };
int vg_ba_set_marg_mode(vg_handle* h, int mode);
/* ABI 12.  Form of the IMU factors' weight sqrt_info = LLT(covariance^-1).matrixL()^T (factor/imu_factor.h:64).  The Cholesky factor of
 * the inverse is unique, so both forms are the same matrix up to rounding -- the covariance is badly conditioned, and the rounding of
 * this factor is one of the inputs of the trust-region decisions (profiles/r06_flip_stats.json):
 *   VG_IMU_INFO_FACTOR     (default)  U^-1 of covariance = U U^T (U upper): the inverse is never formed;
 *   VG_IMU_INFO_REFERENCE             covariance.inverse() by partial-pivot LU, then the LLT of its lower triangle, as the reference
 *                                     spells it (operation order of oracle/ref_stubs' stand-in Eigen, no contraction).
 * Takes effect at the next upload / vg_ba_optimize. */
enum { VG_IMU_INFO_FACTOR = 0, VG_IMU_INFO_REFERENCE = 1 };
int vg_ba_set_imu_info_mode(vg_handle* h, int mode);
int vg_ba_reduce_layout(vg_handle* h, size_t* count_system, size_t* count_norms);

/* Batched factor evaluation for parity tests (rows B2-B5 of SURVEY.md 8(a)): evaluates every
 * projection / IMU / prior factor of the problem at its input state WITHOUT the robust-loss
 * correction and returns residuals and tangent-space Jacobians.
 *   proj_r  [F x 2], proj_J [F x 2 x 20] columns = [pose_i(6) pose_j(6) ex(6) lambda(1) td(1)]
 *   imu_r   [(K-1) x 15], imu_J [(K-1) x 15 x 30] columns = [pose_i(6) sb_i(9) pose_j(6) sb_j(9)]
 *   prior_r [prior_n]
 * F = sum(lm_nobs - 1) + relo_n.  Any output pointer may be NULL. */
int vg_ba_eval_factors(vg_handle* h, const vg_ba_problem* in, double* proj_r, double* proj_J,
                       double* imu_r, double* imu_J, double* prior_r);

/* =============================================================================================
 * FE — FeatureTracker::readImage()  (feature_tracker/src/feature_tracker.cpp:81-167)
 *
 * The handle holds `n_cams` independent camera streams (trackerData[NUM_OF_CAM] in
 * feature_tracker_node.cpp:21; also used as the batch dimension of BASELINE.json configs[1]).  Per stream the
 * device keeps the pyramids of the previous and the current frame (cur_img / forw_img of feature_tracker.h:52).
 *
 *   vg_fe_push_frames   <- `forw_img = img` incl. the optional CLAHE (:87-104)  + the pyramid build that
 *                          cv::calcOpticalFlowPyrLK does internally; the former current frame becomes previous
 *   vg_fe_track         <- cv::calcOpticalFlowPyrLK(cur_img, forw_img, cur_pts, forw_pts, status, err,
 *                          cv::Size(21,21), 3)                                    (:113)
 *   vg_fe_detect        <- cv::goodFeaturesToTrack(forw_img, n_pts, max_corners, 0.01, MIN_DIST, mask)  (:149)
 *
 * Staged variants (*_upload / *_async / *_download) let a benchmark time the device work with inputs resident in HBM.
 * ============================================================================================= */
int vg_fe_configure(vg_handle* h, int width, int height, int n_cams, int max_points);
/* imgs[cam] = top-left pixel of an 8-bit single-channel frame with row stride `stride` bytes; every stream needs a frame
 * (the batched streams advance together: a NULL entry is VG_ERR_BAD_ARG and leaves the device state untouched).  equalize != 0 applies CLAHE(3.0, 8x8) first (EQUALIZE of feature_tracker/src/parameters.cpp:59). */
int vg_fe_push_frames(vg_handle* h, const uint8_t* const* imgs, int stride, int equalize);
int vg_fe_upload_frames(vg_handle* h, const uint8_t* const* imgs, int stride);     /* H2D only            */
int vg_fe_build_async(vg_handle* h, int equalize);                                 /* CLAHE + pyramids    */
/* Two device-resident frame slots, one per pyramid set.  upload_frames fills the slot of the set the NEXT build will fill and
 * selects it; a frame that is not equalized then IS level 0 of its pyramid (no copy).  Consequence: after upload_frames the
 * older of the two pyramids is incomplete until build_async has run — upload, build, then track (push_frames does the first
 * two).  frame_slot returns the selected slot; select_frames re-selects a slot that was uploaded earlier (benchmarks alternate
 * between two resident frames; building from the slot the current pyramid already uses falls back to a copy). */
int vg_fe_frame_slot(vg_handle* h);
int vg_fe_select_frames(vg_handle* h, int slot);
/* Pyramidal LK from the previous to the current frame of stream `cam`.  prev_xy / next_xy are (x, y) float pairs.
 * status / err have OpenCV's meaning (the inBorder() filter of feature_tracker.cpp:115-117 is the caller's). */
int vg_fe_track(vg_handle* h, int cam, const float* prev_xy, int n, float* next_xy, uint8_t* status, float* err);
int vg_fe_track_upload(vg_handle* h, const float* prev_xy /* [n_cams][max_points][2] */, const int* n /* [n_cams] */);
int vg_fe_track_async(vg_handle* h);
int vg_fe_track_download(vg_handle* h, float* next_xy, uint8_t* status, float* err);
/* Shi-Tomasi corners of the CURRENT frame of stream `cam`; mask is height x width (0 = excluded) or NULL.
 * out_xy must hold max_corners pairs; corners are integer pixel positions in acceptance order. */
int vg_fe_detect(vg_handle* h, int cam, const uint8_t* mask, int max_corners, double quality, double min_dist,
                 float* out_xy, int* out_n);
int vg_fe_detect_upload(vg_handle* h, const uint8_t* const* masks /* per cam or NULL */, const int* max_corners);
int vg_fe_detect_async(vg_handle* h, double quality, double min_dist);
int vg_fe_detect_download(vg_handle* h, float* out_xy /* [n_cams][max_points][2] */, int* out_n /* [n_cams] */);
/* FeatureTracker::setMask() (feature_tracker.cpp:36-69) for all streams: the points of stream c
 * (pts_xy[c][i], track_cnt[c][i], i < n[c]; arrays are [n_cams][max_points]) are visited by track_cnt descending
 * (stable), a point is kept iff the mask at its rounded position is still 255, and every kept point blanks the filled
 * disc of `radius` (MIN_DIST) around it.  base_masks[c] = the FISHEYE mask (:38-41) or NULL (all 255); base_masks may
 * be NULL.  kept_index[c][k] = index of the k-th kept point, n_kept[c] their number.  The final mask stays on the
 * device as the mask of stream c for vg_fe_detect_masked(). */
int vg_fe_set_mask(vg_handle* h, const float* pts_xy, const int* track_cnt, const int* n, const uint8_t* const* base_masks,
                   int radius, int* kept_index, int* n_kept);
/* cv::goodFeaturesToTrack(forw_img, n_pts, max_corners, quality, min_dist, mask) (:149) with the mask left on the
 * device by vg_fe_set_mask: no mask upload */
int vg_fe_detect_masked(vg_handle* h, int cam, int max_corners, double quality, double min_dist, float* out_xy, int* out_n);
/* FeatureTracker::undistortedPoints() lifting (:258-271): PinholeCamera::liftProjective with the 8-step recursive
 * distortion model (camera_model PinholeCamera.cc:450-510, :646-661).  intr = fx fy cx cy k1 k2 p1 p2; out = (x/z, y/z)
 * as float (cv::Point2f). */
int vg_fe_undistort(vg_handle* h, const float* pts_xy, int n, const double* intr, float* out_xy);
/* FeatureTracker::rejectWithF() (feature_tracker.cpp:169-202): cv::findFundamentalMat(un_cur_pts, un_forw_pts, FM_RANSAC,
 * threshold, 0.99, status) on n >= 8 correspondences given in the pixel coordinates of the virtual pinhole camera
 * (FOCAL_LENGTH * x/z + COL/2, ...).  Follows OpenCV 3.3's registrators as recalled (oracle/ASSUMPTIONS.md F9): cv::RNG((uint64)-1)
 * sample schedule, 7-point solver, error = max of the two squared point-to-epipolar-line distances (float) <= threshold^2, adaptive
 * iteration bound, LMedS instead of RANSAC below 15 points.  status[i] = 1 for the inliers of the winning model; if no model is
 * found every status is 1 (nothing is rejected).  n_inliers / F_out (row-major 3x3, F33 = 1) may be NULL.  A pure function of the
 * input.  SURVEY.md 8(f) row 3. */
int vg_fe_reject_with_f(vg_handle* h, const float* cur_un_xy, const float* forw_un_xy, int n, double threshold, uint8_t* status,
                        int* n_inliers, double* F_out);
/* debugging / parity taps: copy a pyramid level of the current (which = 0) or previous (1) frame, or the
 * min-eigenvalue map of the last detect, to host memory.  The map (cv::cornerMinEigenVal inside cv::goodFeaturesToTrack,
 * feature_tracker.cpp:149) is an on-chip intermediate of the detection: vg_fe_keep_eig(h, 1) makes every FOLLOWING detection
 * write it to device memory as well (4 bytes per pixel and stream, allocated at that call); vg_fe_get_eig without it is
 * VG_ERR_BAD_ARG. */
int vg_fe_get_level(vg_handle* h, int cam, int which, int level, uint8_t* out, int* w, int* hgt);
int vg_fe_keep_eig(vg_handle* h, int on);
int vg_fe_get_eig(vg_handle* h, int cam, float* out);
int vg_fe_get_mask(vg_handle* h, int cam, uint8_t* out);          /* current device mask of the stream */

/* ---- One call per frame (ABI 11): FeatureTracker::readImage() of ONE stream (a handle configured with n_cams == 1) from `forw_img = img`
 * to undistortedPoints(), feature_tracker.cpp:81-167, without the host between the steps:
 *   upload (frame + cur_pts, one block) -> CLAHE -> pyramid -> calcOpticalFlowPyrLK -> status && inBorder -> reduceVector
 *   publish:      -> liftProjective of both point sets -> findFundamentalMat (RANSAC; sample schedule and iteration bounds resident on the
 *                    device) -> reduceVector -> [status download, `order` callback, order upload] -> setMask walk -> goodFeaturesToTrack
 *                    with MAX_CNT - kept corners -> addPoints -> liftProjective of the final list -> ONE download
 *   not publish:  -> liftProjective of the survivors -> ONE download
 * The one thing a published frame still asks the host in the middle is the ORDER of setMask's walk: the reference sorts by track_cnt with
 * std::sort (:48), which is not stable -- the order among equal counts is whatever the platform's sort makes of the sequence, so the
 * caller's own std::sort call decides it (feature_tracker_readimage.cpp passes a callback that runs the reference's sort).  order == NULL:
 * the list is walked as it stands -- which IS the stable order by count for a caller that keeps the reference's list discipline (kept
 * points in walk order, then new points with count 1: the counts are then non-increasing along the list at all times).
 * Two cases go back to the host inside the call (same results, more round trips): 8 <= survivors < 15 (findFundamentalMat switches to
 * LMedS) and a RANSAC sample OpenCV would have redrawn for collinearity (its schedule then depends on the points).
 * All pointers of vg_fe_frame_out point into pinned buffers of the handle and stay valid until the next vg_fe_* call on it.
 * The FIRST call on a stream builds its resident tables (RANSAC schedules for every point count up to max_points: 28 KB each, tens of
 * milliseconds of host time once); limits: n_cams == 1, max_points <= 2048, rejectWithF on at most 1024 tracking survivors. */
typedef struct vg_fe_frame_out {
    int n1;                      /* survivors of tracking + border test                                           (:115-124) */
    int n2;                      /* survivors of rejectWithF (== n1 when it did not run)                          (:193-198) */
    int ransac_ran;              /* publish && n1 >= 8                                                            (:171)     */
    int n_kept, n_new;           /* setMask's survivors, new corners (publish only)                               (:64, :149) */
    int n_final;                 /* length of the list the frame ends with: n_kept + n_new, or n1 when not published          */
    const uint8_t* status_lk;    /* [n]   calcOpticalFlowPyrLK status && inBorder(forw_pts[i])                               */
    const uint8_t* status_f;     /* [n1]  findFundamentalMat's mask over the tracking survivors (ransac_ran only)            */
    const float* forw_xy;        /* [n][2] tracked position of every input point                                             */
    const int* kept;             /* [n_kept] positions IN THE WALK ORDER of the points setMask kept                          */
    const float* new_xy;         /* [n_new][2] goodFeaturesToTrack's corners                                                 */
    const float* un_xy;          /* [n_final][2] liftProjective (x/z, y/z) of the final list: kept points in walk order, then new ones */
    int ransac_best, ransac_niters, fallback;   /* diagnostics: winning iteration, iterations that counted, RI_FB_* bits that sent the
                                                   estimate back to the host                                                 */
} vg_fe_frame_out;
/* called once per published frame after rejectWithF: `after` holds n1, n2, status_lk, status_f, forw_xy; write the walk order into
 * order[0 .. n2) as indices into the list of the n2 survivors (a permutation); return 0 (anything else aborts the frame with
 * VG_ERR_BAD_ARG).
 * Failure semantics (ABI 12): every error that follows from the arguments alone -- sizes, a point list without a previous frame -- is
 * returned BEFORE the frame is uploaded: the stream is untouched.  An error after that (the callback's, a detection overflow, a HIP
 * error) leaves the device stream one frame ahead of the caller; the callback must therefore not change the caller's own state
 * (work on copies, apply the statuses after VG_OK -- host/dropin/feature_tracker_readimage.cpp), and the caller re-starts the stream
 * (vg_fe_configure) before it tracks again. */
typedef int (*vg_fe_order_fn)(void* user, const vg_fe_frame_out* after, int* order);
typedef struct vg_fe_frame_in {
    int struct_size;             /* sizeof(vg_fe_frame_in) */
    const uint8_t* img;          /* the frame, width x height of vg_fe_configure */
    int stride;                  /* bytes per row */
    int equalize;                /* EQUALIZE */
    int publish;                 /* PUB_THIS_FRAME */
    const float* cur_xy;         /* [n][2] cur_pts */
    int n;
    int max_cnt;                 /* MAX_CNT  (<= max_points of vg_fe_configure) */
    int min_dist;                /* MIN_DIST */
    double quality;              /* goodFeaturesToTrack qualityLevel: 0.01 at :149 */
    double f_threshold;          /* F_THRESHOLD */
    double focal_length;         /* FOCAL_LENGTH (rejectWithF's virtual camera, :178) */
    double intr[8];              /* PinholeCamera: fx fy cx cy k1 k2 p1 p2 */
    const uint8_t* base_mask;    /* fisheye_mask (height x width, contiguous) or NULL; uploaded when the pointer changes */
    vg_fe_order_fn order;        /* or NULL */
    void* user;
} vg_fe_frame_in;
int vg_fe_read_image(vg_handle* h, const vg_fe_frame_in* in, vg_fe_frame_out* out);

#ifdef __cplusplus
}
#endif
#endif /* VINSGPU_H */
