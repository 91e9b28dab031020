// vg_handle.h — the opaque handle behind include/vinsgpu.h (host only).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "ba_layout.h"

// pinned host staging that grows on demand (hipHostMalloc: the D2H / H2D copies run as DMA at the PCIe rate)
template <typename T>
struct PinnedBuf {
    T* p = nullptr;
    size_t cap = 0, n = 0;
    hipError_t resize(size_t need) {
        n = need;
        if (need <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipHostMalloc((void**)&p, need * sizeof(T), hipHostMallocDefault);
        if (e == hipSuccess) cap = need;
        return e;
    }
    T* data() { return p; }
    size_t size() const { return n; }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = n = 0; }
};

// Windows that stay on the device between frames (vg_ba_seq_*, csrc/ba_seq.hip): what the kernels get by value
struct SeqDev {
    int K, FT, NIN;                  // frames per window, capacity of the track table, capacity of one frame's observation list
    int fi_stride, fd_stride;        // track table per window: ints [SEQ_HDR | id | start | nobs | solve_flag | landmark][FT], doubles [depth[FT] | obs[FT][K][8]]
    int ii_stride, id_stride;        // frame staging per window: ints [n_obs, has_merged, valid_new, valid_merged, ... | id[NIN]], doubles [pose 7 | sb 9 | imu_new 472 | imu_merged 472 | rows NIN x 8]
    int sp_stride;                   // prior block table per window: [n, nblocks | kind[K+4] | idx[K+4]]
    int max_iters, marg_mode;
    double init_depth, min_parallax;
    int* ft_i[2];                    // two tables: the slide writes the other one
    double* ft_d[2];
    int* sp;
    int* in_i;
    double* in_d;
    int* info;                       // [nwin][VG_SEQ_INFO_INTS]
};
struct BaSeq {
    bool active = false;
    int nwin = 0, cur = 0;
    SeqDev D = {};
    PinnedBuf<int> h_in_i, h_info;
    PinnedBuf<double> h_in_d;
};

struct BaBatch {
    BaLayout L;
    BaLayout* dL = nullptr;          // device copy of L (kernels read it through scalar loads) + the gather plan behind it
    size_t dL_bytes = 0;
    BaLayout dL_host;                // what dL holds (re-sent only when the layout changes)
    bool dL_valid = false;
    BaPtrs P = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t cap_ia = 0, cap_di = 0, cap_sc = 0, cap_out = 0, cap_iout = 0, cap_mout = 0, cap_miout = 0, cap_mscr = 0, cap_rb1 = 0, cap_rb2 = 0;
    std::vector<int> margin, nL;
    PinnedBuf<int> h_iout, h_miout;  // download staging
    PinnedBuf<double> h_out, h_mout;
    int* h_ia = nullptr;             // pinned host staging of the packed batch (hipHostMalloc: DMA at full PCIe rate)
    double* h_di = nullptr;
    size_t hcap_ia = 0, hcap_di = 0;
    int nwin = 0;
    int rounds = 0;                  // launches of the linearise / accumulate / solve triple = max over windows of max_iters
    bool uploaded = false, any_margin = false;
    bool solved_recorded = false;    // ev_fork recorded behind ba_final_kernel of the current run
    bool force_large = false;        // vg_ba_set_large_window: take the large-window path whatever the size
    int marg_mode = 0;               // vg_ba_set_marg_mode: VG_MARG_SQRT (default) / VG_MARG_EIGEN
    int imu_info_mode = 0;           // vg_ba_set_imu_info_mode: VG_IMU_INFO_FACTOR (default) / VG_IMU_INFO_REFERENCE
    int res_L = 0, res_F = 0, res_O = 0, res_N = 0;   // vg_ba_reserve: capacities every layout is built for at least
    int fused_min = -1;              // vg_ba_set_fused_min_windows (-1: environment VG_BA_FUSED_MIN, else 32; 0: never)
    bool no_env = false;             // handle made by vg_create_config: no environment variable shapes its behaviour
    int pack_threads = 0;            // host threads sharing the packing of a batch (0: environment VG_PACK_THREADS, else 8)
    BaSeq seq;                       // vg_ba_seq_*
    void* allreduce = nullptr;       // vg_allreduce_fn of the large-window path (nullptr: single rank)
    void* allreduce_user = nullptr;
    // ---- prior factors on the device (BaPtrs::pri).  slot[w] describes what window slot w holds: the block table of the last
    //      prior uploaded or carried there.  A run that marginalizes leaves its result in mout / miout; the first upload that
    //      asks for a resident prior (VG_PRIOR_RESIDENT) moves every valid result into the slots (carry kernel).
    struct PriorSlot { int n = 0, nb = 0; std::vector<int> kind, idx; };
    std::vector<PriorSlot> slot;
    int slot_K = 0, slot_po_r0 = 0, slot_pld = 0, slot_pstride = 0;   // layout the slots were written with
    size_t cap_pri = 0;
    double* h_pri = nullptr;         // pinned staging of the priors that come from the host
    size_t hcap_pri = 0;
    bool mout_pending = false;       // the last run's marginalization result has not been carried yet
    std::vector<int> run_margin;     // margin flags + output layout of that run
    int run_nwin = 0, run_K = 0, run_mo_J0 = 0, run_mo_r0 = 0, run_mo_x0 = 0, run_mo_stride = 0, run_mi_stride = 0, run_mcap = 0;
    // ---- the solve pipeline as a hipGraph (vg_ba_set_launch_mode).  The ~36 launches of one batch solve have the same kernel
    //      arguments from run to run (device layout block, buffer pointers, cost_only flags): captured once per (pointers, grid
    //      sizes, rounds) key and replayed with one hipGraphLaunch; a re-allocation or a different factor-count class re-captures.
    struct GraphKey {
        const void* dL = nullptr;
        BaPtrs P = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        int dims[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    };
    int launch_mode = -1;            // VG_LAUNCH_*; -1: not resolved yet (environment VG_BA_LAUNCH_MODE, else the default)
    hipGraphExec_t gexec = nullptr;
    GraphKey gkey;
    bool graph_unavailable = false;  // stream capture failed once on this handle: stay with direct launches
    long long n_graph_launches = 0, n_graph_captures = 0;
    double flops = 0, flops_marg = 0, bytes_in = 0, bytes_out = 0;
    double flops_k[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // algorithmic flops per kernel class (VG_BA_KERNEL_*), one run of the batch
};

struct FeState;   // fe_host.hip

struct vg_handle {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    hipStream_t aux = nullptr;                    // second stream: the IMU / prior linearisation runs beside the projection factors
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string err;
    BaBatch ba;
    FeState* fe = nullptr;
    void* rccl_comm = nullptr;                    // ncclComm_t of vg_ba_rccl_init (csrc/vg_rccl.hip)
    void* imu_buf = nullptr;                      // device scratch of vg_imu_preintegrate (grown on demand)
    size_t imu_cap = 0;
    void* ransac_buf = nullptr;                   // device scratch of vg_fe_reject_with_f (fixed size, allocated on first use)
};
