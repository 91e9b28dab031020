// ba_pipeline.hip — sliding-window bundle adjustment on gfx950 (CDNA4) as a short pipeline of launches per
// trust-region iteration, every launch sized to the parallelism of its phase.
//
// Replaces the arithmetic under Estimator::optimization() (vins_estimator/src/estimator.cpp:670-1003): factor
// evaluation (factor/*.cpp,h), the Ceres trust-region solve configured at :803-818 (DENSE_SCHUR + DOGLEG + Jacobi
// scaling + Cauchy loss corrector; third-party behaviour restated in oracle/ASSUMPTIONS.md) and the gauge fix of
// double2vector() (:530-619).
//
//   ba_prologue_kernel   (1 workgroup / window)   IMU sqrt_info factors, J0^T J0 of the prior, solver state init
//   per round r = 0 .. max_iters-1:
//     ba_linearize_imu_kernel / ba_linearize_proj_kernel  (1 IMU + prior workgroup per window; tiles of 256 projection
//                           factors, whole batch in one grid) residuals + Jacobians at the point to be judged -> loss-corrected records, IMU
//                           Hessian blocks, prior residual, cost partials
//     ba_accumulate_kernel (one wavefront per 6x6 block of the camera system + landmark lanes) J^T J, J^T r of the
//                           projection factors, per-landmark h, b, W   — deterministic owner sums, no atomics
//     ba_solve_kernel      (1 workgroup / window) accept / reject the pending candidate, then the linear algebra of
//                           the next step: landmark Schur complement (v_mfma_f64_16x16x4), block-Thomas elimination
//                           of the speed-bias chain, dense Cholesky of the Rc x Rc camera part in LDS, back
//                           substitution, dogleg step, candidate state
//   the linearize kernels (cost only) + ba_final_kernel: judge the last candidate, gauge fix, outputs.
//
// The point evaluated in round r is the CANDIDATE of round r-1, linearised speculatively: if the solve kernel accepts
// it (the common case) its linearisation is already there; if it rejects, the Gauss-Newton step and gradient of the
// current point are reused exactly as DoglegStrategy does.  All solver state between launches lives in a small
// per-window control block in HBM; every sum has a fixed order, so results are bit-reproducible.
#include <hip/hip_runtime.h>
#include <vector>
#include <mutex>
#include <cstring>
#include <cstdlib>
#include "ba_layout.h"
#include "ba_factors.h"
#include "../../include/vinsgpu.h"

extern __shared__ __attribute__((aligned(16))) char bp_smem[];

// The per-round solve kernel is built twice from this one source (round 6): ba_solve_kernel on 4 wavefronts
// (two windows per CU: the form a batch runs) here, and ba_solve_w8_kernel on 8 wavefronts (one window per CU: the form a few
// windows on an otherwise empty chip run -- the drop-in's operating point) by ba_solve_w8.hip, which includes this file with
// SV_NT = 512 and BA_SOLVE_W8_TU defined: every phase function is then compiled a second time inside namespace vg_w8, the other
// kernels and the host code of this file are left out.  The phase functions take their thread count from SV_NT / SV_NW only.
#ifdef BA_SOLVE_W8_TU
#undef BA_PROFILE_DETAIL                    // (the development timers / dumps and their host entry points belong to the main unit)
#undef BA_DEBUG_DUMP
namespace vg_w8 {
#endif

#define NOINL __device__ __noinline__
// Pointers handed to a non-inlined phase function are generic, and generic accesses compile to flat_load / flat_store even
// when they always hit LDS (slower issue and latency than ds_read / ds_write, and they tie up both memory counters).  The
// phase functions of the single-workgroup solve therefore re-type their LDS operands explicitly.  (The CPU emulation of
// tests/simt has one address space.)
// (the address-space typedefs lds_d / lds_i / glb_d / glb_i: vg_target.h)
#define AS_LDS(p) ((lds_d*)(p))
#define AS_LDS_C(p) ((const lds_d*)(p))
#define AS_GLB(p) ((glb_d*)(p))
#define AS_GLB_C(p) ((const glb_d*)(p))
#define AS_GLB_CI(p) ((const glb_i*)(p))
// where the arrays of the solve carve that the large-window path keeps in HBM live: LDS (single-workgroup path) or HBM
template <bool BIG> struct MovT { typedef lds_d D; typedef lds_i I; };
template <> struct MovT<true> { typedef glb_d D; typedef glb_i I; };

#define LDSB ((double*)bp_smem)
typedef double double4_t __attribute__((vector_size(32)));      // v_mfma_f64_16x16x4 accumulator (4 VGPR pairs)

// optional phase timers of the solve kernel (build with -DBA_PROFILE): thread 0 accumulates s_memtime deltas per phase into
// the trace area of the output slab (vg_ba_summary.prof), summed over the rounds of a solve
#ifdef BA_PROFILE
#define PROF_DECL long long _pt = clock64(); double* _pf = out + L.oo_trace + 5 * VG_MAX_ITERS
#define PROF_ADD(id) do { if (c.tid == 0) { const long long _n = clock64(); _pf[id] += (double)(_n - _pt); _pt = _n; } } while (0)
#else
#define PROF_DECL
#define PROF_ADD(id)
#endif
// finer timers for development (build with -DBA_PROFILE_DETAIL): threads 0 and 128 of workgroup 0 accumulate clock deltas of
// the sub-phases of chain_schur / cholesky_aug / assemble_small into a device array read back by vg_debug_detail_profile
#ifdef BA_PROFILE_DETAIL
__device__ double g_dprof[128];     // [0, 32) thread 0, ids 0 .. 31 | [32, 64) thread 128 | [64, 128) the same for ids 32 .. 63
#define DP_DECL long long _dp = clock64()
#define DP_ADD(id) do { if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 128)) { const long long _n = clock64(); g_dprof[(threadIdx.x ? 32 : 0) + ((id) & 31) + ((id) >= 32 ? 64 : 0)] += (double)(_n - _dp); _dp = _n; } } while (0)
extern "C" int vg_debug_detail_profile(double* out64, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dprof), sizeof(double) * 64);
    if (e == hipSuccess && reset) { double z[64] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_dprof), z, sizeof(z)); }
    return e == hipSuccess ? 0 : -2;
}
extern "C" int vg_debug_detail_profile_hi(double* out64, int reset) {      // ids 32 .. 63
    hipError_t e = hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dprof), sizeof(double) * 64, sizeof(double) * 64);
    if (e == hipSuccess && reset) { double z[64] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_dprof), z, sizeof(z), sizeof(double) * 64); }
    return e == hipSuccess ? 0 : -2;
}
#else
#define DP_DECL
#define DP_ADD(id)
#endif
// development aid (build with -DBA_DEBUG_DUMP): window 0 copies its LDS image of the reduced system into a device array after each
// phase of its first iteration, read back by vg_debug_dump (tests/manual/dbg_dump.py compares the hardware with the emulator)
#ifdef BA_DEBUG_DUMP
#define DBG_SLOT 16384
__device__ double g_dbg[6 * DBG_SLOT];
__global__ void dbg_copy_kernel(double* out, int n) { for (int k = threadIdx.x; k < n; k += blockDim.x) out[k] = g_dbg[k]; }
extern "C" int vg_debug_dump(double* out, int n) {
    n = n < 6 * DBG_SLOT ? n : 6 * DBG_SLOT;
    double* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(double) * n) != hipSuccess) return -2;
    hipLaunchKernelGGL(dbg_copy_kernel, dim3(1), dim3(256), 0, 0, d, n);
    const hipError_t e = hipMemcpy(out, d, sizeof(double) * n, hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? 0 : -2;
}
#define DBG_DUMP(slot) do { if (blockIdx.x == 0 && s.it == 1 && s.mu <= 1e-8) { __syncthreads(); \
    for (int k_ = c.tid; k_ < L.lds_solve / 8; k_ += SV_NT) g_dbg[(slot) * DBG_SLOT + k_] = LDSB[k_]; \
    if ((slot) >= 2) for (int k_ = c.tid; k_ < 9 * L.K * L.ldc; k_ += SV_NT) g_dbg[5 * DBG_SLOT + k_] = m.xp[k_]; \
    __syncthreads(); } } while (0)
#else
#define DBG_DUMP(slot)
#endif
enum { PF_JUDGE = 0, PF_ASM, PF_DG, PF_BUILD, PF_CHAIN, PF_SCHUR, PF_CHOL, PF_BACK, PF_CBACK, PF_LMY, PF_NORMS, PF_CAND, PF_TAIL };

// (R-vectors of the solve kernel in LDS, V_G .. V_DI: ba_layout.h)

struct Ctx {
    const BaLayout* Lp;      // layout lives in device memory: uniform scalar loads on demand
    const int* hdr;
    const int* ia;           // int arrays of this window
    const double* di;        // double inputs
    const double* pri;       // prior factor [x0 | r0 | J0]
    double* sc;              // scratch
    int tid, lane, wave;
    int nL, nF, nprior, nblk;
    double focal, tr, row, gnorm;
};

DEV BaLayout layout_load(const BaLayout* Lp) { return const_load(Lp); }      // scalar loads (ba_math.h)

DEV void ctx_init(Ctx& c, const BaLayout* Lp, const BaPtrs& P, int w) {
    const BaLayout L = layout_load(Lp);
    c.Lp = Lp;
    c.ia = P.iarr + (size_t)w * L.istride;
    c.hdr = c.ia + L.io_hdr;
    c.di = P.din + (size_t)w * L.dstride;
    c.sc = P.scr + (size_t)w * L.sstride;
    c.pri = P.pri + (size_t)w * L.pstride;
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6;
    c.nL = uni(c.hdr[H_L]); c.nF = uni(c.hdr[H_F]); c.nprior = uni(c.hdr[H_NPRIOR]); c.nblk = uni(c.hdr[H_NBLK]);
    c.focal = uni(c.di[L.do_par + P_FOCAL]); c.tr = uni(c.di[L.do_par + P_TR]); c.row = uni(c.di[L.do_par + P_ROW]);
    c.gnorm = uni(c.di[L.do_par + P_GNORM]);
}

// deterministic workgroup-wide reductions through `red` (>= 2 * waves doubles of LDS), result uniform in every thread
DEV double block_sum(double* red, int nw, int lane, int wave, double v) {
    v = wave_sum_all(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += red[w];
    return uni(s);
}
DEV void block_sum2(double* red, int nw, int lane, int wave, double& a, double& b) {
    a = wave_sum_all(a);
    b = wave_sum_all(b);
    __syncthreads();
    if (lane == 0) { red[wave] = a; red[nw + wave] = b; }
    __syncthreads();
    double s = 0.0, t = 0.0;
    for (int w = 0; w < nw; ++w) { s += red[w]; t += red[nw + w]; }
    a = uni(s); b = uni(t);
}
DEV double block_max(double* red, int nw, int lane, int wave, double v) {
    v = wave_max_all(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = red[0];
    for (int w = 1; w < nw; ++w) s = fmax(s, red[w]);
    return uni(s);
}

DEV int col_pose(const BaLayout& L, int i) { return 6 * i; }
DEV int col_ex(const BaLayout& L) { return 6 * L.Kp; }
DEV int col_td(const BaLayout& L) { return 6 * L.Kp + 6 * L.e; }
DEV int col_sb(const BaLayout& L, int i) { return L.Rc + 9 * i; }
DEV int tri(int i, int j) { return i * (i + 1) / 2 + j; }   // packed lower, j <= i

// one state copy: [pose Kp*7 | sb K*9 | ex 7 | td 1]
DEV const double* st_pose(const BaLayout& L, const double* x, int i) { return x + 7 * i; }
DEV const double* st_sb(const BaLayout& L, const double* x, int i) { return x + 7 * L.Kp + 9 * i; }
DEV const double* st_ex(const BaLayout& L, const double* x) { return x + 7 * L.Kp + 9 * L.K; }

DEV double* lin_buf(const Ctx& c, int which) { return c.sc + c.Lp->so_buf + (size_t)which * c.Lp->buf_stride; }

// vg_ba_problem::max_solver_time_s against the device's constant-rate wall clock (BA_WALL_HZ, vg_target.h).  The clock is read by ONE thread and the verdict handed to the workgroup through LDS:
// wavefronts read the clock at different instants and must not disagree about a branch that contains barriers.
DEV bool time_is_up(const double* ctl, double max_s, double* lds_slot, int tid) {
    if (max_s <= 0.0) return false;                // (uniform: a kernel argument of the window)
    __syncthreads();
    if (tid == 0) *lds_slot = ((double)wall_clock64() - ctl[C_T0]) >= max_s * BA_WALL_HZ ? 1.0 : 0.0;
    __syncthreads();
    return *lds_slot != 0.0;
}

// ---- solver state carried between launches -----------------------------------------------------------------------
struct Ctl {
    int it, nacc, ninv, term, status, reuse, cur, pending, done, scaled, phase;
    double radius, mu, mu_solved, cost, x_norm, alpha, gtn2, gnn2, gtgn, dnorm, model, step_norm, x_norm_c, init_cost, qcam;
    double gnn2c, gtgnc, step2c, xn2c;        // large-window path (ba_layout.h)
};
DEV void ctl_load(Ctl& s, const double* p) {
    s.it = uni((int)p[C_IT]); s.nacc = uni((int)p[C_NACC]); s.ninv = uni((int)p[C_NINV]); s.term = uni((int)p[C_TERM]); s.status = uni((int)p[C_STATUS]);
    s.reuse = uni((int)p[C_REUSE]); s.cur = uni((int)p[C_CUR]); s.pending = uni((int)p[C_PENDING]); s.done = uni((int)p[C_DONE]); s.scaled = uni((int)p[C_SCALED]);
    s.radius = uni(p[C_RADIUS]); s.mu = uni(p[C_MU]); s.mu_solved = uni(p[C_MUSOLVED]); s.cost = uni(p[C_COST]); s.x_norm = uni(p[C_XNORM]);
    s.alpha = uni(p[C_ALPHA]); s.gtn2 = uni(p[C_GTN2]); s.gnn2 = uni(p[C_GNN2]); s.gtgn = uni(p[C_GTGN]); s.dnorm = uni(p[C_DNORM]); s.model = uni(p[C_MODEL]);
    s.step_norm = uni(p[C_STEPNORM]); s.x_norm_c = uni(p[C_XNORMC]); s.init_cost = uni(p[C_INITCOST]); s.qcam = uni(p[C_QCAM]);
    s.phase = uni((int)p[C_PHASE]); s.gnn2c = uni(p[C_GNN2C]); s.gtgnc = uni(p[C_GTGNC]); s.step2c = uni(p[C_STEP2C]); s.xn2c = uni(p[C_XN2C]);
}
DEV void ctl_store(const Ctl& s, double* p) {
    p[C_IT] = s.it; p[C_NACC] = s.nacc; p[C_NINV] = s.ninv; p[C_TERM] = s.term; p[C_STATUS] = s.status;
    p[C_REUSE] = s.reuse; p[C_CUR] = s.cur; p[C_PENDING] = s.pending; p[C_DONE] = s.done; p[C_SCALED] = s.scaled;
    p[C_RADIUS] = s.radius; p[C_MU] = s.mu; p[C_MUSOLVED] = s.mu_solved; p[C_COST] = s.cost; p[C_XNORM] = s.x_norm;
    p[C_ALPHA] = s.alpha; p[C_GTN2] = s.gtn2; p[C_GNN2] = s.gnn2; p[C_GTGN] = s.gtgn; p[C_DNORM] = s.dnorm; p[C_MODEL] = s.model;
    p[C_STEPNORM] = s.step_norm; p[C_XNORMC] = s.x_norm_c; p[C_INITCOST] = s.init_cost; p[C_QCAM] = s.qcam;
    p[C_PHASE] = s.phase; p[C_GNN2C] = s.gnn2c; p[C_GTGNC] = s.gtgnc; p[C_STEP2C] = s.step2c; p[C_XN2C] = s.xn2c;
}

// every field wave-uniform again (values recomputed on the VALU since the load are vector registers to the compiler)
DEV void ctl_uniform(Ctl& s) {
    s.it = uni(s.it); s.nacc = uni(s.nacc); s.ninv = uni(s.ninv); s.term = uni(s.term); s.status = uni(s.status);
    s.reuse = uni(s.reuse); s.cur = uni(s.cur); s.pending = uni(s.pending); s.done = uni(s.done); s.scaled = uni(s.scaled);
    s.radius = uni(s.radius); s.mu = uni(s.mu); s.mu_solved = uni(s.mu_solved); s.cost = uni(s.cost); s.x_norm = uni(s.x_norm);
    s.alpha = uni(s.alpha); s.gtn2 = uni(s.gtn2); s.gnn2 = uni(s.gnn2); s.gtgn = uni(s.gtgn); s.dnorm = uni(s.dnorm); s.model = uni(s.model);
    s.step_norm = uni(s.step_norm); s.x_norm_c = uni(s.x_norm_c); s.init_cost = uni(s.init_cost); s.qcam = uni(s.qcam);
    s.phase = uni(s.phase);
}

// Judge the pending candidate (TrustRegionMinimizer: parameter tolerance, function tolerance, step quality; then
// DoglegStrategy::StepAccepted / StepRejected).  Uniform: every thread computes the same from the same HBM values;
// thread 0 writes the trace.  Returns true if the candidate became the current point.
template <typename P>
DEV double sum_partials(P part, int nbl) {
    double cs = 0.0;
    for (int b = 0; b < nbl; ++b) cs += part[b];
    return uni(cs);
}
DEV bool judge_candidate(Ctl& s, double cs, const BaLayout& L, double* out, int* iout, int tid) {
    const double cost_cand = 0.5 * cs;
    const int slot = s.it - 1;
    bool accepted = false;
    int flag = 1;
    if (s.step_norm <= 1e-8 * (s.x_norm + 1e-8)) s.term = VG_TERM_CONVERGENCE;
    else if (fabs(s.cost - cost_cand) <= 1e-6 * s.cost) s.term = VG_TERM_CONVERGENCE;
    else {
        const double rho = (s.cost - cost_cand) / s.model;
        if (rho > 1e-3) {
            accepted = true;
            flag = 3;
            ++s.nacc;
            s.cur ^= 1;
            s.cost = cost_cand;
            s.x_norm = s.x_norm_c;
            if (rho < 0.25) s.radius *= 0.5;
            if (rho > 0.75) s.radius = fmax(s.radius, 3.0 * s.dnorm);
            s.mu = fmax(1e-8, 2.0 * s.mu / 10.0);
            s.reuse = 0;
        } else {
            s.radius *= 0.5;
            s.reuse = 1;
        }
    }
    s.pending = 0;
    if (tid == 0) {
        out[L.oo_trace + 1 * VG_MAX_ITERS + slot] = cost_cand;
        iout[4 + slot] = flag;
    }
    return accepted;
}

// ================================================================================================
// Prologue: IMU sqrt_info = U^-1 where covariance = U U^T (U upper) — exactly
// LLT(covariance^-1).matrixL().transpose() of imu_factor.h:64 (the Cholesky factor of the inverse is unique) without
// forming the badly conditioned inverse; J0^T J0 of the prior; state copy 0; control block.
// ================================================================================================
// The same factor formed the way imu_factor.h:64 SPELLS it -- covariance.inverse() by partial-pivot LU solved against the identity,
// then the LLT of the inverse's lower triangle, transposed -- in the operation order of the stand-in Eigen that oracle/_ref is built
// on (oracle/ref_stubs/eigen3/Eigen/Dense: MatrixBase::inverse(), LLT::compute()) and without contraction: the weights that build
// uses, bit for bit.  Selected by vg_config::imu_info_mode = VG_IMU_INFO_REFERENCE / vg_ba_set_imu_info_mode (does
// the form of this factor decide the sequence-level trust-region flips?  measured: profiles/r06_flip_stats.json).  One wavefront
// per factor, lane = row (LU, LLT) or column (the solve); LDS per wavefront: LU / L [225] | inverse [225] | permutation [15].
NOINL void imu_sqrt_info_ref(const Ctx& c_in, int fbeg, int fend) {
#pragma clang fp contract(off)
    const Ctx c = c_in;
    const BaLayout L = *c.Lp;
    double* A = LDSB + c.wave * 512;
    double* V = A + 240;
    double* Pm = V + 240;
    const int nimu = fend < L.K - 1 ? fend : L.K - 1;      // this workgroup's factors: [fbeg, fend) in passes of BA_NW
    const int* valid = c.ia + L.io_imu_valid;
    const int i = c.lane;
    for (int base = fbeg; base < nimu; base += BA_NW) {
        const int f = base + c.wave;
        const bool act = f < nimu && valid[f];
        const bool row = act && i < 15;
        const double* cov = c.di + L.do_imu + f * BA_IMU_STRIDE + IM_COV;
        if (act)
            for (int k = c.lane; k < 225; k += 64) A[k] = cov[k];
        if (row) Pm[i] = (double)i;
        __syncthreads();
        for (int k = 0; k < 15; ++k) {                 // LU in place, row pivoting: the first row of the largest magnitude
            const double v = (row && i >= k) ? fabs(A[i * 15 + k]) : -1.0;
            const double m = wave_max_all(v);
            const unsigned long long eq = __ballot(row && i >= k && v == m);
            const int piv = eq ? (int)__builtin_ctzll(eq) : k;
            __syncthreads();
            if (row && piv != k) {
                const double t = A[k * 15 + i];
                A[k * 15 + i] = A[piv * 15 + i];
                A[piv * 15 + i] = t;
                if (i == 0) { const double q = Pm[k]; Pm[k] = Pm[piv]; Pm[piv] = q; }
            }
            __syncthreads();
            if (row && i > k) {
                const double fk = A[i * 15 + k] / A[k * 15 + k];
                A[i * 15 + k] = fk;
                for (int j = k + 1; j < 15; ++j) A[i * 15 + j] -= fk * A[k * 15 + j];
            }
            __syncthreads();
        }
        if (row) {                                      // column i of the inverse: L U x = P e_i
            double y[15], x[15];
#pragma unroll
            for (int r = 0; r < 15; ++r) {
                double sacc = (Pm[r] == (double)i) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 15; ++k) if (k < r) sacc -= A[r * 15 + k] * y[k];
                y[r] = sacc;
            }
#pragma unroll
            for (int r = 14; r >= 0; --r) {
                double sacc = y[r];
#pragma unroll
                for (int k = 0; k < 15; ++k) if (k > r) sacc -= A[r * 15 + k] * x[k];
                x[r] = sacc / A[r * 15 + r];
            }
#pragma unroll
            for (int r = 0; r < 15; ++r) V[r * 15 + i] = x[r];
        }
        __syncthreads();
        for (int j = 0; j < 15; ++j) {                  // LLT of the lower triangle of V, L built in A
            if (row && i == j) {
                double d = V[j * 15 + j];
                for (int k = 0; k < j; ++k) d -= A[j * 15 + k] * A[j * 15 + k];
                A[j * 15 + j] = sqrt(d);
            }
            __syncthreads();
            if (row && i > j) {
                double sacc = V[i * 15 + j];
                for (int k = 0; k < j; ++k) sacc -= A[i * 15 + k] * A[j * 15 + k];
                A[i * 15 + j] = sacc / A[j * 15 + j];
            }
            __syncthreads();
        }
        double* Uo = c.sc + L.so_imuU + f * 225;        // sqrt_info = L^T (upper), zeros below the diagonal
        if (row)
            for (int r = 0; r < 15; ++r) Uo[r * 15 + i] = (i >= r) ? A[i * 15 + r] : 0.0;
        __syncthreads();
    }
}

NOINL void imu_sqrt_info(const Ctx& c_in, int fbeg, int fend) {
    const Ctx c = c_in;                       // local copies: fields read through the references would be re-loaded (flat_load +
    const BaLayout L = *c.Lp;                 // full s_waitcnt) after every store that might alias them
    // Round 6: a factor is one wavefront's business from the load to the store -- lane i holds row i of the 15 x 15 matrix in
    // registers, a pivot column reaches the other rows through v_readlane, and nothing in between waits for the workgroup (it used
    // to be three workgroup barriers and an LDS round trip per column, 45 per pass).  Same operations on the same operands in the same
    // order: the factor is the one the LDS form produced.
    lds_d* A = AS_LDS(LDSB + c.wave * 256);          // 15x15 scratch per wavefront (the factor, for the column solves below)
    const int nimu = fend < L.K - 1 ? fend : L.K - 1;      // this workgroup's factors: [fbeg, fend) in passes of BA_NW
    const glb_i* valid = AS_GLB_CI(c.ia + L.io_imu_valid);
    const int li = c.lane < 15 ? c.lane : 14;
    for (int base = fbeg; base < nimu; base += BA_NW) {
        const int f = uni(base + c.wave);
        if (!(f < nimu && valid[f])) continue;           // (uniform per wavefront; no workgroup barrier below)
        const glb_d* cov = AS_GLB_C(c.di + L.do_imu + f * BA_IMU_STRIDE + IM_COV);
        double a[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) a[k] = cov[li * 15 + k];
        // UL factorisation, columns from the last to the first: A = U U^T
#pragma unroll
        for (int j = 14; j >= 0; --j) {
            const double d = sqrt(readlane_f64(a[j], j));
            a[j] = c.lane < j ? a[j] / d : (c.lane == j ? d : a[j]);
            // row i (< j), entries kk = i .. j-1:  A[i][kk] -= A[i][j] A[kk][j]
#pragma unroll
            for (int kk = 0; kk < j; ++kk) {
                const double ukj = readlane_f64(a[j], kk);
                a[kk] = (c.lane <= kk) ? a[kk] - a[j] * ukj : a[kk];
            }
        }
        if (c.lane < 15) {
#pragma unroll
            for (int k = 0; k < 15; ++k) A[c.lane * 15 + k] = a[k];
        }
        __builtin_amdgcn_wave_barrier();
        // X = U^-1 (upper): lane = column j, back-substitute upward
        glb_d* Uo = AS_GLB(c.sc + L.so_imuU + f * 225);
        if (c.lane < 15) {
            const int j = c.lane;
            double x[15];
#pragma unroll
            for (int i = 14; i >= 0; --i) {
                double s = (i == j) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 14; k > i; --k) s -= (k <= j) ? A[i * 15 + k] * x[k] : 0.0;
                x[i] = (i <= j) ? s / A[i * 15 + i] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 15; ++i) Uo[i * 15 + j] = x[i];
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
}

#ifndef BA_SOLVE_W8_TU
extern "C" __global__ __launch_bounds__(BA_NT) void ba_prologue_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) {
    const BaLayout& L = *Lp;
    Ctx c;
    ctx_init(c, Lp, P, blockIdx.x);
    double* x = c.sc + L.so_x;
    DP_DECL;
    // A full batch runs one workgroup per window through everything below (gridDim.y == 1).  A few windows on an empty chip (the
    // latency layout, BaLayout::pro_split: the same windows whose IMU / prior linearisation is spread over workgroups) put the
    // independent pieces side by side, grid (nwin, 1 + ceil((K-1) / BA_NW) + 2): part 0 the clears, the state copy and the control
    // block; parts 1 .. nI one pass of BA_NW sqrt_info factors each; part nI+1 the transposed copy of J0; part nI+2 J0^T J0.  The
    // pieces write disjoint arrays and read only the caller's inputs: the launch lasts as long as its longest piece (one window:
    // 40 -> ~12 us).  Same operations per piece: identical results.
    const bool all = gridDim.y == 1;
    const int part = blockIdx.y, nI = (L.K - 1 + BA_NW - 1) / BA_NW;
    if (all || part == 0)
    {   // outputs of this window start from zero in every run (status words, iteration trace, "new prior valid" flag)
        double* out = P.out + (size_t)blockIdx.x * L.ostride;
        int* iout = P.iout + (size_t)blockIdx.x * L.oi_stride;
        int* miout = P.miout + (size_t)blockIdx.x * L.mi_stride;
        for (int k = c.tid; k < L.ostride; k += BA_NT) out[k] = 0.0;
        for (int k = c.tid; k < L.oi_stride; k += BA_NT) iout[k] = 0;
        for (int k = c.tid; k < L.mi_stride; k += BA_NT) miout[k] = 0;
    }
    if (all || part == 0) {
        for (int k = c.tid; k < 7 * L.Kp; k += BA_NT) x[k] = c.di[L.do_pose + k];
        for (int k = c.tid; k < 9 * L.K; k += BA_NT) x[7 * L.Kp + k] = c.di[L.do_sb + k];
        if (c.tid < 7) x[7 * L.Kp + 9 * L.K + c.tid] = c.di[L.do_ex + c.tid];
        if (c.tid == 7) x[7 * L.Kp + 9 * L.K + 7] = c.di[L.do_td];
        for (int k = c.tid; k < c.nL; k += BA_NT) c.sc[L.so_lam + k] = c.di[L.do_lam + k];
        if (c.tid < C_NCTL) {
            double v = 0.0;
            if (c.tid == C_RADIUS) v = 1e4;
            if (c.tid == C_MU || c.tid == C_MUSOLVED) v = 1e-8;
            if (c.tid == C_T0) v = (double)wall_clock64();
            c.sc[L.so_ctl + c.tid] = v;
        }
    }
    DP_ADD(40);
    if (all || (part >= 1 && part <= nI)) {
        const int fbeg = all ? 0 : (part - 1) * BA_NW, fend = all ? L.K - 1 : part * BA_NW;
        if (L.imu_info) imu_sqrt_info_ref(c, fbeg, fend);
        else imu_sqrt_info(c, fbeg, fend);
    }
    DP_ADD(41);
    if (c.nprior && (all || part > nI)) {
        const bool do_copy = all || part == nI + 1, do_prod = all || part == nI + 2;
        // J0^T J0 once per solve, J0 staged in LDS
        const int n = c.nprior;
        const double* J0 = c.pri + L.po_J0;
        double* Hp = c.sc + L.so_Hp;
        // the transposed copy the linearisation reads (its dot products run down the columns of J0)
        if (do_copy) {
            double* J0t = c.sc + L.so_J0t;
            for (int wk = c.tid; wk < n * n; wk += BA_NT) { const int r = wk / n, cc = wk % n; J0t[cc * L.Ncap + r] = J0[r * L.pld + cc]; }
        }
        // J0 staged in LDS when it fits (host: lds_pro); the two loops are spelled out so that the staged one keeps its
        // LDS addressing (a pointer that may be either costs flat loads in the inner loop)
        __syncthreads();
        DP_ADD(42);
        if (!do_prod) {
            // (this workgroup's piece was the transposed copy)
        } else if ((size_t)n * n * 8 <= (size_t)L.lds_pro) {
            double* J0s = LDSB;                  // n x n, row stride n
            for (int wk = c.tid; wk < n * n; wk += BA_NT) J0s[wk] = J0[(wk / n) * L.pld + wk % n];
            __syncthreads();
            DP_ADD(43);
            // J0^T J0 as X^T X on the matrix cores (round 4; it was n (n + 1) / 2 dot products of length n from LDS, two reads per
            // multiply-add): a wavefront owns lower 16 x 16 tiles (ta >= tb), A[i][k] = J0[4 s + k][16 ta + i] and B[k][j] =
            // J0[4 s + k][16 tb + j] come from the same staged rows
            const int T = (n + 15) >> 4, ntile = T * (T + 1) / 2, nks = (n + 3) >> 2;
            const int wave = c.tid >> 6, lane = c.tid & 63, li = lane & 15, lk = lane >> 4;
            for (int t = wave; t < ntile; t += BA_NT / 64) {
                int ta, tb;
                tri_decode(t, ta, tb);
                const int ca = 16 * ta + li, cb = 16 * tb + li;
                double4_t acc = (double4_t){0, 0, 0, 0};
                for (int ks = 0; ks < nks; ++ks) {
                    const int r = 4 * ks + lk;
                    const double av = (r < n && ca < n) ? J0s[r * n + ca] : 0.0;
                    const double bv = (r < n && cb < n) ? J0s[r * n + cb] : 0.0;
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int a = 16 * ta + lk + 4 * q, bb = 16 * tb + li;        // D[row = lk + 4 q][col = li]
                    if (a < n && bb <= a) {
                        Hp[a * L.Ncap + bb] = acc[q];
                    }
                }
            }
        } else {
            const int ld = L.pld;
            for (int wk = c.tid; wk < n * (n + 1) / 2; wk += BA_NT) {
                int a, bb;
                tri_decode(wk, a, bb);
                double s0 = 0.0, s1 = 0.0;
                int r = 0;
                for (; r + 1 < n; r += 2) { s0 += J0[r * ld + a] * J0[r * ld + bb]; s1 += J0[(r + 1) * ld + a] * J0[(r + 1) * ld + bb]; }
                if (r < n) s0 += J0[r * ld + a] * J0[r * ld + bb];
                Hp[a * L.Ncap + bb] = s0 + s1;
            }
        }
    }
    DP_ADD(44);
}
#endif

// ================================================================================================
// Linearisation kernel
// ================================================================================================
// All IMU factors at state x, one (half-)wavefront per factor: lanes 0..29 = Jacobian columns, lane 30 = the residual
// "column"; every lane evaluates the (cheap) factor context, weights its column with the upper-triangular U = sqrt_info
// read from an LDS copy and parks it in LDS; the workgroup then forms every factor's 30x30 Hessian block and J^T r.
//   imuJ [f][512]: 465 lower Hessian entries + 30 gradient entries.   JAC = false: residual only.
// Returns this thread's share of sum r^2.
template <bool JAC>
NOINL double imu_pass(const Ctx& c_in, const double* x_, double* imuJ_, int fbeg, int fend) {
    const Ctx c = c_in;                       // local copies: fields read through the references would be re-loaded (flat_load +
    const BaLayout L = *c.Lp;                 // full s_waitcnt) after every store that might alias them
    const int nimu = fend;                    // this workgroup's factors: [fbeg, fend)
    const glb_i* valid = AS_GLB_CI(c.ia + L.io_imu_valid);
    const glb_d* x = AS_GLB_C(x_);
    glb_d* imuJ = AS_GLB(imuJ_);
    const glb_d* gU = AS_GLB_C(c.sc + L.so_imuU);
    const int nb = fend - fbeg < BA_IMU_BATCH ? fend - fbeg : BA_IMU_BATCH;      // factors per pass (one pass per workgroup normally)
    double* Us = LDSB;                                      // [nb][225]
    double* panels = Us + ((nb * 225 + 1) & ~1);            // [nb][15][32]
    double cost = 0.0;
    const int per = nb > BA_NW ? 2 : 1;
    const int half = c.lane >> 5, hl = c.lane & 31;
    for (int f0 = fbeg; f0 < nimu; f0 += nb) {
        const int nf = nimu - f0 < nb ? nimu - f0 : nb;
        for (int k0 = c.tid; k0 < nf * 225; k0 += 8 * BA_NT) {      // (the loads of a thread issued together, then the LDS stores)
            double uv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int k = k0 + u * BA_NT; uv[u] = k < nf * 225 ? gU[f0 * 225 + k] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int k = k0 + u * BA_NT; if (k < nf * 225) Us[k] = uv[u]; }
        }
        __syncthreads();
        const int fl = c.wave * per + half;                 // factor of this (half-)wavefront inside the pass
        const int f = f0 + fl;
        const bool act = fl < nf && half < per && valid[f];
        if (act) {
            const glb_d* pre = AS_GLB_C(c.di + L.do_imu + f * BA_IMU_STRIDE);
            double pi[7], si[9], pj[7], sj[9];          // the four state blocks of the factor, through typed loads
#pragma unroll
            for (int k = 0; k < 7; ++k) { pi[k] = x[7 * f + k]; pj[k] = x[7 * (f + 1) + k]; }
#pragma unroll
            for (int k = 0; k < 9; ++k) { si[k] = x[7 * L.Kp + 9 * f + k]; sj[k] = x[7 * L.Kp + 9 * (f + 1) + k]; }
            const double* U = Us + fl * 225;
            double* panel = panels + fl * 480;
            ImuCtx ic;
            imu_ctx<JAC>(pre, pi, si, pj, sj, c.gnorm, ic);
            double raw[15];
            if (JAC && hl < 30) imu_raw_col(ic, pre, hl, raw);
            else {
#pragma unroll
                for (int q = 0; q < 15; ++q) raw[q] = ic.r[q];
            }
            if (hl <= 30) {
#pragma unroll
                for (int r = 0; r < 15; ++r) {
                    double s = 0.0;
#pragma unroll
                    for (int k = 0; k < 15; ++k) if (k >= r) s += U[r * 15 + k] * raw[k];
                    if (JAC) panel[r * 32 + hl] = s;
                    if (hl == 30) cost += s * s;
                }
            }
        }
        __syncthreads();
        if (JAC) {
            // H = X^T X of the weighted panel X (15 rows x [30 Jacobian columns | residual]) on v_mfma_f64_16x16x4: per factor the three
            // lower 16x16 tiles of the 31x31 product, four k-steps each (row 15 does not exist: zero); a wavefront takes the
            // (factor, tile) tasks t = wave, wave + 8, ...  Row 30 of the product is J^T r (the gradient entries 465 ..).
            // (Until round 4: 495 dot products of 15 terms per factor from LDS, 30 reads each -- a third of the pass.)
            const int k4 = c.lane >> 4, col = c.lane & 15;
            for (int t = c.wave; t < 3 * nf; t += BA_NW) {
                const int ff = t / 3, tile = t - 3 * ff;
                if (!valid[f0 + ff]) continue;                 // (uniform)
                const int tm = tile > 0 ? 1 : 0, tn = tile > 1 ? 1 : 0;
                const lds_d* panel = AS_LDS_C(panels + ff * 480);
                double av[4], bv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 4 * q + k4;
                    av[q] = r < 15 ? panel[r * 32 + 16 * tm + col] : 0.0;
                    bv[q] = r < 15 ? panel[r * 32 + 16 * tn + col] : 0.0;
                }
                double4_t acc = (double4_t){0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q], bv[q], acc, 0, 0, 0);
                glb_d* dst = imuJ + (f0 + ff) * 512;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int a = 16 * tm + k4 + 4 * reg, b = 16 * tn + col;      // D[row = (lane >> 4) + 4 reg][col = lane & 15]
                    if (a < 30 && b <= a) dst[a * (a + 1) / 2 + b] = acc[reg];
                    else if (a == 30 && b < 30) dst[465 + b] = acc[reg];
                }
            }
        }
        __syncthreads();
    }
    return cost;
}

// Prior (MarginalizationFactor::Evaluate, marginalization_factor.cpp:333-381): dx per kept block, r = r0 + J0 dx.
DEV const double* state_block(const BaLayout& L, const double* x, int kind, int idx) {
    if (kind == VG_BLK_POSE) return st_pose(L, x, idx);
    if (kind == VG_BLK_SPEEDBIAS) return st_sb(L, x, idx);
    if (kind == VG_BLK_EXPOSE) return st_ex(L, x);
    return st_ex(L, x) + 7;     // td
}
// LDS use: dx [Ncap] + part [4 Ncap] at `lds`.  Writes r to pr and (gpr != nullptr) J0^T r to gpr (global).  Returns this
// thread's share of sum r^2.
NOINL double prior_pass(const Ctx& c_in, const double* x, double* pr_, double* gpr_, double* lds_) {
    const Ctx c = c_in;                       // local copies: fields read through the references would be re-loaded (flat_load +
    const BaLayout L = *c.Lp;                 // full s_waitcnt) after every store that might alias them
    if (c.nprior == 0) return 0.0;
    lds_d* dx = AS_LDS(lds_);
    lds_d* part = dx + L.Ncap;
    glb_d* pr = AS_GLB(pr_);
    glb_d* gpr = AS_GLB(gpr_);
    const glb_i* kind = AS_GLB_CI(c.ia + L.io_pb_kind);
    const glb_i* idx = AS_GLB_CI(c.ia + L.io_pb_idx);
    const glb_i* off = AS_GLB_CI(c.ia + L.io_pb_off);
    const glb_i* x0off = AS_GLB_CI(c.ia + L.io_pb_x0off);
    __syncthreads();
    for (int b = c.tid; b < c.nblk; b += BA_NT) {
        const glb_d* xb = AS_GLB_C(state_block(L, x, kind[b], idx[b]));
        const glb_d* x0 = AS_GLB_C(c.pri + L.po_x0 + x0off[b]);
        lds_d* d = dx + off[b];
        if (kind[b] == VG_BLK_SPEEDBIAS) {
#pragma unroll
            for (int k = 0; k < 9; ++k) d[k] = xb[k] - x0[k];
        } else if (kind[b] == VG_BLK_TD) {
            d[0] = xb[0] - x0[0];
        } else {
            d[0] = xb[0] - x0[0]; d[1] = xb[1] - x0[1]; d[2] = xb[2] - x0[2];
            const double q0[4] = {x0[3], x0[4], x0[5], x0[6]}, qb[4] = {xb[3], xb[4], xb[5], xb[6]};
            double qi[4], dq[4];
            q_inv(q0, qi);
            q_mul(qi, qb, dq);
            const double sgn = (dq[3] >= 0) ? 2.0 : -2.0;
            d[3] = sgn * dq[0]; d[4] = sgn * dq[1]; d[5] = sgn * dq[2];
        }
    }
    __syncthreads();
    const int n = c.nprior;
    const glb_d* J0t = AS_GLB_C(c.sc + L.so_J0t);     // J0t[c*Ncap + r] = J0[r][c]  (coalesced over r)
    const glb_d* r0 = AS_GLB_C(c.pri + L.po_r0);
    double cost = 0.0;
    // r = r0 + J0 dx, the n-term dot product of every row split over 4 threads (host guarantees 4 Ncap <= BA_NT)
    // (round 6: the loads of a partial sum are requested TOGETHER, sixteen at a time, from clamped addresses -- the loop over k waited
    //  for every J0 entry before it asked for the next: 19 dependent HBM round trips per product, most of the cost-only launch and
    //  of a spread round of a single window.  Terms beyond n enter as exact zeros: the same sums.)
    for (int w = c.tid; w < 4 * L.Ncap; w += BA_NT) {
        const int r = w % L.Ncap, q = w / L.Ncap;
        const int rc = r < n ? r : 0;
        double s = 0.0;
        for (int k0 = q; k0 < n; k0 += 64) {
            double jv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int k = k0 + 4 * u; jv[u] = J0t[(k < n ? k : q) * L.Ncap + rc]; }
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int k = k0 + 4 * u; s += ((r < n && k < n) ? jv[u] : 0.0) * dx[k < n ? k : q]; }
        }
        part[w] = s;
    }
    __syncthreads();
    __syncthreads();
    lds_d* rl = dx;                              // the residual replaces dx in LDS
    for (int r = c.tid; r < n; r += BA_NT) {
        const double s = r0[r] + ((part[r] + part[L.Ncap + r]) + (part[2 * L.Ncap + r] + part[3 * L.Ncap + r]));
        pr[r] = s;
        rl[r] = s;
        cost += s * s;
    }
    __syncthreads();
    if (gpr_) {
        // gradient of the prior J0^T r (the solve kernel adds it to g through the prior column map), same 4-way split
        const glb_d* J0 = AS_GLB_C(c.pri + L.po_J0);   // row-major: J0[r*pld + c] coalesced over c
        for (int w = c.tid; w < 4 * L.Ncap; w += BA_NT) {
            const int a = w % L.Ncap, q = w / L.Ncap;
            const int ac = a < n ? a : 0;
            double s = 0.0;
            for (int k0 = q; k0 < n; k0 += 64) {
                double jv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int k = k0 + 4 * u; jv[u] = J0[(k < n ? k : q) * L.pld + ac]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int k = k0 + 4 * u; s += ((a < n && k < n) ? jv[u] : 0.0) * rl[k < n ? k : q]; }
            }
            part[w] = s;
        }
        __syncthreads();
        for (int a = c.tid; a < n; a += BA_NT)
            gpr[a] = (part[a] + part[L.Ncap + a]) + (part[2 * L.Ncap + a] + part[3 * L.Ncap + a]);
    }
    return cost;
}

// prior_pass for the fused factor kernel: J0 (n x n, 46 KB for n = 76) is read from HBM ONCE, into LDS, and both products -- r = r0 +
// J0 dx and J0^T r -- run from there (prior_pass reads J0^T for the first and J0 for the second: two dependent passes over HBM).
// All loads of J0 are issued before dx is formed, so their latency runs under that phase.  Same four-way split and the same order of
// every sum as prior_pass: bit-identical results.  LDS at `lds_`: J0 [n][n + 1] | dx [Ncap] | part [4 Ncap]; needs n^2 <= 12 BA_NT.
DEV double prior_staged(const Ctx& c, const BaLayout& L, const double* x, double* pr_, double* gpr_, double* lds_) {
    const int n = c.nprior, ld = n + 1;
    lds_d* Jl = AS_LDS(lds_);
    lds_d* dx = Jl + ((n * ld + 1) & ~1);
    lds_d* part = dx + L.Ncap;
    glb_d* pr = AS_GLB(pr_);
    glb_d* gpr = AS_GLB(gpr_);
    const glb_i* kind = AS_GLB_CI(c.ia + L.io_pb_kind);
    const glb_i* idx = AS_GLB_CI(c.ia + L.io_pb_idx);
    const glb_i* off = AS_GLB_CI(c.ia + L.io_pb_off);
    const glb_i* x0off = AS_GLB_CI(c.ia + L.io_pb_x0off);
    const glb_d* J0 = AS_GLB_C(c.pri + L.po_J0);
    double pj[12];
#pragma unroll
    for (int u = 0; u < 12; ++u) {
        const int e = c.tid + u * BA_NT;
        const int r = e / n, cc = e - r * n;
        pj[u] = e < n * n ? J0[r * L.pld + cc] : 0.0;
    }
    for (int b = c.tid; b < c.nblk; b += BA_NT) {
        const glb_d* xb = AS_GLB_C(state_block(L, x, kind[b], idx[b]));
        const glb_d* x0 = AS_GLB_C(c.pri + L.po_x0 + x0off[b]);
        lds_d* d = dx + off[b];
        if (kind[b] == VG_BLK_SPEEDBIAS) {
#pragma unroll
            for (int k = 0; k < 9; ++k) d[k] = xb[k] - x0[k];
        } else if (kind[b] == VG_BLK_TD) {
            d[0] = xb[0] - x0[0];
        } else {
            d[0] = xb[0] - x0[0]; d[1] = xb[1] - x0[1]; d[2] = xb[2] - x0[2];
            const double q0[4] = {x0[3], x0[4], x0[5], x0[6]}, qb[4] = {xb[3], xb[4], xb[5], xb[6]};
            double qi[4], dq[4];
            q_inv(q0, qi);
            q_mul(qi, qb, dq);
            const double sgn = (dq[3] >= 0) ? 2.0 : -2.0;
            d[3] = sgn * dq[0]; d[4] = sgn * dq[1]; d[5] = sgn * dq[2];
        }
    }
#pragma unroll
    for (int u = 0; u < 12; ++u) {
        const int e = c.tid + u * BA_NT;
        const int r = e / n, cc = e - r * n;
        if (e < n * n) Jl[r * ld + cc] = pj[u];
    }
    __syncthreads();
    const glb_d* r0 = AS_GLB_C(c.pri + L.po_r0);
    double cost = 0.0;
    for (int w = c.tid; w < 4 * L.Ncap; w += BA_NT) {
        const int r = w % L.Ncap, q = w / L.Ncap;
        double sacc = 0.0;
        if (r < n)
            for (int k = q; k < n; k += 4) sacc += Jl[r * ld + k] * dx[k];
        part[w] = sacc;
    }
    __syncthreads();
    lds_d* rl = dx;                              // the residual replaces dx
    double rv = 0.0;
    const bool own = c.tid < n;                  // (n <= Ncap <= BA_NT / 4)
    if (own) rv = r0[c.tid] + ((part[c.tid] + part[L.Ncap + c.tid]) + (part[2 * L.Ncap + c.tid] + part[3 * L.Ncap + c.tid]));
    __syncthreads();
    if (own) { pr[c.tid] = rv; rl[c.tid] = rv; cost += rv * rv; }
    __syncthreads();
    for (int w = c.tid; w < 4 * L.Ncap; w += BA_NT) {
        const int a = w % L.Ncap, q = w / L.Ncap;
        double sacc = 0.0;
        if (a < n)
            for (int k = q; k < n; k += 4) sacc += Jl[k * ld + a] * rl[k];
        part[w] = sacc;
    }
    __syncthreads();
    for (int a = c.tid; a < n; a += BA_NT)
        gpr[a] = (part[a] + part[L.Ncap + a]) + (part[2 * L.Ncap + a] + part[3 * L.Ncap + a]);
    __syncthreads();
    return cost;
}

// inputs of one projection factor, copied into registers through global-memory typed loads (the state / observation
// pointers are generic: reading the factor's 38 doubles through them would be 38 flat_loads)
struct ProjIn {
    double pi[7], pj[7], oi[8], oj[8], ex[8];
    double lam;
    int i, j, l;
};
DEV void proj_fetch(const Ctx& c, int f, const double* x_, const double* lam_, ProjIn& p) {
    const BaLayout& L = *c.Lp;
    const glb_i* ia = AS_GLB_CI(c.ia);
    const glb_d* x = AS_GLB_C(x_);
    const glb_d* obs = AS_GLB_C(c.di + L.do_obs);
    p.i = ia[L.io_fac_i + f];
    p.j = ia[L.io_fac_j + f];
    p.l = ia[L.io_fac_lm + f];
    const int oi = ia[L.io_fac_oi + f], oj = ia[L.io_fac_oj + f];
#pragma unroll
    for (int k = 0; k < 7; ++k) { p.pi[k] = x[7 * p.i + k]; p.pj[k] = x[7 * p.j + k]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        p.oi[k] = obs[oi * BA_OBS_STRIDE + k];
        p.oj[k] = obs[oj * BA_OBS_STRIDE + k];
        p.ex[k] = x[7 * L.Kp + 9 * L.K + k];
    }
    p.lam = AS_GLB_C(lam_)[p.l];
}
// the same with the state read from an LDS copy (fused factor kernel, round 6: the poses and the extrinsic pose were 22 of a factor's 39
// global loads, and a CU's L1 moves 64 B per cycle; the observations and the inverse depth still come from HBM)
DEV void proj_fetch_staged(const Ctx& c, int f, const lds_d* xs, const double* lam_, ProjIn& p) {
    const BaLayout& L = *c.Lp;
    const glb_i* ia = AS_GLB_CI(c.ia);
    const glb_d* obs = AS_GLB_C(c.di + L.do_obs);
    p.i = ia[L.io_fac_i + f];
    p.j = ia[L.io_fac_j + f];
    p.l = ia[L.io_fac_lm + f];
    const int oi = ia[L.io_fac_oi + f], oj = ia[L.io_fac_oj + f];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        p.oi[k] = obs[oi * BA_OBS_STRIDE + k];
        p.oj[k] = obs[oj * BA_OBS_STRIDE + k];
    }
    p.lam = AS_GLB_C(lam_)[p.l];
#pragma unroll
    for (int k = 0; k < 7; ++k) { p.pi[k] = xs[7 * p.i + k]; p.pj[k] = xs[7 * p.j + k]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) p.ex[k] = xs[7 * L.Kp + 9 * L.K + k];
}
DEV void proj_jac(const Ctx& c, const ProjIn& p, double* r, double* Ji, double* Jj, double* Jex,
                  double* Jl, double* Jtd) {
    const BaLayout& L = *c.Lp;
    if (L.t) {
        if (L.e) proj_eval<true, true, true>(p.pi, p.pj, p.ex, p.lam, p.oi, p.oj, p.ex[7], c.focal, c.tr, c.row, r, Ji, Jj, Jex, Jl, Jtd);
        else proj_eval<true, true, false>(p.pi, p.pj, p.ex, p.lam, p.oi, p.oj, p.ex[7], c.focal, c.tr, c.row, r, Ji, Jj, Jex, Jl, Jtd);
    } else {
        if (L.e) proj_eval<false, true, true>(p.pi, p.pj, p.ex, p.lam, p.oi, p.oj, 0.0, c.focal, c.tr, c.row, r, Ji, Jj, Jex, Jl, Jtd);
        else proj_eval<false, true, false>(p.pi, p.pj, p.ex, p.lam, p.oi, p.oj, 0.0, c.focal, c.tr, c.row, r, Ji, Jj, Jex, Jl, Jtd);
    }
}

// One projection factor -> loss-corrected record (CauchyLoss(1.0): rho = log(1+s), residual and Jacobian * sqrt(rho'),
// rho'' < 0 branch of corrector.cc).  Record (REC doubles): [0..11] Ji as (row0,row1) pairs per column | [12..23] Jj |
// [24,25] Jl | [26,27] r | [28..39] Jex | [40,41] Jtd.   Returns rho(s).
NOINL double proj_linearize(const Ctx& c_in, int f, const double* x, const double* lam, double* recs) {
    const Ctx c = c_in;                       // local copies: fields read through the references would be re-loaded (flat_load +
    const BaLayout L = *c.Lp;                 // full s_waitcnt) after every store that might alias them
    ProjIn p;
    proj_fetch(c, f, x, lam, p);
    double r[2], Ji[12], Jj[12], Jex[12], Jl[2], Jtd[2];
    proj_jac(c, p, r, Ji, Jj, Jex, Jl, Jtd);
    const double s = r[0] * r[0] + r[1] * r[1];
    const double sq = sqrt(1.0 / (1.0 + s));
    glb_d* rec = AS_GLB(recs) + (size_t)AS_GLB_CI(c.ia)[L.io_fac_slot + f] * L.REC;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        rec[2 * k] = sq * Ji[k]; rec[2 * k + 1] = sq * Ji[6 + k];
        rec[12 + 2 * k] = sq * Jj[k]; rec[12 + 2 * k + 1] = sq * Jj[6 + k];
    }
    rec[24] = sq * Jl[0]; rec[25] = sq * Jl[1];
    rec[26] = sq * r[0]; rec[27] = sq * r[1];
    if (L.e) {
#pragma unroll
        for (int k = 0; k < 6; ++k) { rec[28 + 2 * k] = sq * Jex[k]; rec[28 + 2 * k + 1] = sq * Jex[6 + k]; }
    }
    if (L.t) { rec[28 + 12 * L.e] = sq * Jtd[0]; rec[29 + 12 * L.e] = sq * Jtd[1]; }
    return log1p(s);
}
NOINL double proj_cost(const Ctx& c_in, int f, const double* x, const double* lam) {
    const Ctx c = c_in;                       // local copies: fields read through the references would be re-loaded (flat_load +
    const BaLayout L = *c.Lp;                 // full s_waitcnt) after every store that might alias them
    ProjIn p;
    proj_fetch(c, f, x, lam, p);
    double r[2];
    if (L.t) proj_eval<true, false, false>(p.pi, p.pj, p.ex, p.lam, p.oi, p.oj, p.ex[7], c.focal, c.tr, c.row, r, 0, 0, 0, 0, 0);
    else proj_eval<false, false, false>(p.pi, p.pj, p.ex, p.lam, p.oi, p.oj, 0.0, c.focal, c.tr, c.row, r, 0, 0, 0, 0, 0);
    return log1p(r[0] * r[0] + r[1] * r[1]);
}

// Projection factors: grid (nbf, nwin), tiles of BA_LIN_NT factors, thread per factor, no LDS beyond the reduction.
// cost_only != 0: residuals only (the last candidate of a solve).
#ifndef BA_PROJ_WAVES
#define BA_PROJ_WAVES 1
#endif
#ifndef BA_SOLVE_W8_TU
extern "C" __global__ __launch_bounds__(BA_LIN_NT, BA_PROJ_WAVES) void ba_linearize_proj_kernel(const BaLayout* __restrict__ Lp, BaPtrs P, int cost_only) {
    const BaLayout& L = *Lp;
    Ctx c;
    ctx_init(c, Lp, P, blockIdx.y);
    const double* ctl = c.sc + L.so_ctl;
    if (ctl[C_DONE] != 0.0) return;
    const int which = ((int)ctl[C_CUR]) ^ (ctl[C_PENDING] != 0.0 ? 1 : 0);      // the point to evaluate: candidate if one is pending
    const double* x = c.sc + L.so_x + which * L.nst;
    const double* lam = c.sc + L.so_lam + which * L.Lcap;
    const int b = blockIdx.x;
    __shared__ double red[2 * (BA_LIN_NT / 64)];
    double share = 0.0;
    const int f = b * BA_LIN_NT + c.tid;
    if (f < c.nF) share = cost_only ? proj_cost(c, f, x, lam) : proj_linearize(c, f, x, lam, c.sc + L.so_rec);
    const double tot = block_sum(red, BA_LIN_NT / 64, c.lane, c.wave, share);
    if (c.tid == 0) c.sc[L.so_part + b] = tot;
}
#endif

// IMU factors + prior: one workgroup per window (LDS: sqrt_info copies + weighted Jacobian panels).
#ifndef BA_SOLVE_W8_TU
extern "C" __global__ __launch_bounds__(BA_NT) void ba_linearize_imu_kernel(const BaLayout* __restrict__ Lp, BaPtrs P, int cost_only) {
    const BaLayout& L = *Lp;
    Ctx c;
    ctx_init(c, Lp, P, blockIdx.y);
    const double* ctl = c.sc + L.so_ctl;
    if (ctl[C_DONE] != 0.0) return;
    const int which = ((int)ctl[C_CUR]) ^ (ctl[C_PENDING] != 0.0 ? 1 : 0);
    const double* x = c.sc + L.so_x + which * L.nst;
    double* buf = lin_buf(c, which);
    __shared__ double red[2 * BA_NW];
    // grid (nig + nprw, nwin): workgroup g < nig linearises the IMU factors [g igs, (g + 1) igs); the prior is handled by
    // workgroup nig (nprw = 1) or by workgroup 0 (nprw = 0).  Few windows in the batch (latency matters, the chip is empty):
    // nig = (K-1)/2 groups + a prior workgroup — the factors are independent chains of dependent steps, spread over
    // workgroups the launch lasts as long as one group.  Full batches: one workgroup per window (nig = 1, nprw = 0) — six
    // half-empty workgroups per window cost more issue slots than the latency they save (measured: 40 -> 90 us per launch).
    const int g = blockIdx.x;
    double share = 0.0;
    if (g < L.nig) {
        const int fbeg = g * L.igs, fend = (fbeg + L.igs < L.K - 1) ? fbeg + L.igs : L.K - 1;
        if (cost_only) share = imu_pass<false>(c, x, buf + L.bo_imuJ, fbeg, fend);
        else share = imu_pass<true>(c, x, buf + L.bo_imuJ, fbeg, fend);
    }
    if (L.nprw ? g == L.nig : g == 0) share += prior_pass(c, x, buf + L.bo_pr, cost_only ? nullptr : buf + L.bo_gpr, LDSB);
    const double tot = block_sum(red, BA_NW, c.lane, c.wave, share);
    if (c.tid == 0) c.sc[L.so_part + L.nbf + g] = tot;
}
#endif

// ================================================================================================
// Accumulation kernel: J^T J / J^T r of the projection factors from their records.
// ================================================================================================
// sum over slots [b,e) of  rec[offA..+1] . rec[offB..+1]   (four independent accumulation chains: the eight 16-byte loads
// of four consecutive records are in flight together -- a task is a chain of L2 round trips, not of flops; fixed order
// -> deterministic)
DEV double seg_dot(const double* recs, int REC, int b, int e, int offA, int offB) {
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    const double* pa = recs + (size_t)b * REC + offA;
    const double* pb = recs + (size_t)b * REC + offB;
    int s = b;
    for (; s + 3 < e; s += 4, pa += 4 * REC, pb += 4 * REC) {
        const double2 a0 = *(const double2*)(pa), b0 = *(const double2*)(pb);
        const double2 a1 = *(const double2*)(pa + REC), b1 = *(const double2*)(pb + REC);
        const double2 a2 = *(const double2*)(pa + 2 * REC), b2 = *(const double2*)(pb + 2 * REC);
        const double2 a3 = *(const double2*)(pa + 3 * REC), b3 = *(const double2*)(pb + 3 * REC);
        acc0 += a0.x * b0.x + a0.y * b0.y;
        acc1 += a1.x * b1.x + a1.y * b1.y;
        acc2 += a2.x * b2.x + a2.y * b2.y;
        acc3 += a3.x * b3.x + a3.y * b3.y;
    }
    const int rem = e - s;                        // 0..3 records left: their loads are issued together too
    if (rem > 0) {
        const double2 z = {0.0, 0.0};
        const double2 a0 = *(const double2*)(pa), b0 = *(const double2*)(pb);
        const double2 a1 = rem > 1 ? *(const double2*)(pa + REC) : z, b1 = rem > 1 ? *(const double2*)(pb + REC) : z;
        const double2 a2 = rem > 2 ? *(const double2*)(pa + 2 * REC) : z, b2 = rem > 2 ? *(const double2*)(pb + 2 * REC) : z;
        acc0 += a0.x * b0.x + a0.y * b0.y;
        acc1 += a1.x * b1.x + a1.y * b1.y;
        acc2 += a2.x * b2.x + a2.y * b2.y;
    }
    return (acc0 + acc1) + (acc2 + acc3);
}

// Diagonal pose block a of the camera system: it visits every factor anchored at or targeting frame a (~10x the visits
// of an off-diagonal block), so one workgroup of 4 wavefronts owns it: 21 lower entries + 6 gradient entries on lanes
// 0-26 / 27-53 of every wavefront = 8 lane segments that each take an eighth of every (anchor, target) slot range; the
// eight partial sums are combined through LDS in a fixed order.
DEV void diag_pose_task(const Ctx& c, int a, const double* recs, double* Sp, double* gp, double* part) {
    const BaLayout& L = *c.Lp;
    const int Kp = L.Kp, REC = L.REC;
    const int* ptr = c.ia + L.io_pair_ptr;
    const int seg = c.lane >= 27 ? 1 : 0, e = c.lane - 27 * seg;
    const int sidx = 2 * c.wave + seg;            // 0..7
    const bool on = c.lane < 54;
    const bool isg = e >= 21;
    int p2 = 0, q2 = 0;
    if (!isg) tri_decode(e, p2, q2); else p2 = e - 21;
    double sum = 0.0;
    if (on) {
        const int oB_i = isg ? 26 : 2 * q2, oB_j = isg ? 26 : 12 + 2 * q2;
        {
            const int s0 = ptr[a * Kp], len = ptr[(a + 1) * Kp] - s0;
            sum += seg_dot(recs, REC, s0 + len * sidx / 8, s0 + len * (sidx + 1) / 8, 2 * p2, oB_i);
        }
        for (int a2 = 0; a2 < a; ++a2) {
            const int s0 = ptr[a2 * Kp + a], len = ptr[a2 * Kp + a + 1] - s0;
            sum += seg_dot(recs, REC, s0 + len * sidx / 8, s0 + len * (sidx + 1) / 8, 12 + 2 * p2, oB_j);
        }
        part[sidx * 27 + e] = sum;
    }
    __syncthreads();
    if (c.tid < 27) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) tot += part[q * 27 + c.tid];
        if (c.tid >= 21) gp[6 * a + c.tid - 21] = tot;
        else Sp[tri(6 * a + p2, 6 * a + q2)] = tot;
    }
}

// Owner task = one off-diagonal (<=6)x(<=6) block of the camera part of S: lane = entry (p,q).  The slot table is sorted
// by (anchor, target) pair, so an entry that involves pose j only visits the factors with that pair: fixed summation
// order, no atomics.
DEV void owner_task(const Ctx& c, int task, int lane, const double* recs, double* Sp, double* gp) {
    const BaLayout& L = *c.Lp;
    const int Kp = L.Kp, REC = L.REC;
    const int* ptr = c.ia + L.io_pair_ptr;
    const int offEx = 28, offTd = 28 + 12 * L.e;
    int br, bc;
    tri_decode(task, br, bc);
    const int kr = br < Kp ? 0 : (br == Kp && L.e ? 1 : 2);     // 0 pose, 1 ex, 2 td
    const int kc = bc < Kp ? 0 : (bc == Kp && L.e ? 1 : 2);
    const int dr = kr == 2 ? 1 : 6, dc = kc == 2 ? 1 : 6;
    const int rowbase = kr == 0 ? 6 * br : (kr == 1 ? col_ex(L) : col_td(L));
    const int colbase = kc == 0 ? 6 * bc : (kc == 1 ? col_ex(L) : col_td(L));
    // (neither diagonal blocks nor gradient entries reach this function: diag_pose_task / global_task own them)
    const int p = lane / 6, q = lane % 6;
    if (!(lane < 36 && p < dr && q < dc)) return;
    double acc = 0.0;
    if (kr == 0 && kc == 0) {
        // row block br = target j, column block bc = anchor i
        acc += seg_dot(recs, REC, ptr[bc * Kp + br], ptr[bc * Kp + br + 1], 12 + 2 * p, 2 * q);
    } else {
        const int oA = (kr == 1 ? offEx : offTd) + 2 * p;
        // kc == 0: the blocks among ex / td are global_task()'s
        const int a = bc;
        acc += seg_dot(recs, REC, ptr[a * Kp], ptr[(a + 1) * Kp], oA, 2 * q);
        for (int a2 = 0; a2 < a; ++a2)
            acc += seg_dot(recs, REC, ptr[a2 * Kp + a], ptr[a2 * Kp + a + 1], oA, 12 + 2 * q);
    }
    Sp[tri(rowbase + p, colbase + q)] = acc;
}

// The blocks among the extrinsic pose and td (<= 7 columns: <= 28 lower entries + <= 7 gradient entries) visit EVERY factor:
// one workgroup, entry e on thread e of each of 7 thread segments, every segment takes a seventh of the slot range, the
// partial sums are combined through LDS in a fixed order (same scheme as diag_pose_task).
DEV void global_task(const Ctx& c, const double* recs, double* Sp, double* gp, double* part) {
    const BaLayout& L = *c.Lp;
    const int REC = L.REC;
    const int ng = 6 * L.e + L.t, ntri = ng * (ng + 1) / 2, ne = ntri + ng;
    const int nslots = (c.ia + L.io_pair_ptr)[L.Kp * L.Kp];
    const int seg = c.tid / 36, e = c.tid - 36 * seg;
    const bool isg = e >= ntri;
    int p = 0, q = 0;
    if (!isg) tri_decode(e, p, q); else p = e - ntri;
    // global column k -> record offset of its 2-vector: ex columns 28 + 2k, td 28 + 12 e
    if (seg < 7 && e < ne) {
        const int oA = 28 + 2 * p, oB = isg ? 26 : 28 + 2 * q;
        part[seg * 36 + e] = seg_dot(recs, REC, nslots * seg / 7, nslots * (seg + 1) / 7, oA, oB);
    }
    __syncthreads();
    if (c.tid < ne) {
        double tot = 0.0;
#pragma unroll
        for (int s7 = 0; s7 < 7; ++s7) tot += part[s7 * 36 + c.tid];
        const int base = col_ex(L);               // ex columns first, td last: contiguous from 6 Kp
        if (isg) gp[base + p] = tot;
        else Sp[tri(base + p, base + q)] = tot;
    }
}

// per-landmark sums: h = sum Jl.Jl, b = sum Jl.r, W column -> Wt[col][l] (landmark index fastest: coalesced stores).
// ONE pass over the landmark's records, every Wt row stored exactly once (a zero fill followed by scattered 8-byte stores costs
// ~2.5x the write traffic once the zero-filled lines have left L2).  pack_window stores the factors of a landmark by strictly
// increasing target frame (start + 1, start + 2, ..., then the relocalisation pose K) and the anchor precedes every target, so
// the target columns can be written as the factors come by -- zeros for the frames in between -- and only the anchor's column
// (a sum over all factors, like h / b and the ex / td rows) waits for the end.  The task is a chain of L2 round trips (slot
// index -> record), not of flops: the slot and target indices of the next factor are requested one factor ahead.  (Until round
// 3 this was two passes, the second re-reading every record to walk the columns in order: same values, same order of every
// sum, twice the chain.)
DEV void landmark_task(const Ctx& c, int l, const double* recs, double* buf) {
    const BaLayout& L = *c.Lp;
    const int REC = L.REC;
    double* Wt = buf + L.bo_Wt + l;
    const size_t ldw = L.Lcap;
    if (l >= c.nL) {
        for (int row = 0; row < L.RcPad; ++row) Wt[row * ldw] = 0.0;
        return;
    }
    const int fb = c.ia[L.io_lm_fbeg + l], fe = c.ia[L.io_lm_fbeg + l + 1];
    double h = 0.0, b = 0.0, wi[6] = {0, 0, 0, 0, 0, 0};
    const int anchor = fb < fe ? c.ia[L.io_fac_i + fb] : -1;
    int jn = 0;                                   // next pose column to store
    int slot_n = 0, j_n = 0;
    if (fb < fe) { slot_n = c.ia[L.io_fac_slot + fb]; j_n = c.ia[L.io_fac_j + fb]; }
    for (int f = fb; f < fe; ++f) {
        const double* rec = recs + (size_t)slot_n * REC;
        const int j = j_n;
        if (f + 1 < fe) { slot_n = c.ia[L.io_fac_slot + f + 1]; j_n = c.ia[L.io_fac_j + f + 1]; }
        const double l0 = rec[24], l1 = rec[25];
        double v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) v[k] = rec[12 + 2 * k] * l0 + rec[12 + 2 * k + 1] * l1;
        h += l0 * l0 + l1 * l1;
        b += l0 * rec[26] + l1 * rec[27];
#pragma unroll
        for (int k = 0; k < 6; ++k) wi[k] += rec[2 * k] * l0 + rec[2 * k + 1] * l1;
        for (; jn < j; ++jn) {                    // frames without a factor of this landmark (the anchor's column comes last)
            if (jn == anchor) continue;
#pragma unroll
            for (int k = 0; k < 6; ++k) Wt[(size_t)(col_pose(L, jn) + k) * ldw] = 0.0;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) Wt[(size_t)(col_pose(L, j) + k) * ldw] = v[k];
        jn = j + 1;
    }
    for (; jn < L.Kp; ++jn) {
        if (jn == anchor) continue;
#pragma unroll
        for (int k = 0; k < 6; ++k) Wt[(size_t)(col_pose(L, jn) + k) * ldw] = 0.0;
    }
    if (anchor >= 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) Wt[(size_t)(col_pose(L, anchor) + k) * ldw] = wi[k];
    }
    buf[L.bo_h + l] = h;
    buf[L.bo_b + l] = b;
    int row = 6 * L.Kp;
    if (L.e | L.t) {
        // the rows of the extrinsic pose / td (only when they are estimated): a pass of their own over the records (now in L1 /
        // L2), so that the pass above -- the one every configuration runs -- keeps the kernel at 4 wavefronts per SIMD
        double wex[6] = {0, 0, 0, 0, 0, 0}, wtd = 0.0;
        const int otd = 28 + 12 * L.e;
        for (int f = fb; f < fe; ++f) {
            const double* rec = recs + (size_t)c.ia[L.io_fac_slot + f] * REC;
            const double l0 = rec[24], l1 = rec[25];
            if (L.e) {
#pragma unroll
                for (int k = 0; k < 6; ++k) wex[k] += rec[28 + 2 * k] * l0 + rec[28 + 2 * k + 1] * l1;
            }
            if (L.t) wtd += rec[otd] * l0 + rec[otd + 1] * l1;
        }
        if (L.e) {
#pragma unroll
            for (int k = 0; k < 6; ++k) Wt[(size_t)(col_ex(L) + k) * ldw] = anchor >= 0 ? wex[k] : 0.0;
            row += 6;
        }
        if (L.t) { Wt[(size_t)col_td(L) * ldw] = anchor >= 0 ? wtd : 0.0; row += 1; }
    }
    if (L.big) { Wt[row * ldw] = b; ++row; }      // large-window path: the rhs travels as row Rc of W (ba_big_schur_kernel)
    for (; row < L.RcPad; ++row) Wt[row * ldw] = 0.0;
}

// grid nba * up(nwin, 8) workgroups of BA_ACC_NT threads.  Workgroups are dealt to the 8 XCDs round-robin by their
// linear id and every XCD has its own L2: id % 8 selects the window inside a group of 8 windows, so that ALL workgroups
// of a window run on one XCD and its projection records (read ~2.5 times by the tasks below) are fetched into one L2
// only.  Inside a window: workgroups 0 .. Kp-1 = the diagonal pose blocks; workgroup Kp = the blocks among ex / td (if
// estimated); afterwards wavefront tasks: the other owner blocks (host table io_task_list of packed-triangle block
// indices), then 64 landmarks per wavefront.
#ifndef BA_SOLVE_W8_TU
extern "C" __global__ __launch_bounds__(BA_ACC_NT) void ba_accumulate_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) {
    const BaLayout& L = *Lp;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, rest = bid >> 3;
    const int bx = rest % L.nba, w = (rest / L.nba) * 8 + xcd;
    if (w >= L.nwin) return;
    Ctx c;
    ctx_init(c, Lp, P, w);
    const double* ctl = c.sc + L.so_ctl;
    if (ctl[C_DONE] != 0.0) return;
    const int which = ((int)ctl[C_CUR]) ^ (ctl[C_PENDING] != 0.0 ? 1 : 0);
    double* buf = lin_buf(c, which);
    const double* recs = c.sc + L.so_rec;
    __shared__ double part[7 * 36 + 4];
    if (bx < L.Kp) { diag_pose_task(c, bx, recs, buf + L.bo_Sp, buf + L.bo_gp, part); return; }
    const int ng = L.e + L.t;
    if (ng && bx == L.Kp) { global_task(c, recs, buf + L.bo_Sp, buf + L.bo_gp, part); return; }
    const int g = (bx - L.Kp - (ng ? 1 : 0)) * (BA_ACC_NT / 64) + c.wave;
    const int nother = L.ntask - L.Kp - ng * (ng + 1) / 2;
    if (g < nother) owner_task(c, c.ia[L.io_task_list + g], c.lane, recs, buf + L.bo_Sp, buf + L.bo_gp);
    else {
        const int l = (g - nother) * 64 + c.lane;
        if (l < L.Lcap) landmark_task(c, l, recs, buf);
    }
}
#endif

// ================================================================================================
// Projection factors, fused: linearise + J^T J / J^T r in ONE kernel, the factor records never reach HBM.
// ================================================================================================
// One workgroup per window (no extrinsic / td columns, single-workgroup path; everything else keeps the two kernels above).
// The window's landmarks are taken in CHUNKS -- chunk c = the landmarks whose first factor index lies in [c la_chq, (c + 1) la_chq);
// the factor list is landmark-major, so a chunk is a contiguous factor range of at most la_chq + Kp - 2 factors -- and per chunk:
//   (1) thread per factor: ProjectionFactor::Evaluate + Cauchy corrector (projection_factor.cpp:21-121, corrector.cc) -> the record
//       goes to LDS as the 2 x 16 block X_f = [Ji (6) | Jj (6) | Jl | r | 0 0] (rows = the two residual components);
//   (2) camera blocks on MFMA: every factor of the (anchor i, target j) pair contributes X_f^T X_f, whose 16 x 16 product holds
//       Ji^T Ji, Jj^T Jj, Jj^T Ji, Ji^T r, Jj^T r at once.  A wavefront owns the pairs p = wave (mod 8), finds their factors in the
//       chunk with a ballot over the pair keys and feeds them two at a time (k = 4) to v_mfma_f64_16x16x4_f64 -- both operands of
//       X^T X are the SAME register (A[i][k] and B[k][j] sit in the same lane), fixed order -> bit-reproducible; the pair sums
//       are kept in LDS (90 doubles per pair) across the chunks;
//   (3) landmark sums (h, b, the W column) by one thread per landmark over its contiguous records, stores coalesced over landmarks.
// Afterwards the diagonal blocks are summed from the pair blocks in a fixed order and Sp / gp go to the linearisation buffer: what
// reaches HBM is Sp, gp, h, b, Wt and one cost partial -- the 42-double records (101.7 MB written + 87.9 MB read back per
// 256-window launch, profiles/r03s_pmc_hbm.txt) are gone, and so is one launch per round.
#define LA_NT BA_NT                // (imu_pass / prior_pass are written for BA_NT threads)
#define LA_RS 33                  // doubles per staged record: rows at 0 and 16, odd stride (conflict-free b64 stores by thread)
#define LA_NPW 6                  // wavefronts that own pair blocks; the other two of the workgroup take the landmark sums
#define LA_MAXP 13                // pairs per pair wavefront: Kp (Kp - 1) / 2 <= 78 for Kp <= 13
#define LA_QCAP 80                // pairs a window can have (78 for Kp = 13), padded
static_assert(LA_QCAP <= 128, "the pair prefix of ba_linacc_proj_kernel takes two pairs per lane of one wavefront");
#define LA_PAIR 90                // kept doubles of a pair block: ii 21 | jj 21 | ji 36 (row = target's column, col = anchor's) | gi 6 | gj 6
// pairs (i, j), i < j, enumerated by distance d = j - i first: q = (d - 1) Kp - (d - 1) d / 2 + i.  The pairs of one distance -- and the
// near-diagonal ones carry most factors -- are consecutive, so q mod LA_NPW deals them evenly to the pair wavefronts.
// inclusive prefix sum over the 64 lanes of a wavefront: four row_shr steps inside the rows of sixteen, the row totals through SGPRs
DEV int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);      // row_shr:1 (lanes without a source add 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);      // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);      // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);      // row_shr:8
    const int r0 = __builtin_amdgcn_readlane(v, 15), r1 = __builtin_amdgcn_readlane(v, 31), r2 = __builtin_amdgcn_readlane(v, 47);
    const int row = (int)(threadIdx.x & 63) >> 4;
    return v + (row > 0 ? r0 : 0) + (row > 1 ? r1 : 0) + (row > 2 ? r2 : 0);
}
DEV int la_pair_q(int i, int j, int Kp) { const int d = j - i; return (d - 1) * Kp - (d - 1) * d / 2 + i; }
__device__ __forceinline__ void linacc_body(const BaLayout* __restrict__ Lp, const BaPtrs& P) {
    const BaLayout L = layout_load(Lp);
    Ctx c;
    ctx_init(c, Lp, P, blockIdx.x);
    const double* ctl = c.sc + L.so_ctl;
    if (ctl[C_DONE] != 0.0) return;
    const int which = uni(((int)ctl[C_CUR]) ^ (ctl[C_PENDING] != 0.0 ? 1 : 0));
    const double* x = c.sc + L.so_x + which * L.nst;
    const double* lam = c.sc + L.so_lam + which * L.Lcap;
    glb_d* buf = AS_GLB(lin_buf(c, which));
    const glb_i* ia = AS_GLB_CI(c.ia);
    lds_d* rec = AS_LDS(LDSB);
    lds_d* Pb = AS_LDS(LDSB + L.la_P);
    lds_i* key = (lds_i*)(LDSB + L.la_key);          // (anchor << 4 | target) of the factor a chunk thread evaluated
    lds_i* lstart = key + L.la_chf;                   // [64] first landmark of every chunk
    lds_i* posx = lstart + 64;                        // [la_chf] where that thread's record was staged (records are grouped by pair)
    lds_i* pstart = posx + L.la_chf;                  // [LA_QCAP + 1] first record of every pair in the chunk
    lds_i* wcnt = pstart + LA_QCAP + 2;               // [LA_NT / 64][LA_QCAP] factors of pair q evaluated by wavefront w
    __shared__ double red[2 * (LA_NT / 64)];
    const int Kp = L.Kp, npair = Kp * (Kp - 1) / 2, chq = L.la_chq;
    const int nL = c.nL;
    DP_DECL;
    // ---- the prior and the IMU factors of the same point, by this workgroup too (they used to be a launch of their own,
    //      ba_linearize_imu_kernel).  Their LDS scratch (J0 copy, dx, partial sums / sqrt_info copies, weighted panels) is the start
    //      of the record area.
    {
        double share = 0.0;
        if (c.nprior > 0) {
            if (c.nprior * c.nprior <= 12 * LA_NT && 4 * L.Ncap <= 4 * LA_NT && (c.nprior * (c.nprior + 1) + 2 + 5 * L.Ncap) <= L.la_P)
                share = prior_staged(c, L, x, lin_buf(c, which) + L.bo_pr, lin_buf(c, which) + L.bo_gpr, LDSB);
            else
                share = prior_pass(c, x, lin_buf(c, which) + L.bo_pr, lin_buf(c, which) + L.bo_gpr, LDSB);
        }
        DP_ADD(23);
        share += imu_pass<true>(c, x, lin_buf(c, which) + L.bo_imuJ, 0, L.K - 1);
        const double tot_imu = block_sum(red, LA_NT / 64, c.lane, c.wave, share);
        for (int g = c.tid; g < L.nbl - L.nbf; g += LA_NT) c.sc[L.so_part + L.nbf + g] = g == 0 ? tot_imu : 0.0;
        __syncthreads();
        DP_ADD(22);
    }
    lds_d* xs = AS_LDS(LDSB + L.la_x);
    for (int k = c.tid; k < 7 * L.Kp + 9 * L.K + 8; k += LA_NT) xs[k] = AS_GLB_C(x)[k];
    const int nchunk = nL > 0 ? uni(ia[L.io_lm_fbeg + nL - 1] / chq + 1) : 0;     // (the LAST landmark's chunk: a chunk is never empty, landmarks have <= Kp - 1 factors)
    // first landmark of every chunk
    for (int t = c.tid; t < nL; t += LA_NT) {
        const int cl = ia[L.io_lm_fbeg + t] / chq;
        if (t == 0 || ia[L.io_lm_fbeg + t - 1] / chq != cl) lstart[cl] = t;
    }
    if (c.tid == 0) lstart[nchunk] = nL;
    for (int e = c.tid; e < npair * LA_PAIR; e += LA_NT) Pb[e] = 0.0;
    // Wavefronts 0 .. LA_NPW-1 own the pair blocks (pair q -> wavefront q mod LA_NPW), the last two the landmark sums: the two phases
    // of a chunk only READ the staged records, so they run side by side (MFMA + LDS reads there, VALU + HBM stores here).
    const bool pair_wave = c.wave < LA_NPW;
    __syncthreads();
    DP_ADD(24);
    // where the four accumulator entries of this lane go inside a pair block (-1: not kept)
    int pidx[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = (c.lane >> 4) + 4 * reg, col = c.lane & 15;
        int idx = -1;
        if (row < 6) {
            if (col <= row) idx = row * (row + 1) / 2 + col;              // ii
            else if (col == 13) idx = 78 + row;                           // gi
        } else if (row < 12) {
            if (col < 6) idx = 42 + (row - 6) * 6 + col;                  // ji
            else if (col <= row && col < 12) idx = 21 + (row - 6) * (row - 5) / 2 + (col - 6);   // jj
            else if (col == 13) idx = 84 + (row - 6);                     // gj
        }
        pidx[reg] = idx;
    }
    double cost = 0.0;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int l0 = lstart[ch], l1 = lstart[ch + 1];
        const int f0 = ia[L.io_lm_fbeg + l0], f1 = ia[L.io_lm_fbeg + l1];
        const int nf = uni(f1 - f0);
        if (nf > L.la_chf) __builtin_trap();                  // (a landmark with more than Kp - 1 factors: the packer never builds one)
        // ---- (1) where every record goes: the records of a chunk are staged GROUPED BY PAIR (a pair's factors are scattered over the
        //      landmark-major factor list: found by ballots over 64-factor blocks they came one or two per MFMA trip).  Stable
        //      counting sort by pair index: rank among the same pair inside the wavefront (ballots), wavefront counts and pair
        //      starts through LDS -- the order inside a pair is the factor order, whatever the timing: bit-reproducible sums.
        ProjIn pin;
        int q = -1, rank = 0;
        if (c.tid < nf) {
            proj_fetch_staged(c, f0 + c.tid, xs, lam, pin);
            q = la_pair_q(pin.i, pin.j, Kp);
        }
        for (int e = c.tid; e < (LA_NT / 64) * LA_QCAP; e += LA_NT) wcnt[e] = 0;
        __syncthreads();
        DP_ADD(45);
        {
            // the lanes of this wavefront with the same pair: seven ballots (one per bit of q < 128) instead of one trip per
            // distinct pair (a wavefront's 64 factors touch 30-40 pairs: ~600 instructions, round 6: ~80)
            const unsigned long long act = __ballot(q >= 0);
            unsigned long long same = act;
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                const bool bit = q >= 0 && ((q >> b) & 1);
                const unsigned long long mb = __ballot(bit);
                same &= bit ? mb : ~mb;
            }
            const unsigned long long lt = (1ull << c.lane) - 1ull;
            rank = __popcll(same & lt);
            if (q >= 0 && rank == 0) wcnt[c.wave * LA_QCAP + q] = __popcll(same);
        }
        __syncthreads();
        // pair totals, pair starts and the wavefronts' offsets inside a pair, by wavefront 0 alone: lane p takes the pairs p and p + 64
        // (all reads of a lane issued together, the running sum over the pairs on the DPP network).  It used to be a thread per pair
        // walking the wavefront counts through LDS and then ONE thread adding up the pair totals: 55 dependent LDS round trips.
        if (c.wave == 0) {
            int cw0[LA_NT / 64], cw1[LA_NT / 64];
            const int p0 = c.lane, p1 = c.lane + 64;
#pragma unroll
            for (int w = 0; w < LA_NT / 64; ++w) {
                cw0[w] = p0 < npair ? wcnt[w * LA_QCAP + p0] : 0;
                cw1[w] = p1 < npair ? wcnt[w * LA_QCAP + p1] : 0;
            }
            int t0 = 0, t1 = 0;
#pragma unroll
            for (int w = 0; w < LA_NT / 64; ++w) { const int a = cw0[w], b = cw1[w]; cw0[w] = t0; cw1[w] = t1; t0 += a; t1 += b; }
            const int i0 = wave_scan_incl(t0);
            const int tot0 = __builtin_amdgcn_readlane(i0, 63);
            const int i1 = wave_scan_incl(t1) + tot0;
            if (p0 < npair) {
                pstart[p0 + 1] = i0;
#pragma unroll
                for (int w = 0; w < LA_NT / 64; ++w) wcnt[w * LA_QCAP + p0] = i0 - t0 + cw0[w];      // first record of (pair, wavefront)
            }
            if (p1 < npair) {
                pstart[p1 + 1] = i1;
#pragma unroll
                for (int w = 0; w < LA_NT / 64; ++w) wcnt[w * LA_QCAP + p1] = i1 - t1 + cw1[w];
            }
            if (c.lane == 0) pstart[0] = 0;
        }
        __syncthreads();
        DP_ADD(46);
        // ---- (2) linearise, record to its sorted place
        if (c.tid < nf) {
            const int pos = wcnt[c.wave * LA_QCAP + q] + rank;
            double r[2], Ji[12], Jj[12], Jl[2];
            proj_eval<false, true, false>(pin.pi, pin.pj, pin.ex, pin.lam, pin.oi, pin.oj, 0.0, c.focal, c.tr, c.row, r, Ji, Jj, nullptr, Jl, nullptr);
            const double s = r[0] * r[0] + r[1] * r[1];
            const double sq = sqrt(1.0 / (1.0 + s));
            lds_d* qr = rec + pos * LA_RS;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                qr[k] = sq * Ji[k]; qr[16 + k] = sq * Ji[6 + k];
                qr[6 + k] = sq * Jj[k]; qr[22 + k] = sq * Jj[6 + k];
            }
            qr[12] = sq * Jl[0]; qr[28] = sq * Jl[1];
            qr[13] = sq * r[0]; qr[29] = sq * r[1];
            key[c.tid] = pin.i * 16 + pin.j;
            posx[c.tid] = pos;
            cost += log1p(s);
        }
        __syncthreads();
        DP_ADD(25);
        if (pair_wave) {
            // ---- (3) pair blocks on MFMA: the factors of a pair are contiguous; four per trip -- both operand reads, then both MFMAs
            double4_t acc[LA_MAXP];
#pragma unroll
            for (int k = 0; k < LA_MAXP; ++k) acc[k] = (double4_t){0, 0, 0, 0};
            const int k4 = c.lane >> 4, col = c.lane & 15;
            const int xo = (k4 & 1) * 16 + col;               // this lane's element of a record: row k4 & 1, column col
            const int hi = k4 >> 1;
            const bool incol = col < 14;
#pragma unroll
            for (int k = 0; k < LA_MAXP; ++k) {
                const int qk = c.wave + LA_NPW * k;
                if (qk >= npair) continue;                    // (uniform)
                const int pb0 = uni(pstart[qk]), pe0 = uni(pstart[qk + 1]);
                DP_ADD(29);
                for (int e = pb0; e < pe0; e += 4) {
                    const int f0r = e + hi, f1r = e + 2 + hi;
                    const double v0 = (f0r < pe0 && incol) ? rec[f0r * LA_RS + xo] : 0.0;
                    const double v1 = (f1r < pe0 && incol) ? rec[f1r * LA_RS + xo] : 0.0;
                    acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(v0, v0, acc[k], 0, 0, 0);
                    if (e + 2 < pe0) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(v1, v1, acc[k], 0, 0, 0);      // (uniform)
                }
                DP_ADD(30);
            }
            // D[row = (lane >> 4) + 4 reg][col = lane & 15] -> the kept entries of the pair block, added to the pair table (the first chunk
            // stores; later chunks read the four entries of a lane together, then write)
#pragma unroll
            for (int k = 0; k < LA_MAXP; ++k) {
                if (c.wave + LA_NPW * k >= npair) continue;
                lds_d* pb = Pb + (c.wave + LA_NPW * k) * LA_PAIR;
                double old4[4];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) old4[reg] = (ch > 0 && pidx[reg] >= 0) ? pb[pidx[reg]] : 0.0;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    if (pidx[reg] >= 0) pb[pidx[reg]] = old4[reg] + acc[k][reg];
            }
            DP_ADD(26);
        } else {
            // ---- (3) landmark sums: thread per landmark of the chunk, its records are contiguous; the W column is written as the
            //      factors come by (targets ascend), zeros in between, the anchor's rows (a sum over all factors) last
            for (int t = c.tid - LA_NPW * 64; t < l1 - l0; t += LA_NT - LA_NPW * 64) {
                const int l = l0 + t;
                glb_d* Wt = buf + L.bo_Wt + l;
                const size_t ldw = L.Lcap;
                const int fb = ia[L.io_lm_fbeg + l] - f0, fe = ia[L.io_lm_fbeg + l + 1] - f0;
                double h = 0.0, b = 0.0, wi[6] = {0, 0, 0, 0, 0, 0};
                const int anchor = fb < fe ? key[fb] >> 4 : -1;
                int jn = 0;
                for (int f = fb; f < fe; ++f) {
                    const lds_d* q = rec + posx[f] * LA_RS;
                    const int j = key[f] & 15;
                    const double l0v = q[12], l1v = q[28];
                    double v[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) v[k] = q[6 + k] * l0v + q[22 + k] * l1v;
                    h += l0v * l0v + l1v * l1v;
                    b += l0v * q[13] + l1v * q[29];
#pragma unroll
                    for (int k = 0; k < 6; ++k) wi[k] += q[k] * l0v + q[16 + k] * l1v;
                    for (; jn < j; ++jn) {
                        if (jn == anchor) continue;
#pragma unroll
                        for (int k = 0; k < 6; ++k) Wt[(size_t)(6 * jn + k) * ldw] = 0.0;
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) Wt[(size_t)(6 * j + k) * ldw] = v[k];
                    jn = j + 1;
                }
                for (; jn < Kp; ++jn) {
                    if (jn == anchor) continue;
#pragma unroll
                    for (int k = 0; k < 6; ++k) Wt[(size_t)(6 * jn + k) * ldw] = 0.0;
                }
                if (anchor >= 0) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) Wt[(size_t)(6 * anchor + k) * ldw] = wi[k];
                }
                for (int row = 6 * Kp; row < L.RcPad; ++row) Wt[(size_t)row * ldw] = 0.0;
                buf[L.bo_h + l] = h;
                buf[L.bo_b + l] = b;
            }
            DP_ADD(27);
        }
        __syncthreads();                                      // records and keys consumed
    }
    // ---- the columns of the unused landmark slots
    for (int w = c.tid; w < (L.Lcap - nL) * L.RcPad; w += LA_NT) {
        const int row = w / (L.Lcap - nL), l = nL + w % (L.Lcap - nL);
        buf[L.bo_Wt + (size_t)row * L.Lcap + l] = 0.0;
    }
    // ---- camera part: off-diagonal block (j, i) = the pair's Jj^T Ji; diagonal block a = its anchor share over the targets
    //      j > a, then its target share over the anchors i < a (fixed order)
    {
        const int nent = L.Rc * (L.Rc + 1) / 2;
        for (int e = c.tid; e < nent + L.Rc; e += LA_NT) {
            double s = 0.0;
            if (e < nent) {
                int row, colm;
                tri_decode(e, row, colm);
                const int br = row / 6, pr = row - 6 * br, bc = colm / 6, pc = colm - 6 * bc;
                if (br != bc) s = Pb[la_pair_q(bc, br, Kp) * LA_PAIR + 42 + pr * 6 + pc];
                else {
                    const int t = pr * (pr + 1) / 2 + pc;
                    for (int j = br + 1; j < Kp; ++j) s += Pb[la_pair_q(br, j, Kp) * LA_PAIR + t];
                    for (int i = 0; i < br; ++i) s += Pb[la_pair_q(i, br, Kp) * LA_PAIR + 21 + t];
                }
                buf[L.bo_Sp + e] = s;
            } else {
                const int g = e - nent, a = g / 6, k = g - 6 * a;
                for (int j = a + 1; j < Kp; ++j) s += Pb[la_pair_q(a, j, Kp) * LA_PAIR + 78 + k];
                for (int i = 0; i < a; ++i) s += Pb[la_pair_q(i, a, Kp) * LA_PAIR + 84 + k];
                buf[L.bo_gp + g] = s;
            }
        }
    }
    const double tot = block_sum(red, LA_NT / 64, c.lane, c.wave, cost);
    DP_ADD(28);
    if (c.tid < L.nbf) c.sc[L.so_part + c.tid] = c.tid == 0 ? tot : 0.0;
}
#ifndef BA_SOLVE_W8_TU
extern "C" __global__ __launch_bounds__(LA_NT) void ba_linacc_proj_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) { linacc_body(Lp, P); }
#endif

// ================================================================================================
// Solve kernel
// ================================================================================================
struct SolveLds {
    double *S, *XC, *D, *E, *dinv, *vec, *red, *wd, *z, *di;
    int* pmap;
    int ldc;
    // single-workgroup path: staged coupling rows of the block pair being eliminated (LDS), reduced column -> prior index (LDS),
    // parked rows X_k (HBM)
    double* ring;
    int* pinv;
    double* xp;
};
DEV void lds_carve(const BaLayout& L, double* sc, SolveLds& m) {
    m.S = LDSB + L.l_S; m.XC = nullptr; m.D = LDSB + L.l_D; m.E = LDSB + L.l_E; m.dinv = LDSB + L.l_dinv;
    m.vec = LDSB + L.l_vec; m.red = LDSB + L.l_red; m.wd = LDSB + L.l_wd; m.z = LDSB + L.l_z;
    m.pmap = nullptr;
    m.ldc = L.ldc;
    m.di = m.vec + V_DI * L.Rpad;
    m.ring = LDSB + L.l_ring; m.pinv = (int*)(LDSB + L.l_pinv); m.xp = sc + L.so_xp;
}
// large-window carve: S, the reduction scratch and 1/L_jj in LDS, the rest in HBM scratch (generic pointers: the helpers
// below do not care)
DEV void big_carve(const BaLayout& L, double* sc, SolveLds& m) {
    double* hb = sc + L.so_bigm;
    m.S = LDSB + L.l_S; m.red = LDSB + L.l_red; m.di = LDSB + L.l_di;
    m.XC = hb + L.l_XC; m.D = hb + L.l_D; m.E = hb + L.l_E; m.dinv = hb + L.l_dinv;
    m.vec = hb + L.l_vec; m.wd = hb + L.l_wd; m.z = hb + L.l_z;
    m.pmap = (int*)(hb + L.l_pmap);
    m.ldc = L.ldc;
    m.ring = nullptr; m.pinv = nullptr; m.xp = nullptr;
}

// The phase functions of the solve kernels are not inlined.  A Ctx / SolveLds handed over by reference has to sit in the caller's
// private stack frame (scratch stores at kernel entry, scratch / flat loads in every callee) and arrives as vector data, so the
// callee's address arithmetic runs on the VALU.  Everything in them is a function of the kernel arguments and the workgroup
// index: each phase rebuilds it from the kernarg segment with scalar loads instead (the reference arguments are dropped as dead
// arguments; the CPU emulation of tests/simt lays the launch arguments out like a kernarg segment, so it runs this code too).
DEV void phase_ctx(Ctx& c, BaLayout& L, SolveLds& m, int big) {        // big: 0 / 1, -1 = whichever path the layout says
    // (llvm.amdgcn.kernarg.segment.ptr is null outside a kernel; the implicit-argument pointer is handed down to callees and
    //  the hidden arguments start right behind the explicit ones: (const BaLayout* Lp, BaPtrs P) for both solve kernels)
    typedef vg_kernarg_ptr KArg;
    static_assert(sizeof(const BaLayout*) + sizeof(BaPtrs) == 96 && alignof(BaPtrs) == 8, "explicit kernel arguments of the solve kernels");
    KArg ka = (KArg)__builtin_amdgcn_implicitarg_ptr() - 96;
    const BaLayout* Lp;
    BaPtrs P;
    __builtin_memcpy(&Lp, ka, sizeof(Lp));
    __builtin_memcpy(&P, ka + 8, sizeof(BaPtrs));
    L = layout_load(Lp);
    ctx_init(c, Lp, P, blockIdx.x);
    if (big < 0 ? L.big != 0 : big != 0) big_carve(L, c.sc, m); else lds_carve(L, c.sc, m);
}
// the kernels compare what the phases will derive with their real arguments once (a wrong hidden-argument offset must not be silent)
#define PHASE_SELF_CHECK(cref) do { Ctx c_; BaLayout L_; SolveLds m_; phase_ctx(c_, L_, m_, 0); if (c_.sc != (cref).sc || c_.ia != (cref).ia || c_.pri != (cref).pri) __builtin_trap(); } while (0)
#define PHASE_ENTER(BIGV) Ctx c; BaLayout L; SolveLds m; phase_ctx(c, L, m, BIGV); VG_KEEP_ALIVE(&c_in); (void)m_in

// local column (0..29) of IMU factor f -> reduced column
DEV int imu_col(const BaLayout& L, int f, int lc) {
    if (lc < 6) return col_pose(L, f) + lc;
    if (lc < 15) return col_sb(L, f) + lc - 6;
    if (lc < 21) return col_pose(L, f + 1) + lc - 15;
    return col_sb(L, f + 1) + lc - 21;
}
// The stored blocks of the reduced system with their address spaces spelled out: S always in LDS; XC / D / E / the R-vectors /
// the prior column map in LDS (MV = lds_d) or in HBM (large-window path, MV = glb_d).
template <typename MV, typename MI>
struct SysPtrs {
    lds_d* S;
    MV *XC, *D, *E, *vec;
    MI* pmap;
    int ldc;
};
template <bool BIG>
DEV SysPtrs<typename MovT<BIG>::D, typename MovT<BIG>::I> sys_ptrs(const SolveLds& m) {
    typedef typename MovT<BIG>::D MV;
    typedef typename MovT<BIG>::I MI;
    SysPtrs<MV, MI> q;
    q.S = AS_LDS(m.S); q.XC = (MV*)m.XC; q.D = (MV*)m.D; q.E = (MV*)m.E; q.vec = (MV*)m.vec; q.pmap = (MI*)m.pmap; q.ldc = m.ldc;
    return q;
}
// add v to the Hessian entry (ca, cb) of the reduced system, ca != cb or ca == cb, wherever that entry is stored:
//   camera x camera -> packed S;  sb_k x sb_k -> D_k (full 9x9, both triangles);  sb_k x sb_k-1 -> E_k;
//   sb_k x camera -> XC row 9k+r (the coupling block that the chain elimination turns into X_k)
template <typename Q>
DEV void hess_add(const BaLayout& L, const Q& q, int ca, int cb, double v) {
    if (ca < cb) { const int t = ca; ca = cb; cb = t; }
    const int Rc = L.Rc;
    if (ca < Rc) { q.S[tri(ca, cb)] += v; return; }
    const int ka = (ca - Rc) / 9, ra = (ca - Rc) - 9 * ka;
    if (cb < Rc) { q.XC[(9 * ka + ra) * q.ldc + cb] += v; return; }
    const int kb = (cb - Rc) / 9, rb = (cb - Rc) - 9 * kb;
    if (ka == kb) {
        q.D[81 * ka + 9 * ra + rb] += v;
        if (ra != rb) q.D[81 * ka + 9 * rb + ra] += v;
    } else {
        q.E[81 * ka + 9 * ra + rb] += v;          // ka == kb + 1 (the host rejects priors coupling non-adjacent speed-bias blocks)
    }
}

// Unscaled Gauss-Newton system of the current point, LARGE-WINDOW path: S (camera, packed lower, LDS), g, chain blocks D, E, XC (HBM).
// Sp / gp = the rank-summed camera J^T J / J^T r of the projection factors.  (Single-workgroup path: assemble_small below.)
NOINL void assemble_big(const Ctx& c_in, const SolveLds& m_in, const double* buf_, const double* Sp_, const double* gp_) {
    PHASE_ENTER(true);
    typedef glb_d MV;
    const auto q = sys_ptrs<true>(m);
    const glb_d* buf = AS_GLB_C(buf_);
    const glb_d* Sp = AS_GLB_C(Sp_);
    const glb_d* gp = AS_GLB_C(gp_);
    const int Rc = L.Rc, K = L.K;
    const int camtri = Rc * (Rc + 1) / 2;
    MV* g = q.vec + V_G * L.Rpad;
    __syncthreads();
    // (XC / D / E live in HBM and were cleared by the Schur kernel's extra workgroups)
    // (a copy loop HBM -> LDS waits for every load before its store: eight loads of a thread are requested together, round 6)
    for (int k0 = c.tid; k0 < camtri; k0 += 8 * BA_NT) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + u * BA_NT; v[u] = Sp[k < camtri ? k : 0]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = k0 + u * BA_NT; if (k < camtri) q.S[k] = v[u]; }
    }
    for (int k = c.tid; k < L.Rpad; k += BA_NT) g[k] = k < Rc ? gp[k] : 0.0;
    __syncthreads();
    // ---- IMU Hessian blocks: factors k and k+1 share the blocks of frame k+1, so even and odd factors are added in two
    //      rounds (inside a round every entry has exactly one writer)
    {
        const int nimu = K - 1;
        const glb_i* valid = AS_GLB_CI(c.ia + L.io_imu_valid);
        const glb_d* imuJ = buf + L.bo_imuJ;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int nf = (nimu - par + 1) / 2;
            {
                // Thread = entry e of every factor of this parity, eight factors at a time: all loads of the IMU blocks first,
                // then the read-modify-writes (inside a round every entry has exactly one writer, so the slots are distinct).
                // One trip per entry would be a chain of HBM round trips; most targets are in HBM too.
                // Entries of the camera part live in LDS: they are added directly.
                // The destination of entry e is the same for every factor up to a shift by the frame index, so it is decoded ONCE
                // per thread: camera x camera -> packed S at (6 f + xa, 6 f + xb); everything else -> base + f * stride in the
                // block that stores it (XC: 9 ldc + 6, D / E: 81, gradient: 6 or 9), D entries off the diagonal with a mirror.
                const int e = c.tid;
                int xa = -1, xb = 0, stride = 0;
                MV* base0 = nullptr;
                MV* base1 = nullptr;
                if (e < 465) {
                    int a, b;
                    tri_decode(e, a, b);                             // a >= b
                    const int ba = a < 6 ? 0 : a < 15 ? 1 : a < 21 ? 2 : 3, oa = a - (ba == 0 ? 0 : ba == 1 ? 6 : ba == 2 ? 15 : 21);
                    const int bb = b < 6 ? 0 : b < 15 ? 1 : b < 21 ? 2 : 3, ob = b - (bb == 0 ? 0 : bb == 1 ? 6 : bb == 2 ? 15 : 21);
                    const bool sa = ba & 1, sb = bb & 1;             // speed-bias block?
                    if (!sa && !sb) { xa = oa + 3 * ba; xb = ob + 3 * bb; }                          // pose_f / pose_f+1
                    else if (sa && !sb) { base0 = q.XC + (9 * (ba >> 1) + oa) * q.ldc + 3 * bb + ob; stride = 9 * q.ldc + 6; }
                    else if (!sa && sb) { base0 = q.XC + ob * q.ldc + 6 + oa; stride = 9 * q.ldc + 6; }  // pose_f+1 x sb_f
                    else if (ba == bb) {
                        base0 = q.D + 81 * (ba >> 1) + 9 * oa + ob; stride = 81;
                        if (oa != ob) base1 = q.D + 81 * (ba >> 1) + 9 * ob + oa;
                    } else { base0 = q.E + 81 + 9 * oa + ob; stride = 81; }                          // sb_f+1 x sb_f
                } else if (e < 495) {
                    const int lc = e - 465;
                    const int bl = lc < 6 ? 0 : lc < 15 ? 1 : lc < 21 ? 2 : 3, ol = lc - (bl == 0 ? 0 : bl == 1 ? 6 : bl == 2 ? 15 : 21);
                    if (bl & 1) { base0 = g + Rc + 9 * (bl >> 1) + ol; stride = 9; }
                    else { base0 = g + 3 * bl + ol; stride = 6; }
                }
                for (int i0 = 0; i0 < nf; i0 += 8) {
                    MV* p0[8];
                    MV* p1[8];
                    double v[8], t0[8], t1[8];
                    bool on[8];
                    int vld[8];
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {             // the loads of the chunk, nothing else: value and validity flag
                        const int f = 2 * (i0 + qq) + par;       // in ONE round trip (clamped addresses, masked afterwards)
                        const int fc = f < nimu ? f : nimu - 1;
                        vld[qq] = valid[fc];
                        v[qq] = imuJ[fc * 512 + e];
                    }
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        on[qq] = i0 + qq < nf && e < 495 && vld[qq];
                        v[qq] = on[qq] ? v[qq] : 0.0;
                    }
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        const int f = 2 * (i0 + qq) + par;
                        p0[qq] = (on[qq] && base0) ? base0 + f * stride : nullptr;
                        p1[qq] = (on[qq] && base1) ? base1 + f * stride : nullptr;
                        if (on[qq] && xa >= 0) q.S[tri(6 * f + xa, 6 * f + xb)] += v[qq];
                    }
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) { t0[qq] = p0[qq] ? *p0[qq] : 0.0; t1[qq] = p1[qq] ? *p1[qq] : 0.0; }
#pragma unroll
                    for (int qq = 0; qq < 8; ++qq) {
                        if (p0[qq]) *p0[qq] = t0[qq] + v[qq];
                        if (p1[qq]) *p1[qq] = t1[qq] + v[qq];
                    }
                }
            }
            __syncthreads();
        }
    }
    // ---- prior: H += J0^T J0, g += J0^T r
    if (c.nprior) {
        const int n = c.nprior;
        const glb_d* Hp = AS_GLB_C(c.sc + L.so_Hp);
        const glb_d* gpr = buf + L.bo_gpr;        // J0^T r, formed by the linearisation kernel
        const int nent = n * (n + 1) / 2 + n;
        for (int w0 = 0; w0 < nent; w0 += 8 * BA_NT) {
            // eight entries per thread: indices and the eight HBM loads first, then the adds (every entry of H is touched by
            // exactly one prior entry: the column map is injective)
            int ca[8], cb[8];
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int w = w0 + c.tid + u * BA_NT;
                ca[u] = -1; cb[u] = -1; v[u] = 0.0;
                if (w < nent) {
                    if (w >= n * (n + 1) / 2) {
                        const int a = w - n * (n + 1) / 2;
                        ca[u] = q.pmap[a]; cb[u] = -2;          // gradient entry
                        v[u] = gpr[a];
                    } else {
                        int a, b;
                        tri_decode(w, a, b);
                        ca[u] = q.pmap[a]; cb[u] = q.pmap[b];
                        v[u] = Hp[a * L.Ncap + b];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (cb[u] == -2) { if (ca[u] >= 0) g[ca[u]] += v[u]; }
                else if (ca[u] >= 0 && cb[u] >= 0) hess_add(L, q, ca[u], cb[u], v[u]);
            }
        }
    }
    __syncthreads();
}

// Unscaled Gauss-Newton system of the current point, single-workgroup path (round 5): every STORED entry gathers its terms --
// camera part S (packed lower) = Sp + IMU pose blocks + prior, g, the chain blocks D_k / E_k -- instead of the former scatter
// (clear, copy, two parity rounds of read-modify-writes for the IMU blocks, a table-driven round for the prior: four barriers and
// a slot table the prologue had to build per solve).  Where an entry's terms live is a function of the layout alone, so the host
// builds the gather plan once per layout (build_asm_plan, ba_host.hip; AsmPlanEntry, ba_layout.h): per entry one 16-byte plan
// read, up to four value loads (all independent), three adds in the order of the scatter -- projection part, even IMU factor, odd
// IMU factor, prior -- and one LDS store; one barrier at the end.  Whether an IMU factor exists and what the prior holds is decided
// here: the validity bit of the factor index, the column -> prior-index map `pinv` (built first; chain_schur uses it too).  The
// coupling rows sb x camera are NOT assembled: the chain elimination gathers them block by block (chain_schur).
#define ASM_NB 9                                  // plan entries per thread and trip
NOINL void assemble_small(const Ctx& c_in, const SolveLds& m_in, const double* buf_) {
    PHASE_ENTER(false);
    const glb_d* buf = AS_GLB_C(buf_);
    const glb_d* imuJ = buf + L.bo_imuJ;
    const glb_d* gpr = buf + L.bo_gpr;            // J0^T r, formed by the linearisation kernel
    const glb_d* Hp = AS_GLB_C(c.sc + L.so_Hp);   // J0^T J0 (lower triangle, row stride Ncap), formed by the prologue kernel
    lds_d* const lds0 = AS_LDS(LDSB);
    lds_i* const pinv = (lds_i*)m.pinv;
    const int R = L.R, Ncap = L.Ncap;
    const AsmPlanEntry* plan = (const AsmPlanEntry*)((const char*)c.Lp + L.pl_off);      // (16 bytes, aligned: one global_load_dwordx4 per entry)
    const int n = L.pl_n;
    // the first trip's plan entries are requested before anything else
    int4 pe[ASM_NB];                               // {dst, base, imu, cols}
#pragma unroll
    for (int u = 0; u < ASM_NB; ++u) { const int w = c.tid + u * SV_NT; pe[u] = vg_load_int4(plan + (w < n ? w : n - 1)); }
    unsigned vm;                                   // bit f: IMU factor f is present
    {
        const glb_i* valid = AS_GLB_CI(c.ia + L.io_imu_valid);
        const int v = c.lane < L.K - 1 ? valid[c.lane] : 0;
        vm = (unsigned)__ballot(v != 0);
    }
    __syncthreads();
    for (int k = c.tid; k < R; k += SV_NT) pinv[k] = -1;
    __syncthreads();
    if (c.nprior) {
        const glb_i* kind = AS_GLB_CI(c.ia + L.io_pb_kind);
        const glb_i* off = AS_GLB_CI(c.ia + L.io_pb_off);
        const glb_i* pcol = AS_GLB_CI(c.ia + L.io_pb_col);
        for (int blk = c.tid; blk < c.nblk; blk += SV_NT) {
            const int kd = kind[blk], o = off[blk], pc = pcol[blk];
            const int sz = (kd == VG_BLK_SPEEDBIAS) ? 9 : (kd == VG_BLK_TD ? 1 : 6);
            if (pc >= 0)
                for (int k = 0; k < sz; ++k) pinv[pc + k] = o + k;
        }
    }
    __syncthreads();
    DP_DECL;
    const bool has_prior = c.nprior != 0;
    for (int w0 = 0; w0 < n; w0 += ASM_NB * SV_NT) {
        double vb[ASM_NB], ve[ASM_NB], vo[ASM_NB], vp[ASM_NB];
        int dst[ASM_NB];
#pragma unroll
        for (int u = 0; u < ASM_NB; ++u) {
            const int4 e = pe[u];
            const bool in = w0 + c.tid + u * SV_NT < n;
            dst[u] = in ? (e.x & 0x0fffffff) : -1;
            const int ie = (e.z & 0xffff) - 1, io = ((e.z >> 16) & 0xffff) - 1;
            const bool one = ie >= 0 && ((vm >> (ie >= 0 ? ie >> 9 : 0)) & 1u), two = io >= 0 && ((vm >> (io >= 0 ? io >> 9 : 0)) & 1u);
            const double b0 = buf[e.y >= 0 ? e.y : 0], e0 = imuJ[one ? ie : 0], o0 = imuJ[two ? io : 0];
            vb[u] = e.y >= 0 ? b0 : 0.0;
            ve[u] = one ? e0 : 0.0;
            vo[u] = two ? o0 : 0.0;
            // prior term: Hessian entries J0^T J0 [pinv ca][pinv cb], gradient entries J0^T r [pinv ca]
            const bool grad = (e.x >> 28) != 0;
            const int ca = grad ? e.w : (e.w >> 16), cb = grad ? e.w : (e.w & 0xffff);
            const int ia = e.w >= 0 ? pinv[ca] : -1, ib = e.w >= 0 ? pinv[cb] : -1;
            const bool pin = has_prior && ia >= 0 && ib >= 0;
            const int hi = ia > ib ? ia : ib, lo = ia > ib ? ib : ia;
            const glb_d* pp = grad ? gpr + (pin ? ia : 0) : Hp + (pin ? hi * Ncap + lo : 0);
            const double p0 = *pp;
            vp[u] = pin ? p0 : 0.0;
        }
        // the next trip's plan entries travel while this trip's values arrive
        if (w0 + ASM_NB * SV_NT < n) {
#pragma unroll
            for (int u = 0; u < ASM_NB; ++u) { const int w = w0 + ASM_NB * SV_NT + c.tid + u * SV_NT; pe[u] = vg_load_int4(plan + (w < n ? w : n - 1)); }
        }
#pragma unroll
        for (int u = 0; u < ASM_NB; ++u)
            if (dst[u] >= 0) lds0[dst[u]] = ((vb[u] + ve[u]) + vo[u]) + vp[u];
    }
    DP_ADD(16);
    __syncthreads();
    DP_ADD(17);
}

// diagonal entry k of the (unscaled) Hessian as stored by assemble()
DEV double hess_diag(const BaLayout& L, const SolveLds& m, int k) {
    if (k < L.Rc) return m.S[tri(k, k)];
    const int kk = (k - L.Rc) / 9, r = (k - L.Rc) - 9 * kk;
    return m.D[81 * kk + 10 * r];
}

// H <- diag(sc) H diag(sc) + mu*Dg^2 on every stored block; rhs (scaled gradient) into the augmented row Rc of S and
// column Rc of XC.  Also returns this thread's share of  t^T H~ t  over the reduced block (H~ = scaled, UN-damped
// Hessian, t = V_T = gt / Dg): the Cauchy-point denominator |J~ t|^2 of DoglegStrategy::ComputeCauchyPoint without a
// second pass over the factors.
template <bool BIG>
NOINL double build_scaled(const Ctx& c_in, const SolveLds& m_in, double mu) {
    PHASE_ENTER(BIG);
    constexpr int NT = BIG ? BA_NT : SV_NT;        // threads of the calling kernel
    mu = uni(mu);                             // (arguments arrive in vector registers)
    typedef typename MovT<BIG>::D MV;
    const auto mq = sys_ptrs<BIG>(m);
    const MV* g = mq.vec + V_G * L.Rpad;
    const MV* sc = mq.vec + V_SC * L.Rpad;
    const MV* dg = mq.vec + V_DG * L.Rpad;
    const MV* tv = mq.vec + V_T * L.Rpad;
    const int Rc = L.Rc, K = L.K, ldc = mq.ldc;
    const int n = Rc * (Rc + 1) / 2;
    double q = 0.0;
    if constexpr (BIG) {
        // large-window path: the vectors live in HBM and every entry of S gathers four of their elements -- the camera parts of the
        // scaling and of t are staged in the chain elimination's LDS scratch first (free here: 2 Rc <= 576 checked by the layout), so
        // the 18.7K entries of a 193-wide system read LDS instead of walking 37 dependent HBM round trips per thread
        lds_d* scs = AS_LDS(LDSB + L.l_cz);
        lds_d* tvs = scs + Rc;
        if (2 * Rc <= 6 * 96) {
            __syncthreads();
            for (int k = c.tid; k < Rc; k += NT) { scs[k] = sc[k]; tvs[k] = tv[k]; }
            __syncthreads();
            for (int w = c.tid; w < n; w += NT) {
                int a, b;
                tri_decode(w, a, b);
                double v = mq.S[w] * scs[a] * scs[b];
                q += v * tvs[a] * tvs[b] * (a == b ? 1.0 : 2.0);
                if (a == b) v += mu * dg[a] * dg[a];
                mq.S[w] = v;
            }
        } else {
            for (int w = c.tid; w < n; w += NT) {
                int a, b;
                tri_decode(w, a, b);
                double v = mq.S[w] * sc[a] * sc[b];
                q += v * tv[a] * tv[b] * (a == b ? 1.0 : 2.0);
                if (a == b) v += mu * dg[a] * dg[a];
                mq.S[w] = v;
            }
        }
    } else {
    for (int w = c.tid; w < n; w += NT) {
        int a, b;
        tri_decode(w, a, b);
        double v = mq.S[w] * sc[a] * sc[b];
        q += v * tv[a] * tv[b] * (a == b ? 1.0 : 2.0);
        if (a == b) v += mu * dg[a] * dg[a];
        mq.S[w] = v;
    }
    }
    for (int k = c.tid; k < Rc; k += NT) mq.S[tri(Rc, k)] = sc[k] * g[k];
    if (c.tid == 0) mq.S[tri(Rc, Rc)] = 0.0;
    for (int w = c.tid; w < 81 * K; w += NT) {
        const int k = w / 81, e = w - 81 * k, r = e / 9, cc = e - 9 * r;
        const int ca = Rc + 9 * k + r, cb = Rc + 9 * k + cc;
        double v = mq.D[w] * sc[ca] * sc[cb];
        q += v * tv[ca] * tv[cb];
        if (r == cc) v += mu * dg[ca] * dg[ca];
        mq.D[w] = v;
        if (k > 0) {
            const int cp = Rc + 9 * (k - 1) + cc;
            const double ve = mq.E[w] * sc[ca] * sc[cp];
            q += 2.0 * ve * tv[ca] * tv[cp];
            mq.E[w] = ve;
        }
    }
    if constexpr (BIG) {
        // (XC in HBM; the single-workgroup path scales its coupling rows as the chain elimination gathers them: chain_schur)
        // large-window path (XC in HBM): before the elimination a speed-bias row of frame k is non-zero only in the columns of
        // poses k-1, k, k+1 (IMU factors k-1 and k) -- and anywhere if the prior holds that speed-bias block
        const glb_i* kind = AS_GLB_CI(c.ia + L.io_pb_kind);
        const glb_i* idx = AS_GLB_CI(c.ia + L.io_pb_idx);
        for (int w = c.tid; w < 9 * K * 20; w += NT) {
            const int row = w / 20, e = w - 20 * row, k = row / 9;
            const int ca = Rc + row;
            if (e == 19) { mq.XC[row * ldc + Rc] = sc[ca] * g[ca]; continue; }
            bool inprior = false;
            for (int b = 0; b < c.nblk; ++b) inprior = inprior || (kind[b] == VG_BLK_SPEEDBIAS && idx[b] == k);
            if (inprior) continue;                     // handled densely below
            const int col = 6 * (k - 1) + e;           // e = 0 .. 17
            if (e >= 18 || col < 0 || col >= 6 * L.Kp) continue;
            const double v = mq.XC[row * ldc + col] * sc[ca] * sc[col];
            q += 2.0 * v * tv[ca] * tv[col];
            mq.XC[row * ldc + col] = v;
        }
        for (int b = 0; b < c.nblk; ++b) {
            if (kind[b] != VG_BLK_SPEEDBIAS) continue;
            const int k = idx[b];
            for (int w = c.tid; w < 9 * Rc; w += NT) {
                const int r = w / Rc, col = w - r * Rc, row = 9 * k + r, ca = Rc + row;
                const double v = mq.XC[row * ldc + col] * sc[ca] * sc[col];
                q += 2.0 * v * tv[ca] * tv[col];
                mq.XC[row * ldc + col] = v;
            }
        }
    }
    return q;
}

DEV double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
// 1/sqrt(x): hardware seed + two Newton steps (full double precision, ~10 dependent ops)
DEV double rsqrt_nr(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    y = y * fma(-hx * y, y, 1.5);
    return y;
}

// 9x9 diagonal block of the speed-bias chain (chain_schur below; chain_eliminate_big on the large-window path): update from the
// neighbour(s) eliminated earlier, then Cholesky.  Returns false (wave-uniform) on a non-positive pivot.
// One wavefront, the block in the accumulator layout of v_mfma_f64_16x16x4_f64 (padded to 16x16 with the identity), as in
// cholesky_aug(): the neighbour update is three MFMAs per coupling block (k = 9 padded to 12; the B operand of A A^T is the A
// operand itself, negated), a pivot is v_readlane + 1/sqrt + one rank-1 MFMA, and no LDS access sits between the first load
// and the final stores.
template <typename P, typename PC>
DEV bool chain_factor(int lane, P Dk, P dinvk, int kind, PC Xc, PC Xu) {
    // kind & 1: Xc = Xe of the upper neighbour as [p][r]  ->  D[r][c] -= sum_p Xc[9p + r] Xc[9p + c]
    // kind & 2: Xu = XuT of the lower neighbour as [r][p]  ->  D[r][c] -= sum_p Xu[9r + p] Xu[9c + p]
    const int jc = lane & 15, kq = lane >> 4;
    double4_t dg;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int i = kq + 4 * reg;
        const bool in = i < 9 && jc < 9;
        const double dv = Dk[in ? 9 * (i > jc ? i : jc) + (i > jc ? jc : i) : 0];
        dg[reg] = in ? dv : (i == jc ? 1.0 : 0.0);
    }
    if (kind & 1) {
        double av[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int pp = 4 * s + kq;
            const bool in = pp < 9 && jc < 9;
            const double v = Xc[in ? 9 * pp + jc : 0];
            av[s] = in ? v : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) dg = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], -av[s], dg, 0, 0, 0);
    }
    if (kind & 2) {
        double av[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int pp = 4 * s + kq;
            const bool in = pp < 9 && jc < 9;
            const double v = Xu[in ? 9 * jc + pp : 0];
            av[s] = in ? v : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) dg = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], -av[s], dg, 0, 0, 0);
    }
    double msk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) msk[k] = kq == k ? 1.0 : 0.0;
    double4_t ld = {0, 0, 0, 0}, di = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const double piv = readlane_d(dg[r >> 2], 16 * (r & 3) + r);
        const double dm = rsqrt_nr(piv) * msk[r & 3];                 // a bad pivot turns everything behind it into NaN / Inf
        const double lv = dg[r >> 2] * dm, nlv = dg[r >> 2] * -dm;
        dg = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, nlv, dg, 0, 0, 0);
        ld[r >> 2] += lv;
        di[r >> 2] += dm;
    }
    bool good = true;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int r = kq + 4 * q;
        if (r < 9) {
            if (!(di[q] > 0.0) || !(di[q] < 1e300)) good = false;
            if (jc >= r && jc < 9) Dk[9 * jc + r] = ld[q];
            if (jc == 0) dinvk[r] = di[q];
        }
    }
    return !__any(!good);
}
// x -= (coupling block)^T x_n for one column: coef[p * SP + r * SR] = coupling entry (row r of this block, row p of the neighbour's X).
// Three rows at a time, the scheduler fenced in between (81 hoisted LDS reads would not fit the register file).  _c: strides known at
// compile time (both sweeps store their coupling blocks as [p][r]: immediate offsets; the block itself may differ per lane).
template <int SP, int SR>
DEV void chain_upd_c(double* x, const lds_d* coef, const double* xn) {
#pragma unroll
    for (int r3 = 0; r3 < 9; r3 += 3) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0;
        const lds_d* cf = coef + r3 * SR;
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            s0 += cf[p * SP] * xn[p];
            s1 += cf[p * SP + SR] * xn[p];
            s2 += cf[p * SP + 2 * SR] * xn[p];
        }
        x[r3] -= s0; x[r3 + 1] -= s1; x[r3 + 2] -= s2;
        __builtin_amdgcn_sched_barrier(0);
    }
}
// ---- the same factorisation in three parts, so that one wavefront can walk TWO blocks in lock step (their pivot chains are
//      independent: the second block hides in the latencies of the first)
struct ChainFac { double4_t dg, ld, di; };
// (XU_PC: the lower neighbour's block is stored [p][c] like the upper one's -- chain_schur; false: [c][p] -- the large-window path)
template <bool XU_PC, typename P, typename PC>
DEV void cf_load(ChainFac& f, int lane, P Dk, int kind, PC Xc, PC Xu) {
    const int jc = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int i = kq + 4 * reg;
        const bool in = i < 9 && jc < 9;
        const double dv = Dk[in ? 9 * (i > jc ? i : jc) + (i > jc ? jc : i) : 0];
        f.dg[reg] = in ? dv : (i == jc ? 1.0 : 0.0);
    }
    if (kind & 1) {
        double av[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int pp = 4 * s + kq;
            const bool in = pp < 9 && jc < 9;
            const double v = Xc[in ? 9 * pp + jc : 0];
            av[s] = in ? v : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) f.dg = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], -av[s], f.dg, 0, 0, 0);
    }
    if (kind & 2) {
        double av[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int pp = 4 * s + kq;
            const bool in = pp < 9 && jc < 9;
            const double v = Xu[in ? (XU_PC ? 9 * pp + jc : 9 * jc + pp) : 0];
            av[s] = in ? v : 0.0;
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) f.dg = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], -av[s], f.dg, 0, 0, 0);
    }
    f.ld = (double4_t){0, 0, 0, 0};
    f.di = (double4_t){0, 0, 0, 0};
}
template <int r>
DEV void cf_pivot(ChainFac& f, const double* msk) {
    const double piv = readlane_d(f.dg[r >> 2], 16 * (r & 3) + r);
    const double dm = rsqrt_nr(piv) * msk[r & 3];                     // a bad pivot turns everything behind it into NaN / Inf
    const double lv = f.dg[r >> 2] * dm, nlv = f.dg[r >> 2] * -dm;
    f.dg = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, nlv, f.dg, 0, 0, 0);
    f.ld[r >> 2] += lv;
    f.di[r >> 2] += dm;
}
template <typename P>
DEV bool cf_store(const ChainFac& f, int lane, P Dk, P dinvk) {
    const int jc = lane & 15, kq = lane >> 4;
    bool good = true;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int r = kq + 4 * q;
        if (r < 9) {
            if (!(f.di[q] > 0.0) || !(f.di[q] < 1e300)) good = false;
            if (jc >= r && jc < 9) Dk[9 * jc + r] = f.ld[q];
            if (jc == 0) dinvk[r] = f.di[q];
        }
    }
    return !__any(!good);
}

// ---- single-workgroup path, round 5: chain elimination and Schur complement in ONE phase, coupling rows not kept in LDS --------
// The speed-bias chain is eliminated from both ends towards the middle block mid = K / 2 as before ("burn at both ends"):
//   top sweep    k = K-1 .. mid+1 : block k is coupled to k-1 through E_k (rows sb_k, columns sb_k-1)
//   bottom sweep k = 0 .. mid-1   : block k is coupled to k+1 through E_k+1^T
//   last         k = mid          : receives the updates of both neighbours
// but a THREAD owns a COLUMN of [C_k | g_k | coupling block] for a whole sweep (threads 0 .. Rc+9: top, Rc+10 .. 2 Rc+19: bottom;
// wavefronts 0-2): the update a block receives from the neighbour eliminated one step earlier involves only the same column of
// that neighbour -- which this thread solved itself and still holds in registers -- and the neighbour's 9x9 coupling block (LDS).
// What the former layout kept for the whole solve (100 rows x ldc = 64 KB of LDS: the reason only one window fitted a CU) shrinks
// to one slot of 18 rows: the solved rows X_k = L_k^-1 [C_k | g_k] of the block pair of the step, staged for the matrix cores, which
// add S -= X^T X to register accumulators right away; the rows are parked in HBM (so_xp) for the back substitution.  The unscaled
// coupling rows are not assembled into LDS either: the 9 x 18 entries of a block that the IMU factors touch are gathered one step
// ahead into a small staging area (two loads per thread), a column scales its nine entries and -- for the Cauchy-point term
// t^T H~ t -- sums them on the way; a block the prior holds adds its J0^T J0 row.
// One step (both sweeps at once):
//   (A) wavefront 3: both diagonal blocks in lock step -- D_k -= (update of the neighbour eliminated one step earlier), 9x9
//       Cholesky in MFMA accumulators;  every column thread: x = scaled raw column - (coupling block)^T x_prev
//   (B) x <- L_k^-1 x (L from LDS) -> ring slot, HBM, registers;  coupling-block columns -> Xe / XuT in the slot E_k
//   (C) all wavefronts: S accumulators += (18 staged rows)^T (18 staged rows) on v_mfma_f64_16x16x4
// then the landmark columns, 32 at a time, exactly as before.  Storage afterwards: D_k = L_k; the coupling block of a pair
// (k, k-1) lives in the slot E_k: for a TOP block k it holds Xe_k = L_k^-1 E_k as [p][c] (p = row of X_k, c = column sb_k-1), for
// a BOTTOM block k-1 it holds L_k-1^-1 E_k^T as [p][c] too (c = column sb_k; the large-window path keeps [c][p]); X in HBM.  Returns this thread's share of the coupling part of
// t^T H~ t; *flag (m.red + 24) = 0 on a non-positive pivot.
#define SCHUR_LW 32                               // landmarks per staged tile
#define SCHUR_LD (SCHUR_LW + 1)
#define SCHUR_PF ((96 * SCHUR_LW + SV_NT - 1) / SV_NT)
// SCHUR_TPW = 16x16 tiles of S per wavefront: 15 tiles (RcPad = 80) over four wavefronts = 4, 21 tiles (RcPad = 96) = 6
template <int SCHUR_TPW>
NOINL double chain_schur(const Ctx& c_in, const SolveLds& m_in, const double* buf_, double mu) {
    PHASE_ENTER(false);
    mu = uni(mu);
    const int K = L.K, Rc = L.Rc, RcPad = L.RcPad, ldc = m.ldc, Ncap = L.Ncap;
    const int mid = K / 2, nstep = mid;            // (K - 1 - mid <= mid)
    lds_d* const S = AS_LDS(m.S);
    lds_d* const D = AS_LDS(m.D);
    lds_d* const E = AS_LDS(m.E);
    lds_d* const dinv = AS_LDS(m.dinv);
    lds_d* const ring = AS_LDS(m.ring);            // rows 0..8: top block of the step, rows 9..17: bottom block
    lds_d* const stage = ring + 18 * ldc;          // [2][9][18]: the IMU part of the raw coupling rows of the NEXT step's two blocks
    lds_d* const wd = AS_LDS(m.wd);                // (aliases the ring: used after the chain)
    const lds_d* g = AS_LDS_C(m.vec + V_G * L.Rpad);
    const lds_d* sc = AS_LDS_C(m.vec + V_SC * L.Rpad);
    const lds_d* tv = AS_LDS_C(m.vec + V_T * L.Rpad);
    const lds_i* pinv = (const lds_i*)m.pinv;
    const glb_d* buf = AS_GLB_C(buf_);
    const glb_d* imuJ = buf + L.bo_imuJ;
    const glb_d* Hp = AS_GLB_C(c.sc + L.so_Hp);
    glb_d* const xp = AS_GLB(m.xp);
    lds_i* flag = (lds_i*)(m.red + 24);
    const int wave = uni(c.wave);
    unsigned vm;                                   // bit f: IMU factor f is present
    {
        const glb_i* valid = AS_GLB_CI(c.ia + L.io_imu_valid);
        const int v = c.lane < K - 1 ? valid[c.lane] : 0;
        vm = (unsigned)__ballot(v != 0);
    }
    if (c.tid == 0) *flag = 1;
    const int nt = RcPad / 16;
    const int ntile = nt * (nt + 1) / 2;
    double4_t acc[SCHUR_TPW];
    int tm[SCHUR_TPW], tn[SCHUR_TPW];
#pragma unroll
    for (int s = 0; s < SCHUR_TPW; ++s) {
        acc[s] = (double4_t){0, 0, 0, 0};
        int a, bq;
        tri_decode(wave + s * SV_NW, a, bq);
        tm[s] = a; tn[s] = bq;
    }
    // tasks of a sweep: Rc + 1 columns of [C_k | g_k], 9 of the coupling block.  Wavefront 0 takes the first 64 tasks of the top
    // sweep, wavefront 1 those of the bottom sweep -- so that inside them the sweep (coupling-block strides, block index) is
    // wave-uniform -- and wavefront 2 what is left of both (the host checks that it fits); wavefront 3 factors the diagonal blocks.
    const int nth = Rc + 10, nfull = nth < 64 ? nth : 64, nrem = nth - nfull;
    const int half = wave < 2 ? wave : (c.lane >= nrem ? 1 : 0);
    const int id = wave < 2 ? c.lane : 64 + c.lane - half * nrem;
    const bool tasked = wave < 2 ? c.lane < nfull : (wave == 2 && c.lane < 2 * nrem);
    const bool col_task = tasked && id <= Rc;
    const bool cpl_task = tasked && id > Rc;
    const bool fac_wave = wave == SV_NW - 1;
    // ---- staging of the IMU part of a block's raw coupling rows: entry (r, c18), c18 = 6 (d + 1) + o for the pose column o of frame
    //      k + d, d = -1, 0, 1.  IMU factor f, local columns: 0-5 pose_f, 6-14 sb_f, 15-20 pose_f+1, 21-29 sb_f+1, packed lower
    //      triangle in imuJ[f][.]: factor k gives (sb_k, pose_k) and (pose_k+1, sb_k), factor k-1 gives (sb_k, pose_k-1) and (sb_k, pose_k)
    double sv[2][2];
    int si_a[2], si_b[2];                                    // this thread's two staged entries: local indices in factor k / factor k-1
    bool si_on[2], si_fa[2], si_fb[2], si_h[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const int e2 = c.tid + SV_NT * qq;
        const int hh = e2 >= 162 ? 1 : 0, e = e2 - 162 * hh;
        const int r = e / 18, c18 = e - 18 * r, d = c18 / 6 - 1, o = c18 - 6 * (d + 1);
        si_on[qq] = e2 < 324; si_h[qq] = hh != 0; si_fa[qq] = d >= 0; si_fb[qq] = d <= 0;
        si_a[qq] = d == 0 ? (6 + r) * (7 + r) / 2 + o : (15 + o) * (16 + o) / 2 + 6 + r;
        si_b[qq] = (21 + r) * (22 + r) / 2 + (d == 0 ? 15 + o : o);
    }
    auto stage_issue = [&](int kt_n, int kb_n) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int k = si_h[qq] ? kb_n : kt_n;
            const bool on = si_on[qq] && k >= 0;
            const bool fa_on = on && si_fa[qq] && k <= K - 2 && ((vm >> (k >= 0 ? k : 0)) & 1u);
            const bool fb_on = on && si_fb[qq] && k >= 1 && ((vm >> (k >= 1 ? k - 1 : 0)) & 1u);
            const double va = imuJ[fa_on ? k * 512 + si_a[qq] : 0], vb = imuJ[fb_on ? (k - 1) * 512 + si_b[qq] : 0];
            // even factor first, then the odd one: the order in which the former scatter rounds added them
            const double a = fa_on ? va : 0.0, b = fb_on ? vb : 0.0;
            sv[qq][0] = (k & 1) ? b : a;
            sv[qq][1] = (k & 1) ? a : b;
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int e2 = c.tid + SV_NT * qq;
            if (e2 < 324) stage[e2] = sv[qq][0] + sv[qq][1];
        }
    };
    // the raw column `id` of block k: staged IMU part (+ the prior's J0^T J0 row if the prior holds this speed-bias block), g for the rhs
    auto raw_col = [&](int k, double* c0) {
#pragma unroll
        for (int r = 0; r < 9; ++r) c0[r] = 0.0;
        if (k < 0 || !col_task) return;
        if (id == Rc) {
#pragma unroll
            for (int r = 0; r < 9; ++r) c0[r] = g[Rc + 9 * k + r];
            return;
        }
        const int p = id / 6, o = id - 6 * p, d = p - k;
        if (id < 6 * K && d >= -1 && d <= 1) {
            const lds_d* st = stage + 162 * half + 6 * (d + 1) + o;
#pragma unroll
            for (int r = 0; r < 9; ++r) c0[r] = st[18 * r];
        }
        const int pbk = pinv[Rc + 9 * k], pj = pinv[id];
        if (pbk >= 0 && pj >= 0) {
            double pv[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int pr_ = pbk + r;
                const int hi = pr_ > pj ? pr_ : pj, lo = pr_ > pj ? pj : pr_;
                pv[r] = Hp[hi * Ncap + lo];
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) c0[r] += pv[r];
        }
    };
    double xprev[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) xprev[r] = 0.0;
    double q = 0.0;
    {
        const bool l0 = nstep == 0;
        const int kt0 = l0 ? mid : K - 1, kb0 = l0 ? -1 : 0;
        stage_issue((l0 || kt0 > mid) ? kt0 : -1, (!l0 && kb0 < mid) ? kb0 : -1);
        stage_store();
    }
    __syncthreads();
    DP_DECL;
    for (int t = 0; t <= nstep; ++t) {
        const bool last = t == nstep;
        // blocks of this step: top kt (coupled downwards), bottom kb (coupled upwards); in the last step only `mid`
        const int kt = last ? mid : K - 1 - t, kb = last ? -1 : t;
        const bool has_t = last || kt > mid, has_b = !last && kb < mid;
        const bool upd_t_from_above = has_t && kt + 1 <= K - 1 && (last ? (K - 1 > mid) : t > 0);
        const bool upd_mid_from_below = last && mid > 0;
        const bool upd_b = has_b && t > 0;
        const int k = half == 0 ? kt : kb;
        const bool act = tasked && (half == 0 ? has_t : has_b);
        // next step's blocks (their raw rows are requested now, staged behind the first barrier of this step)
        int ktn = -1, kbn = -1;
        if (t < nstep) {
            const bool nlast = t + 1 == nstep;
            const int k2 = nlast ? mid : K - 2 - t;
            if (nlast || k2 > mid) ktn = k2;
            if (!nlast && t + 1 < mid) kbn = t + 1;
        }
        double x[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) x[r] = 0.0;
        stage_issue(ktn, kbn);
        DP_ADD(18);
        // ---- (A)
        if (fac_wave) {
            const int kind_t = (upd_t_from_above ? 1 : 0) | (upd_mid_from_below ? 2 : 0);
            double msk[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) msk[i] = (c.lane >> 4) == i ? 1.0 : 0.0;
            bool ok = true;
#define CF_STEP1(f) cf_pivot<0>(f, msk); cf_pivot<1>(f, msk); cf_pivot<2>(f, msk); cf_pivot<3>(f, msk); cf_pivot<4>(f, msk); \
                    cf_pivot<5>(f, msk); cf_pivot<6>(f, msk); cf_pivot<7>(f, msk); cf_pivot<8>(f, msk);
            if (has_t && has_b) {
                ChainFac ft, fb;
                cf_load<true>(ft, c.lane, D + 81 * kt, kind_t, (const lds_d*)(E + 81 * (kt + 1)), (const lds_d*)(E + 81 * kt));
                cf_load<true>(fb, c.lane, D + 81 * kb, upd_b ? 2 : 0, (const lds_d*)nullptr, (const lds_d*)(E + 81 * kb));
#define CF_STEP(r) cf_pivot<r>(ft, msk); cf_pivot<r>(fb, msk);
                CF_STEP(0) CF_STEP(1) CF_STEP(2) CF_STEP(3) CF_STEP(4) CF_STEP(5) CF_STEP(6) CF_STEP(7) CF_STEP(8)
#undef CF_STEP
                ok = cf_store(ft, c.lane, D + 81 * kt, dinv + 9 * kt);
                ok = cf_store(fb, c.lane, D + 81 * kb, dinv + 9 * kb) && ok;
            } else if (has_t) {
                ChainFac ft;
                cf_load<true>(ft, c.lane, D + 81 * kt, kind_t, (const lds_d*)(E + 81 * (kt + 1)), (const lds_d*)(E + 81 * kt));
                CF_STEP1(ft)
                ok = cf_store(ft, c.lane, D + 81 * kt, dinv + 9 * kt);
            } else if (has_b) {
                ChainFac fb;
                cf_load<true>(fb, c.lane, D + 81 * kb, upd_b ? 2 : 0, (const lds_d*)nullptr, (const lds_d*)(E + 81 * kb));
                CF_STEP1(fb)
                ok = cf_store(fb, c.lane, D + 81 * kb, dinv + 9 * kb);
            }
#undef CF_STEP1
            if (!ok && c.lane == 0) *flag = 0;
        }
        if (act && col_task) {
            double c0[9];
            raw_col(k, c0);
            const double scj = id < Rc ? sc[id] : 1.0, tvj = id < Rc ? tv[id] : 0.0;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int ca = Rc + 9 * k + r;
                const double v = c0[r] * sc[ca] * scj;
                q += 2.0 * v * tv[ca] * tvj;
                x[r] = v;
            }
            DP_ADD(19);
            // x -= (coupling block)^T x_prev;  coef[9 p + r] = coupling entry (row r of this block, row p of the neighbour's X): both
            // sweeps keep their coupling block as [p][r] -- one code path, only the block differs per lane in wavefront 2.
            // (An MFMA form of this update -- 15 MFMAs per block from the staged rows instead of 81 multiply-adds per column -- was
            //  measured and lost, 59K against 24K cycles per round: five dependent tile trips of index arithmetic and LDS reads on ONE
            //  wavefront are slower than 64 columns side by side; profiles/r05p_*.)
            if (half == 0 ? upd_t_from_above : upd_b) chain_upd_c<9, 1>(x, half == 0 ? E + 81 * (kt + 1) : E + 81 * kb, xprev);
            DP_ADD(20);
            if (half == 0 && upd_mid_from_below) {
                double xb[9];
#pragma unroll
                for (int p = 0; p < 9; ++p) xb[p] = ring[(9 + p) * ldc + id];      // X of block mid - 1: the bottom sweep's last rows
                chain_upd_c<9, 1>(x, E + 81 * kt, xb);
            }
        } else if (act && cpl_task && !(half == 0 && last)) {
            const int cc = id - Rc - 1;
            // top: column cc of E_kt -> Xe_kt[.][cc];  bottom: row cc of E_kb+1 -> XuT of the pair (kb + 1, kb)
            const lds_d* e = half == 0 ? E + 81 * kt + cc : E + 81 * (kb + 1) + 9 * cc;
            const int st = half == 0 ? 9 : 1;
#pragma unroll
            for (int r = 0; r < 9; ++r) x[r] = e[r * st];
        }
        DP_ADD(0);
        __syncthreads();
        DP_ADD(1);
        if (*flag == 0) break;
        // ---- (B) x <- L_k^-1 x
        if (act && (col_task || !(half == 0 && last))) {
            const lds_d* Lk = D + 81 * k;
            const lds_d* dk = dinv + 9 * k;

#pragma unroll
            for (int r = 0; r < 9; ++r) {
                double s = x[r];
#pragma unroll
                for (int qq = 0; qq < r; ++qq) s -= Lk[9 * r + qq] * x[qq];
                x[r] = s * dk[r];
            }
            if (col_task) {
                lds_d* ro = ring + (half ? 9 : 0) * ldc + id;
                glb_d* po = xp + (size_t)9 * k * ldc + id;
#pragma unroll
                for (int r = 0; r < 9; ++r) { ro[r * ldc] = x[r]; po[(size_t)r * ldc] = x[r]; xprev[r] = x[r]; }
            } else {
                // (all reads of the slot -- the cpl loads of (A) -- are behind the barrier: the bottom block's rows can come back
                //  transposed, as [p][c] like the top sweep's Xe)
                const int cc = id - Rc - 1;
                lds_d* e = (half == 0 ? E + 81 * kt : E + 81 * (kb + 1)) + cc;
#pragma unroll
                for (int r = 0; r < 9; ++r) e[9 * r] = x[r];
            }
        }
        stage_store();                             // (the staging area's readers -- raw_col in (A) -- are behind the barrier above)
        DP_ADD(2);
        __syncthreads();
        DP_ADD(3);
        // ---- (C) S accumulators += X^T X over the staged rows: k-step u covers rows 4u .. 4u+3 of [top 9 | bottom 9]
        {
            const int nks = has_b ? 5 : 3;
            const int kq = c.lane >> 4, jc = c.lane & 15;
#pragma unroll
            for (int s = 0; s < SCHUR_TPW; ++s) {
                if (wave + s * SV_NW < ntile) {
                    double av[5], bv[5];
#pragma unroll
                    for (int u = 0; u < 5; ++u) {
                        const int row = 4 * u + kq;
                        const bool on = u < nks && (row < 9 ? has_t : (row < 18 && has_b));
                        const lds_d* xr = ring + (on ? row : 0) * ldc + jc;
                        const double a = xr[tm[s] * 16], b = xr[tn[s] * 16];
                        av[u] = on ? a : 0.0;
                        bv[u] = on ? b : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 5; ++u)
                        if (u < nks) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc[s], 0, 0, 0);
                }
            }
        }
        DP_ADD(10);
    }
    __syncthreads();                               // ring consumed (it is the landmark tile from here on); *flag final
    // ---- landmark columns: S -= Wd Wd^T,  Wd[c][l] = sc[c] * Wt[c][l] * lsc[l],  Wd[Rc][l] = b[l] * lsc[l],
    //      lsc[l] = sl[l] / sqrt(sl^2 h + mu dgl^2); the Wt values of the next 32 landmarks are fetched into registers while the
    //      current tile is multiplied.  Element w = tid + SV_NT * i of the (RcPad x SCHUR_LW) tile -> row w / 32, landmark l0 + w % 32.
    {
        const glb_d* Wt = buf + L.bo_Wt;
        const glb_d* h = buf + L.bo_h;
        const glb_d* b = buf + L.bo_b;
        const glb_d* sl = AS_GLB_C(c.sc + L.so_sl);
        const glb_d* dgl = AS_GLB_C(c.sc + L.so_dg + L.Rpad);
        glb_d* lsc = AS_GLB(c.sc + L.so_lsc);
        for (int l = c.tid; l < c.nL; l += SV_NT) lsc[l] = sl[l] / sqrt(sl[l] * sl[l] * h[l] + mu * dgl[l] * dgl[l]);
        double pre[SCHUR_PF], pls[SCHUR_PF];
        const int nel = RcPad * SCHUR_LW;
        auto fetch = [&](int l0) {
#pragma unroll
            for (int i = 0; i < SCHUR_PF; ++i) {
                const int w = c.tid + SV_NT * i;
                const int row = w / SCHUR_LW, l = l0 + (w % SCHUR_LW);
                const bool in = w < nel && row <= Rc && l < c.nL;
                pre[i] = in ? (row < Rc ? Wt[(size_t)row * L.Lcap + l] : b[l]) : 0.0;
                pls[i] = in ? lsc[l] : 0.0;
            }
        };
        __syncthreads();                           // lsc of this launch visible to every wavefront
        if (c.nL > 0) fetch(0);
        for (int l0 = 0; l0 < c.nL; l0 += SCHUR_LW) {
            __syncthreads();                       // previous tile consumed
            DP_ADD(11);
#pragma unroll
            for (int i = 0; i < SCHUR_PF; ++i) {
                const int w = c.tid + SV_NT * i;
                if (w < nel) {
                    const int row = w / SCHUR_LW, kk = w % SCHUR_LW;
                    wd[row * SCHUR_LD + kk] = (row < Rc ? sc[row] : 1.0) * pre[i] * pls[i];
                }
            }
            DP_ADD(12);
            __syncthreads();
            DP_ADD(13);
            if (l0 + SCHUR_LW < c.nL) fetch(l0 + SCHUR_LW);
#pragma unroll
            for (int s = 0; s < SCHUR_TPW; ++s) {
                if (wave + s * SV_NW < ntile) {
                    // the sixteen operand reads of a tile first, then its eight MFMAs back to back
                    double av[SCHUR_LW / 4], bv[SCHUR_LW / 4];
#pragma unroll
                    for (int kk = 0; kk < SCHUR_LW / 4; ++kk) {
                        av[kk] = wd[(tm[s] * 16 + (c.lane & 15)) * SCHUR_LD + kk * 4 + (c.lane >> 4)];
                        bv[kk] = wd[(tn[s] * 16 + (c.lane & 15)) * SCHUR_LD + kk * 4 + (c.lane >> 4)];
                    }
#pragma unroll
                    for (int kk = 0; kk < SCHUR_LW / 4; ++kk) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc[s], 0, 0, 0);
                }
            }
            DP_ADD(14);
        }
        __syncthreads();
        DP_ADD(15);
    }
    // D[row = (lane>>4) + 4*reg][col = lane&15]
#pragma unroll
    for (int s = 0; s < SCHUR_TPW; ++s) {
        if (wave + s * SV_NW < ntile) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = tm[s] * 16 + (c.lane >> 4) + 4 * reg;
                const int col = tn[s] * 16 + (c.lane & 15);
                if (row <= Rc && col <= row && col < Rc) S[tri(row, col)] -= acc[s][reg];
            }
        }
    }
    __syncthreads();
    return q;
}

#if SV_NW >= 8
// ---- 8-wavefront build only (ba_solve_w8_kernel: one window per CU, the drop-in's operating point): the two halves of chain_schur SIDE BY
// SIDE.  The landmark part of the Schur complement (S -= Wd Wd^T, ~22K of the kernel's ~190K cycles per round) does not depend on the
// elimination of the speed-bias chain (~57K): wavefronts 0 .. 3 run the chain exactly as the 4-wavefront kernel does (same roles, same
// tile ownership, same sums), wavefronts 4 .. 7 run the landmark trips with the thread mapping of the 4-wavefront kernel, both inside ONE
// loop whose iteration is a chain step beside a landmark trip (two workgroup barriers each: the step's A | B | C phases line up with the
// trip's "previous tile consumed" | stage | multiply).  The landmark tile gets its own LDS behind the carve (BaLayout::l_wd8; the ring it
// aliases in chain_schur is live now).  S -= (chain accumulators) and S -= (landmark accumulators) are two passes: the same sums as
// chain_schur up to the order of these two subtractions.
#define CS_NW 4
#define CS_NT (64 * CS_NW)
#define CS_PF ((96 * SCHUR_LW + CS_NT - 1) / CS_NT)
template <int SCHUR_TPW>
NOINL double chain_schur_split(const Ctx& c_in, const SolveLds& m_in, const double* buf_, double mu) {
    PHASE_ENTER(false);
    mu = uni(mu);
    const int K = L.K, Rc = L.Rc, RcPad = L.RcPad, ldc = m.ldc, Ncap = L.Ncap;
    const int mid = K / 2, nstep = mid;            // (K - 1 - mid <= mid)
    lds_d* const S = AS_LDS(m.S);
    lds_d* const D = AS_LDS(m.D);
    lds_d* const E = AS_LDS(m.E);
    lds_d* const dinv = AS_LDS(m.dinv);
    lds_d* const ring = AS_LDS(m.ring);            // rows 0..8: top block of the step, rows 9..17: bottom block
    lds_d* const stage = ring + 18 * ldc;          // [2][9][18]: the IMU part of the raw coupling rows of the NEXT step's two blocks
    lds_d* const wd = AS_LDS(LDSB + L.l_wd8);      // (its own tile behind the carve: the ring is live beside it)
    const lds_d* g = AS_LDS_C(m.vec + V_G * L.Rpad);
    const lds_d* sc = AS_LDS_C(m.vec + V_SC * L.Rpad);
    const lds_d* tv = AS_LDS_C(m.vec + V_T * L.Rpad);
    const lds_i* pinv = (const lds_i*)m.pinv;
    const glb_d* buf = AS_GLB_C(buf_);
    const glb_d* imuJ = buf + L.bo_imuJ;
    const glb_d* Hp = AS_GLB_C(c.sc + L.so_Hp);
    glb_d* const xp = AS_GLB(m.xp);
    lds_i* flag = (lds_i*)(m.red + 24);
    const int wave = uni(c.wave);
    const bool lm_grp = wave >= CS_NW;             // wavefronts 0 .. 3: the chain; 4 .. 7: the landmark columns
    const int ltid = c.tid - CS_NT;                // thread of the landmark group
    unsigned vm;                                   // bit f: IMU factor f is present
    {
        const glb_i* valid = AS_GLB_CI(c.ia + L.io_imu_valid);
        const int v = c.lane < K - 1 ? valid[c.lane] : 0;
        vm = (unsigned)__ballot(v != 0);
    }
    if (c.tid == 0) *flag = 1;
    const int nt = RcPad / 16;
    const int ntile = nt * (nt + 1) / 2;
    double4_t acc[SCHUR_TPW];
    int tm[SCHUR_TPW], tn[SCHUR_TPW];
#pragma unroll
    for (int s = 0; s < SCHUR_TPW; ++s) {
        acc[s] = (double4_t){0, 0, 0, 0};
        int a, bq;
        tri_decode((wave & (CS_NW - 1)) + s * CS_NW, a, bq);
        tm[s] = a; tn[s] = bq;
    }
    // tasks of a sweep: Rc + 1 columns of [C_k | g_k], 9 of the coupling block.  Wavefront 0 takes the first 64 tasks of the top
    // sweep, wavefront 1 those of the bottom sweep -- so that inside them the sweep (coupling-block strides, block index) is
    // wave-uniform -- and wavefront 2 what is left of both (the host checks that it fits); wavefront 3 factors the diagonal blocks.
    const int nth = Rc + 10, nfull = nth < 64 ? nth : 64, nrem = nth - nfull;
    const int half = wave < 2 ? wave : (c.lane >= nrem ? 1 : 0);
    const int id = wave < 2 ? c.lane : 64 + c.lane - half * nrem;
    const bool tasked = wave < 2 ? c.lane < nfull : (wave == 2 && c.lane < 2 * nrem);
    const bool col_task = tasked && id <= Rc;
    const bool cpl_task = tasked && id > Rc;
    const bool fac_wave = wave == CS_NW - 1;
    // ---- staging of the IMU part of a block's raw coupling rows: entry (r, c18), c18 = 6 (d + 1) + o for the pose column o of frame
    //      k + d, d = -1, 0, 1.  IMU factor f, local columns: 0-5 pose_f, 6-14 sb_f, 15-20 pose_f+1, 21-29 sb_f+1, packed lower
    //      triangle in imuJ[f][.]: factor k gives (sb_k, pose_k) and (pose_k+1, sb_k), factor k-1 gives (sb_k, pose_k-1) and (sb_k, pose_k)
    double sv[2][2];
    int si_a[2], si_b[2];                                    // this thread's two staged entries: local indices in factor k / factor k-1
    bool si_on[2], si_fa[2], si_fb[2], si_h[2];
#pragma unroll
    for (int qq = 0; qq < 2; ++qq) {
        const int e2 = (lm_grp ? 324 : c.tid) + CS_NT * qq;      // (the landmark group stages nothing: e2 >= 324)
        const int hh = e2 >= 162 ? 1 : 0, e = e2 - 162 * hh;
        const int r = e / 18, c18 = e - 18 * r, d = c18 / 6 - 1, o = c18 - 6 * (d + 1);
        si_on[qq] = e2 < 324; si_h[qq] = hh != 0; si_fa[qq] = d >= 0; si_fb[qq] = d <= 0;
        si_a[qq] = d == 0 ? (6 + r) * (7 + r) / 2 + o : (15 + o) * (16 + o) / 2 + 6 + r;
        si_b[qq] = (21 + r) * (22 + r) / 2 + (d == 0 ? 15 + o : o);
    }
    auto stage_issue = [&](int kt_n, int kb_n) {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int k = si_h[qq] ? kb_n : kt_n;
            const bool on = si_on[qq] && k >= 0;
            const bool fa_on = on && si_fa[qq] && k <= K - 2 && ((vm >> (k >= 0 ? k : 0)) & 1u);
            const bool fb_on = on && si_fb[qq] && k >= 1 && ((vm >> (k >= 1 ? k - 1 : 0)) & 1u);
            const double va = imuJ[fa_on ? k * 512 + si_a[qq] : 0], vb = imuJ[fb_on ? (k - 1) * 512 + si_b[qq] : 0];
            // even factor first, then the odd one: the order in which the former scatter rounds added them
            const double a = fa_on ? va : 0.0, b = fb_on ? vb : 0.0;
            sv[qq][0] = (k & 1) ? b : a;
            sv[qq][1] = (k & 1) ? a : b;
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const int e2 = (lm_grp ? 324 : c.tid) + CS_NT * qq;
            if (e2 < 324) stage[e2] = sv[qq][0] + sv[qq][1];
        }
    };
    // the raw column `id` of block k: staged IMU part (+ the prior's J0^T J0 row if the prior holds this speed-bias block), g for the rhs
    auto raw_col = [&](int k, double* c0) {
#pragma unroll
        for (int r = 0; r < 9; ++r) c0[r] = 0.0;
        if (k < 0 || !col_task) return;
        if (id == Rc) {
#pragma unroll
            for (int r = 0; r < 9; ++r) c0[r] = g[Rc + 9 * k + r];
            return;
        }
        const int p = id / 6, o = id - 6 * p, d = p - k;
        if (id < 6 * K && d >= -1 && d <= 1) {
            const lds_d* st = stage + 162 * half + 6 * (d + 1) + o;
#pragma unroll
            for (int r = 0; r < 9; ++r) c0[r] = st[18 * r];
        }
        const int pbk = pinv[Rc + 9 * k], pj = pinv[id];
        if (pbk >= 0 && pj >= 0) {
            double pv[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int pr_ = pbk + r;
                const int hi = pr_ > pj ? pr_ : pj, lo = pr_ > pj ? pj : pr_;
                pv[r] = Hp[hi * Ncap + lo];
            }
#pragma unroll
            for (int r = 0; r < 9; ++r) c0[r] += pv[r];
        }
    };
    double xprev[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) xprev[r] = 0.0;
    double q = 0.0;
    {
        const bool l0 = nstep == 0;
        const int kt0 = l0 ? mid : K - 1, kb0 = l0 ? -1 : 0;
        stage_issue((l0 || kt0 > mid) ? kt0 : -1, (!l0 && kb0 < mid) ? kb0 : -1);
        stage_store();
    }
    // ---- landmark group: its inputs and the scaling of its landmarks (visible to its own wavefronts behind the first barrier)
    const glb_d* Wt = buf + L.bo_Wt;
    const glb_d* hl = buf + L.bo_h;
    const glb_d* bl = buf + L.bo_b;
    const glb_d* sl = AS_GLB_C(c.sc + L.so_sl);
    const glb_d* dgl = AS_GLB_C(c.sc + L.so_dg + L.Rpad);
    glb_d* lsc = AS_GLB(c.sc + L.so_lsc);
    if (lm_grp)
        for (int l = ltid; l < c.nL; l += CS_NT) lsc[l] = sl[l] / sqrt(sl[l] * sl[l] * hl[l] + mu * dgl[l] * dgl[l]);
    double pre[CS_PF], pl = 0.0;           // (element w = ltid + CS_NT i: its landmark l0 + w % 32 is the same for every i)
#pragma unroll
    for (int i = 0; i < CS_PF; ++i) pre[i] = 0.0;
    const int nel = RcPad * SCHUR_LW;
    auto fetch = [&](int l0) {
#pragma unroll
        for (int i = 0; i < CS_PF; ++i) {
            const int w = ltid + CS_NT * i;
            const int row = w / SCHUR_LW, l = l0 + (w % SCHUR_LW);
            const bool in = w < nel && row <= Rc && l < c.nL;
            pre[i] = in ? (row < Rc ? Wt[(size_t)row * L.Lcap + l] : bl[l]) : 0.0;
        }
        const int lme = l0 + ((ltid >= 0 ? ltid : 0) % SCHUR_LW);
        const double lv = lsc[lme < c.nL ? lme : 0];
        pl = lme < c.nL ? lv : 0.0;
    };
    // (C) of a chain step is shared by ALL eight wavefronts (tile wave + 8 s): accumulators of their own beside the landmark group's
    constexpr int CS_TPC = (SCHUR_TPW * CS_NW + SV_NW - 1) / SV_NW;
    double4_t accC[CS_TPC];
    int tmc[CS_TPC], tnc[CS_TPC];
#pragma unroll
    for (int s = 0; s < CS_TPC; ++s) {
        accC[s] = (double4_t){0, 0, 0, 0};
        int a, bq;
        tri_decode(wave + s * SV_NW, a, bq);
        tmc[s] = a; tnc[s] = bq;
    }
    auto c_tiles = [&](bool has_t, bool has_b) {
        // k-step u covers rows 4u .. 4u+3 of [top 9 | bottom 9] of the staged rows
        const int nks = has_b ? 5 : 3;
        const int kq = c.lane >> 4, jc = c.lane & 15;
#pragma unroll
        for (int s = 0; s < CS_TPC; ++s) {
            if (wave + s * SV_NW < ntile) {
                double av[5], bv[5];
#pragma unroll
                for (int u = 0; u < 5; ++u) {
                    const int row = 4 * u + kq;
                    const bool on = u < nks && (row < 9 ? has_t : (row < 18 && has_b));
                    const lds_d* xr = ring + (on ? row : 0) * ldc + jc;
                    const double a = xr[tmc[s] * 16], b = xr[tnc[s] * 16];
                    av[u] = on ? a : 0.0;
                    bv[u] = on ? b : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 5; ++u)
                    if (u < nks) accC[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], accC[s], 0, 0, 0);
            }
        }
    };
    __syncthreads();                               // chain: the first step's staged rows; landmark group: lsc
    if (lm_grp && c.nL > 0) fetch(0);
    DP_DECL;
    const int ntrip = (c.nL + SCHUR_LW - 1) / SCHUR_LW;
    const int niter = nstep + 1 > ntrip ? nstep + 1 : ntrip;
    // Two loops with the SAME number of workgroup barriers (s_barrier counts wavefronts, not program locations) and the same exit test:
    // written apart, the registers of a chain step (solved columns, staged entries) and those of a landmark trip (the prefetched tile)
    // do not add up in one live range -- in one loop body the function spilled 133 registers.
    if (!lm_grp) {
    for (int t = 0; t < niter; ++t) {
        const bool ch_on = t <= nstep;
        const bool last = t >= nstep;
        // blocks of this step: top kt (coupled downwards), bottom kb (coupled upwards); in the last step only `mid`
        const int kt = last ? mid : K - 1 - t, kb = last ? -1 : t;
        const bool has_t = last || kt > mid, has_b = !last && kb < mid;
        const bool upd_t_from_above = has_t && kt + 1 <= K - 1 && (last ? (K - 1 > mid) : t > 0);
        const bool upd_mid_from_below = last && mid > 0;
        const bool upd_b = has_b && t > 0;
        const int k = half == 0 ? kt : kb;
        const bool act = tasked && (half == 0 ? has_t : has_b);
        // next step's blocks (their raw rows are requested now, staged behind the first barrier of this step)
        int ktn = -1, kbn = -1;
        if (t < nstep) {
            const bool nlast = t + 1 == nstep;
            const int k2 = nlast ? mid : K - 2 - t;
            if (nlast || k2 > mid) ktn = k2;
            if (!nlast && t + 1 < mid) kbn = t + 1;
        }
        double x[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) x[r] = 0.0;
        if (ch_on) stage_issue(ktn, kbn);
        DP_ADD(18);
        // ---- (A)
        if (ch_on && fac_wave) {
            const int kind_t = (upd_t_from_above ? 1 : 0) | (upd_mid_from_below ? 2 : 0);
            double msk[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) msk[i] = (c.lane >> 4) == i ? 1.0 : 0.0;
            bool ok = true;
#define CF_STEP1(f) cf_pivot<0>(f, msk); cf_pivot<1>(f, msk); cf_pivot<2>(f, msk); cf_pivot<3>(f, msk); cf_pivot<4>(f, msk); \
                    cf_pivot<5>(f, msk); cf_pivot<6>(f, msk); cf_pivot<7>(f, msk); cf_pivot<8>(f, msk);
            if (has_t && has_b) {
                ChainFac ft, fb;
                cf_load<true>(ft, c.lane, D + 81 * kt, kind_t, (const lds_d*)(E + 81 * (kt + 1)), (const lds_d*)(E + 81 * kt));
                cf_load<true>(fb, c.lane, D + 81 * kb, upd_b ? 2 : 0, (const lds_d*)nullptr, (const lds_d*)(E + 81 * kb));
#define CF_STEP(r) cf_pivot<r>(ft, msk); cf_pivot<r>(fb, msk);
                CF_STEP(0) CF_STEP(1) CF_STEP(2) CF_STEP(3) CF_STEP(4) CF_STEP(5) CF_STEP(6) CF_STEP(7) CF_STEP(8)
#undef CF_STEP
                ok = cf_store(ft, c.lane, D + 81 * kt, dinv + 9 * kt);
                ok = cf_store(fb, c.lane, D + 81 * kb, dinv + 9 * kb) && ok;
            } else if (has_t) {
                ChainFac ft;
                cf_load<true>(ft, c.lane, D + 81 * kt, kind_t, (const lds_d*)(E + 81 * (kt + 1)), (const lds_d*)(E + 81 * kt));
                CF_STEP1(ft)
                ok = cf_store(ft, c.lane, D + 81 * kt, dinv + 9 * kt);
            } else if (has_b) {
                ChainFac fb;
                cf_load<true>(fb, c.lane, D + 81 * kb, upd_b ? 2 : 0, (const lds_d*)nullptr, (const lds_d*)(E + 81 * kb));
                CF_STEP1(fb)
                ok = cf_store(fb, c.lane, D + 81 * kb, dinv + 9 * kb);
            }
#undef CF_STEP1
            if (!ok && c.lane == 0) *flag = 0;
        }
        if (ch_on && act && col_task) {
            double c0[9];
            raw_col(k, c0);
            const double scj = id < Rc ? sc[id] : 1.0, tvj = id < Rc ? tv[id] : 0.0;
#pragma unroll
            for (int r = 0; r < 9; ++r) {
                const int ca = Rc + 9 * k + r;
                const double v = c0[r] * sc[ca] * scj;
                q += 2.0 * v * tv[ca] * tvj;
                x[r] = v;
            }
            DP_ADD(19);
            // x -= (coupling block)^T x_prev;  coef[9 p + r] = coupling entry (row r of this block, row p of the neighbour's X): both
            // sweeps keep their coupling block as [p][r] -- one code path, only the block differs per lane in wavefront 2.
            // (An MFMA form of this update -- 15 MFMAs per block from the staged rows instead of 81 multiply-adds per column -- was
            //  measured and lost, 59K against 24K cycles per round: five dependent tile trips of index arithmetic and LDS reads on ONE
            //  wavefront are slower than 64 columns side by side; profiles/r05p_*.)
            if (half == 0 ? upd_t_from_above : upd_b) chain_upd_c<9, 1>(x, half == 0 ? E + 81 * (kt + 1) : E + 81 * kb, xprev);
            DP_ADD(20);
            if (half == 0 && upd_mid_from_below) {
                double xb[9];
#pragma unroll
                for (int p = 0; p < 9; ++p) xb[p] = ring[(9 + p) * ldc + id];      // X of block mid - 1: the bottom sweep's last rows
                chain_upd_c<9, 1>(x, E + 81 * kt, xb);
            }
        } else if (ch_on && act && cpl_task && !(half == 0 && last)) {
            const int cc = id - Rc - 1;
            // top: column cc of E_kt -> Xe_kt[.][cc];  bottom: row cc of E_kb+1 -> XuT of the pair (kb + 1, kb)
            const lds_d* e = half == 0 ? E + 81 * kt + cc : E + 81 * (kb + 1) + 9 * cc;
            const int st = half == 0 ? 9 : 1;
#pragma unroll
            for (int r = 0; r < 9; ++r) x[r] = e[r * st];
        }
        DP_ADD(0);
        __syncthreads();                           // chain: A done | landmark group: the previous tile is consumed
        if (*flag == 0) break;                     // (every thread of the workgroup reads the same value)
        DP_ADD(1);
        // ---- (B) x <- L_k^-1 x
        if (ch_on && act && (col_task || !(half == 0 && last))) {
            const lds_d* Lk = D + 81 * k;
            const lds_d* dk = dinv + 9 * k;

#pragma unroll
            for (int r = 0; r < 9; ++r) {
                double s = x[r];
#pragma unroll
                for (int qq = 0; qq < r; ++qq) s -= Lk[9 * r + qq] * x[qq];
                x[r] = s * dk[r];
            }
            if (col_task) {
                lds_d* ro = ring + (half ? 9 : 0) * ldc + id;
                glb_d* po = xp + (size_t)9 * k * ldc + id;
#pragma unroll
                for (int r = 0; r < 9; ++r) { ro[r * ldc] = x[r]; po[(size_t)r * ldc] = x[r]; xprev[r] = x[r]; }
            } else {
                // (all reads of the slot -- the cpl loads of (A) -- are behind the barrier: the bottom block's rows can come back
                //  transposed, as [p][c] like the top sweep's Xe)
                const int cc = id - Rc - 1;
                lds_d* e = (half == 0 ? E + 81 * kt : E + 81 * (kb + 1)) + cc;
#pragma unroll
                for (int r = 0; r < 9; ++r) e[9 * r] = x[r];
            }
        }
        if (ch_on) stage_store();                             // (the staging area's readers -- raw_col in (A) -- are behind the barrier above)
        DP_ADD(2);
        __syncthreads();                           // chain: the solved rows are staged | landmark group: the tile is staged
        DP_ADD(3);
        if (ch_on) c_tiles(has_t, has_b);           // (C) this wavefront's share of S accumulators += X^T X over the staged rows
        DP_ADD(10);
    }
    } else {
    for (int t = 0; t < niter; ++t) {
        const bool lm_on = t < ntrip;
        const int l0 = t * SCHUR_LW;
        __syncthreads();                           // (chain: A done) the previous tile is consumed
        if (*flag == 0) break;
        if (lm_on) {
#pragma unroll
            for (int i = 0; i < CS_PF; ++i) {
                const int w = ltid + CS_NT * i;
                if (w < nel) {
                    const int row = w / SCHUR_LW, kk = w % SCHUR_LW;
                    wd[row * SCHUR_LD + kk] = (row < Rc ? sc[row] : 1.0) * pre[i] * pl;
                }
            }
        }
        __syncthreads();                           // the tile is staged (chain: the solved rows of step t are staged)
        if (t <= nstep) {
            const bool last = t >= nstep;
            const int kt = last ? mid : K - 1 - t, kb = last ? -1 : t;
            c_tiles(last || kt > mid, !last && kb < mid);
        }
        if (lm_on) {
            if (l0 + SCHUR_LW < c.nL) fetch(l0 + SCHUR_LW);
#pragma unroll
            for (int s = 0; s < SCHUR_TPW; ++s) {
                if ((wave & (CS_NW - 1)) + s * CS_NW < ntile) {
                    double av[SCHUR_LW / 4], bv[SCHUR_LW / 4];
#pragma unroll
                    for (int kk = 0; kk < SCHUR_LW / 4; ++kk) {
                        av[kk] = wd[(tm[s] * 16 + (c.lane & 15)) * SCHUR_LD + kk * 4 + (c.lane >> 4)];
                        bv[kk] = wd[(tn[s] * 16 + (c.lane & 15)) * SCHUR_LD + kk * 4 + (c.lane >> 4)];
                    }
#pragma unroll
                    for (int kk = 0; kk < SCHUR_LW / 4; ++kk) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc[s], 0, 0, 0);
                }
            }
        }
    }
    }
    __syncthreads();                               // *flag final; every accumulator complete
    // D[row = (lane>>4) + 4*reg][col = lane&15]: the chain rows' share (every wavefront, tiles wave + 8 s), then the landmark group's
#pragma unroll
    for (int s = 0; s < CS_TPC; ++s) {
        if (wave + s * SV_NW < ntile) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = tmc[s] * 16 + (c.lane >> 4) + 4 * reg;
                const int col = tnc[s] * 16 + (c.lane & 15);
                if (row <= Rc && col <= row && col < Rc) S[tri(row, col)] -= accC[s][reg];
            }
        }
    }
    __syncthreads();
    if (lm_grp) {
#pragma unroll
        for (int s = 0; s < SCHUR_TPW; ++s) {
            if ((wave & (CS_NW - 1)) + s * CS_NW < ntile) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int row = tm[s] * 16 + (c.lane >> 4) + 4 * reg;
                    const int col = tn[s] * 16 + (c.lane & 15);
                    if (row <= Rc && col <= row && col < Rc) S[tri(row, col)] -= acc[s][reg];
                }
            }
        }
    }
    __syncthreads();
    return q;
}
#endif

// Landmark part of  t^T H~ t  (the Cauchy-point denominator, see build_scaled):  sum_l [ h~_l t_l^2 + 2 t_l (w~_l . t_cam) ]
// with t = gt / Dg.  Only needed when the Gauss-Newton step leaves the trust region, so it is evaluated on demand
// (thread per landmark, one pass over Wt).  Returns this thread's share.
NOINL double cauchy_landmark_term(const Ctx& c_in, const SolveLds& m_in, const double* buf) {
    PHASE_ENTER(false);
    const double* sc = m.vec + V_SC * L.Rpad;
    const double* gt = m.vec + V_GT * L.Rpad;
    const double* dg = m.vec + V_DG * L.Rpad;
    double* tv = m.vec + V_T * L.Rpad;
    const double* Wt = buf + L.bo_Wt;
    const double* h = buf + L.bo_h;
    const double* sl = c.sc + L.so_sl;
    const double* dgl = c.sc + L.so_dg + L.Rpad;
    const double* gtl = c.sc + L.so_gt + L.Rpad;
    __syncthreads();
    for (int k = c.tid; k < L.Rc; k += SV_NT) tv[k] = sc[k] * (gt[k] / dg[k]);       // sc .* t on the camera columns
    __syncthreads();
    double q = 0.0;
    for (int l = c.tid; l < c.nL; l += SV_NT) {
        double wdot = 0.0;
        for (int row = 0; row < L.Rc; ++row) wdot += Wt[(size_t)row * L.Lcap + l] * tv[row];
        const double tl = gtl[l] / dgl[l];
        q += sl[l] * sl[l] * h[l] * tl * tl + 2.0 * tl * sl[l] * wdot;
    }
    return q;
}

// Blocked right-looking Cholesky (NB = 16) of the packed lower triangle S (R x R) held in LDS, with the rhs as
// augmented row R (so row R of the factor is the forward-substituted rhs):
//   (1+2) diagonal block AND panel in one phase, without a barrier in between: every wavefront holds the (symmetric) 16x16
//       diagonal block and ONE 16-row tile of the panel, transposed, in the accumulator layout of v_mfma_f64_16x16x4_f64
//       (D[i = (lane>>4) + 4 reg][j = lane & 15]).  Row r of an accumulator sits in ONE register (r >> 2) of the 16 lanes
//       16 (r & 3) .. +15 — exactly the lanes of k-slot r & 3 of the A and B operands — so pivot r is: read the pivot with
//       v_readlane, scale the row in place (that IS column r of L, for the diagonal block and for the tile), and apply the
//       rank-1 update with one MFMA per accumulator whose other three k-slots are zero.  No cross-lane traffic besides the
//       pivot, no LDS in the dependency chain (the L values are written out on the side).  The diagonal block is factored
//       redundantly by every wavefront that owns a tile; a short block is padded with the identity.
//   (3) the trailing matrix gets its rank-16 update tile by tile on v_mfma_f64_16x16x4_f64.
// Three barriers per 16 columns.  1/L_jj goes to V_DI.  Returns false (uniform) on a bad pivot.
NOINL bool cholesky_aug(const Ctx& c_in, const SolveLds& m_in, int R) {
    PHASE_ENTER(-1);
    lds_d* S = AS_LDS(m.S);
    lds_d* dinvv = AS_LDS(m.di);
    lds_i* flag = (lds_i*)(m.red + 24);
    const int lane = c.lane;
    R = __builtin_amdgcn_readfirstlane(R);              // arguments arrive in vector registers: tell the compiler they are uniform
    const int wave = __builtin_amdgcn_readfirstlane(c.wave);
    const int nwv = L.big ? BA_NW : SV_NW;             // wavefronts of the calling kernel (uniform)
    if (c.tid == 0) *flag = 1;
    __syncthreads();
    DP_DECL;
    double msk[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) msk[k] = (lane >> 4) == k ? 1.0 : 0.0;
    for (int c0 = 0; c0 < R; c0 += 16) {
        const int nb = (R - c0) < 16 ? (R - c0) : 16;
        const int r1 = c0 + nb;
        const int ntp = (R - r1) / 16 + 1;             // panel tiles: rows r1 .. R
        const int jc = lane & 15, kq = lane >> 4;
        double4_t dg0;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int i = kq + 4 * reg;
            const bool in = i < nb && jc < nb;
            const int gi = c0 + (i > jc ? i : jc), gj = c0 + (i > jc ? jc : i);
            const double dv = S[in ? tri(gi, gj) : 0];
            dg0[reg] = in ? dv : (i == jc ? 1.0 : 0.0);
        }
        __syncthreads();                               // every wavefront has the block before wavefront 0 overwrites it with L
        for (int t = wave; t < ntp; t += nwv) {
            const int prow = r1 + 16 * t + jc;         // this lane's panel row (column of the transposed tile)
            double4_t dg = dg0, pt;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int i = kq + 4 * reg;
                const bool pin = i < nb && prow <= R;
                const double pv = S[pin ? tri(prow, c0 + i) : 0];
                pt[reg] = pin ? pv : 0.0;
            }
            // A lone wavefront issues one instruction every ~6-8 cycles whatever the dependencies (profiles/ubench), so the pivot
            // step is written for instruction COUNT: dm = 1/sqrt(pivot) in the lanes of k-slot r & 3 and 0 elsewhere scales the
            // row into column r of L (lv for the block, lp for the tile) with one multiply each; lane (kq, jc) collects the four
            // columns r = kq + 4 q it owns by plain additions (the other lanes add 0) and writes them after the last pivot; a
            // non-positive pivot is not tested here — it turns 1/sqrt and everything after it into NaN / Inf, caught below.
            // (Entries j < r of row r are rounding residue of the eliminated columns; they only reach dead rows / columns.)
            double4_t ld = {0, 0, 0, 0}, lq = {0, 0, 0, 0}, di = {0, 0, 0, 0};
#define CHOL_PIVOT(r)                                                                                         \
            {                                                                                                     \
                const double piv = readlane_d(dg[(r) >> 2], 16 * ((r) & 3) + (r));                                \
                const double dm = rsqrt_nr(piv) * msk[(r) & 3];                                                   \
                const double lv = dg[(r) >> 2] * dm, nlv = dg[(r) >> 2] * -dm, nlp = pt[(r) >> 2] * -dm;          \
                dg = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, nlv, dg, 0, 0, 0);                                  \
                pt = __builtin_amdgcn_mfma_f64_16x16x4f64(lv, nlp, pt, 0, 0, 0);                                  \
                ld[(r) >> 2] += lv;                                                                               \
                lq[(r) >> 2] -= nlp;                                                                              \
                di[(r) >> 2] += dm;                                                                               \
            }
            if (nb == 16) {                            // straight-line: no control flow between the sixteen pivots
#pragma unroll
                for (int r = 0; r < 16; ++r) CHOL_PIVOT(r)
            } else {                                   // the last, short block (padded with the identity): groups of four pivots
#pragma unroll
                for (int r4 = 0; r4 < 16; r4 += 4) {
                    if (r4 < nb) {
#pragma unroll
                        for (int r = r4; r < r4 + 4; ++r) CHOL_PIVOT(r)
                    }
                }
            }
#undef CHOL_PIVOT
            bool good = true;
            lds_d* const outd = S + tri(c0 + jc, c0) + kq;                  // + 4 q: L[c0 + jc][c0 + kq + 4 q]
            lds_d* const outp = S + tri(prow <= R ? prow : R, c0) + kq;     // + 4 q: L[prow][c0 + kq + 4 q]
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = kq + 4 * q;
                if (r < nb) {
                    if (!(di[q] > 0.0) || !(di[q] < 1e300)) good = false;   // pivot <= 0, NaN or Inf
                    if (t == 0 && jc < nb && jc >= r) outd[4 * q] = ld[q];
                    if (prow <= R) outp[4 * q] = lq[q];
                    if (t == 0 && jc == 0) dinvv[c0 + r] = di[q];
                }
            }
            if (!good) *flag = 0;
        }
        DP_ADD(4);
        DP_ADD(5);
        DP_ADD(6);
        __syncthreads();
        DP_ADD(7);
        if (*flag == 0) break;
        // ---- (3) trailing update (only full blocks have anything right of them)
        if (nb == 16 && r1 < R) {
            const int t0 = r1 >> 4;
            const int nt = (R >> 4) + 1;                 // tile rows covering rows 0..R
            const int mm = nt - t0;
            const int ntile = mm * (mm + 1) / 2;
            for (int t = wave; t < ntile; t += nwv) {
                int tr_, tc_;
                tri_decode(t, tr_, tc_);
                const int ti = t0 + tr_, tk = t0 + tc_;
                const int arow = 16 * ti + (lane & 15), brow = 16 * tk + (lane & 15);
                const bool interior = (16 * ti + 15 <= R) && (ti != tk);     // wave-uniform
                const lds_d* pa = S + tri(arow <= R ? arow : R, c0) + (lane >> 4);
                const lds_d* pb = S + tri(brow <= R ? brow : R, c0) + (lane >> 4);
                const double a0 = pa[0], a1 = pa[4], a2 = pa[8], a3 = pa[12];
                const double b0 = pb[0], b1 = pb[4], b2 = pb[8], b3 = pb[12];
                const bool av = arow <= R, bv = brow < R;
                double4_t acc = {0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av ? a0 : 0.0, bv ? b0 : 0.0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av ? a1 : 0.0, bv ? b1 : 0.0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av ? a2 : 0.0, bv ? b2 : 0.0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av ? a3 : 0.0, bv ? b3 : 0.0, acc, 0, 0, 0);
                const int colw = 16 * tk + (lane & 15);
                const int row0 = 16 * ti + (lane >> 4);
                if (interior) {
                    lds_d* q0 = S + tri(row0, colw);
                    lds_d* q1 = S + tri(row0 + 4, colw);
                    lds_d* q2 = S + tri(row0 + 8, colw);
                    lds_d* q3 = S + tri(row0 + 12, colw);
                    const double c0v = *q0, c1v = *q1, c2v = *q2, c3v = *q3;
                    *q0 = c0v - acc[0]; *q1 = c1v - acc[1]; *q2 = c2v - acc[2]; *q3 = c3v - acc[3];
                } else {
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int row = row0 + 4 * reg;
                        if (row <= R && colw < R && colw <= row) S[tri(row, colw)] -= acc[reg];
                    }
                }
            }
        }
        DP_ADD(8);
        __syncthreads();
        DP_ADD(9);
    }
    const bool ok = (*flag != 0);
    __syncthreads();
    return ok;
}

// y_cam <- solve L^T y = (row R of S) by one wavefront (lane owns entries lane and lane+64 of the running rhs in
// registers; the pivot value travels through v_readlane); result in V_Y[0..R).  R <= 128.
NOINL void back_substitute(const Ctx& c_in, const SolveLds& m_in, int R) {
    PHASE_ENTER(false);
    const lds_d* S = AS_LDS_C(m.S);
    const lds_d* dinvv = AS_LDS_C(m.di);
    lds_d* y = AS_LDS(m.vec + V_Y * L.Rpad);
    __syncthreads();
    if (c.wave == 0) {
        const int l0 = c.lane, l1 = c.lane + 64;
        const lds_d* rowR = S + tri(R, 0);
        double v0 = rowR[l0 < R ? l0 : 0], v1 = rowR[l1 < R ? l1 : 0];
        v0 = l0 < R ? v0 : 0.0; v1 = l1 < R ? v1 : 0.0;
        // Four pivots per trip: the rows of L and the 1/L_jj of all four are read first (they do not depend on the running
        // rhs), so that one LDS round trip is paid per four dependent steps instead of per step.
        int j = R - 1;
        for (; j >= 64 + 3; j -= 4) {   // pivot in v1
            double r0[4], r1[4], dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const lds_d* rj = S + tri(j - u, 0);
                r0[u] = rj[l0];
                r1[u] = rj[l1 < j - u ? l1 : 0];
                dv[u] = dinvv[j - u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int own = (j - u) & 63;
                const double xj = readlane_d(v1, own) * dv[u];
                v1 = c.lane == own ? xj : (l1 < j - u ? v1 - r1[u] * xj : v1);
                v0 -= r0[u] * xj;
            }
        }
        for (; j >= 64; --j) {
            const lds_d* rj = S + tri(j, 0);
            const double r0 = rj[l0];
            double r1 = rj[l1 < j ? l1 : 0];
            r1 = l1 < j ? r1 : 0.0;
            const int own = j & 63;
            const double xj = readlane_d(v1, own) * dinvv[j];
            v1 = c.lane == own ? xj : v1 - r1 * xj;
            v0 -= r0 * xj;
        }
        for (; j >= 3; j -= 4) {        // pivot in v0
            double r0[4], dv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                r0[u] = S[tri(j - u, 0) + (l0 < j - u ? l0 : 0)];
                dv[u] = dinvv[j - u];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double xj = readlane_d(v0, j - u) * dv[u];
                v0 = c.lane == j - u ? xj : (l0 < j - u ? v0 - r0[u] * xj : v0);
            }
        }
        for (; j >= 0; --j) {
            const lds_d* rj = S + tri(j, 0);
            double r0 = rj[l0 < j ? l0 : 0];
            r0 = l0 < j ? r0 : 0.0;
            const double xj = readlane_d(v0, j) * dinvv[j];
            v0 = c.lane == j ? xj : v0 - r0 * xj;
        }
        if (c.lane < R) y[c.lane] = v0;
        if (c.lane + 64 < R) y[c.lane + 64] = v1;
    }
    __syncthreads();
}

// speed-bias blocks, from the middle block outwards (two wavefronts, one per direction):
//   y_mid = L^-T z_mid;   top side  y_k = L_k^-T (z_k - Xe_k y_{k-1});   bottom side  y_k = L_k^-T (z_k - Xu_k y_{k+1})
// with z = xg - Xc y_cam for all 9K rows first.
template <typename PL, typename PD>
DEV double chain_backsolve9(PL Lk, PD dinvk, double v, int r) {
    // L^T y = v: backward over rows j = 8 .. 0; lane r accumulates its rhs entry and ends up holding y[r]
#pragma unroll
    for (int j = 8; j >= 0; --j) {
        const double yj = readlane_d(v, j) * dinvk[j];
        const double lrj = (r < j) ? Lk[9 * j + r] : 0.0;      // L[j][r], r < j
        v = (r == j) ? yj : v - lrj * yj;
    }
    return v;
}
template <bool BIG>
NOINL void chain_back_substitute(const Ctx& c_in, const SolveLds& m_in) {
    PHASE_ENTER(BIG);
    const SolveLds& m_ = m;
    typedef typename MovT<BIG>::D MV;
    const int K = L.K, Rc = L.Rc, ldc = m_.ldc;
    const int mid = K / 2;
    MV* const D = (MV*)m_.D;
    MV* const E = (MV*)m_.E;
    MV* const dinv = (MV*)m_.dinv;
    MV* const wd = (MV*)m_.wd;
    MV* const z = (MV*)m_.z;
    MV* y = (MV*)(m_.vec + V_Y * L.Rpad);
    {
        // (single-workgroup path: the rows X_k were parked in HBM by chain_schur -- every thread's loads are independent)
        constexpr int NT = BIG ? BA_NT : SV_NT;
        const glb_d* XR = BIG ? AS_GLB_C(m_.XC) : AS_GLB_C(m_.xp);
        const int nrow = 9 * K;
        // (round 6: sixteen entries of a row requested together from clamped addresses -- the loop over j waited for every entry of the
        //  parked row before it asked for the next one; entries beyond Rc enter as exact zeros: the same sums)
        for (int w = c.tid; w < 4 * nrow; w += NT) {
            const int row = w >> 2, q = w & 3;
            const glb_d* xr = XR + (size_t)row * ldc;
            double s = 0.0;
            for (int j0 = q; j0 < Rc; j0 += 64) {
                double xv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int j = j0 + 4 * u; xv[u] = xr[j < Rc ? j : q]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) { const int j = j0 + 4 * u; s += (j < Rc ? xv[u] : 0.0) * y[j < Rc ? j : q]; }
            }
            z[w] = s;
        }
        __syncthreads();
        for (int row = c.tid; row < nrow; row += NT)
            wd[row] = XR[(size_t)row * ldc + Rc] - ((z[4 * row] + z[4 * row + 1]) + (z[4 * row + 2] + z[4 * row + 3]));
        __syncthreads();
    }
    if (c.wave < 2) {
        const int r = c.lane < 9 ? c.lane : 8;
        // both wavefronts solve the middle block (cheaper than a hand-over through LDS)
        double yprev = chain_backsolve9(D + 81 * mid, dinv + 9 * mid, wd[9 * mid + r], r);
        if (c.wave == 0 && c.lane < 9) y[Rc + 9 * mid + c.lane] = yprev;
        if (c.wave == 0) {
            for (int k = mid + 1; k < K; ++k) {
                const MV* Xe = E + 81 * k;                  // [p][c]
                double v = wd[9 * k + r];
#pragma unroll
                for (int cc = 0; cc < 9; ++cc) v -= Xe[9 * r + cc] * readlane_d(yprev, cc);
                yprev = chain_backsolve9(D + 81 * k, dinv + 9 * k, v, r);
                if (c.lane < 9) y[Rc + 9 * k + c.lane] = yprev;
            }
        } else {
            for (int k = mid - 1; k >= 0; --k) {
                const MV* Xu = E + 81 * (k + 1);            // large-window path: [c][p]; single-workgroup path (chain_schur): [p][c]
                double v = wd[9 * k + r];
#pragma unroll
                for (int cc = 0; cc < 9; ++cc) v -= Xu[BIG ? 9 * cc + r : 9 * r + cc] * readlane_d(yprev, cc);
                yprev = chain_backsolve9(D + 81 * k, dinv + 9 * k, v, r);
                if (c.lane < 9) y[Rc + 9 * k + c.lane] = yprev;
            }
        }
    }
    __syncthreads();
}

DEV void R2ypr_dev(const double* R, double* ypr) {
    const double n0 = R[0], n1 = R[3], n2 = R[6];
    const double o0 = R[1], o1 = R[4];
    const double a0 = R[2], a1 = R[5];
    const double y = atan2(n1, n0);
    const double p = atan2(-n2, n0 * cos(y) + n1 * sin(y));
    const double r = atan2(a0 * sin(y) - a1 * cos(y), -o0 * sin(y) + o1 * cos(y));
    ypr[0] = y / M_PI * 180.0; ypr[1] = p / M_PI * 180.0; ypr[2] = r / M_PI * 180.0;
}

DEV double state_sqnorm_share(const BaLayout& L, const double* x, const double* lam, int nL, int tid, int nt) {
    double s = 0.0;
    const int nx = 7 * L.Kp + 9 * L.K + (L.e ? 7 : 0);
    for (int k = tid; k < nx; k += nt) s += x[k] * x[k];
    if (L.t && tid == 0) s += x[7 * L.Kp + 9 * L.K + 7] * x[7 * L.Kp + 9 * L.K + 7];
    for (int k = tid; k < nL; k += nt) s += lam[k] * lam[k];
    return s;
}

__device__ __forceinline__ void solve_body(const BaLayout* __restrict__ Lp, const BaPtrs& P) {
    const BaLayout L = layout_load(Lp);
    Ctx c;
    const int w = blockIdx.x;
    ctx_init(c, Lp, P, w);
    PHASE_SELF_CHECK(c);
    double* ctlp = c.sc + L.so_ctl;
    // Everything the first phase reads from HBM is requested in one batch (control block, cost partials of the linearisation
    // kernels).
    const double cs_part = sum_partials(AS_GLB_C(c.sc + L.so_part), L.nbl);
    Ctl s;
    ctl_load(s, ctlp);
    if (s.done) return;
    SolveLds m;
    lds_carve(L, c.sc, m);
    // What the phase functions are handed is NOT the context (they rebuild it from the kernel arguments, PHASE_ENTER) but two
    // never-written anchors whose only purpose is an address that keeps the calls out of tail position: the real `c` and `m` never
    // have their address taken, so they live in (mostly scalar) registers instead of being re-read from the private stack after every call.
    Ctx pa_c;
    SolveLds pa_m;
    const int max_iters = c.hdr[H_MAXIT];
    const int R = L.R, Rc = L.Rc, nL = c.nL;
    double* out = P.out + (size_t)w * L.ostride;
    int* iout = P.iout + (size_t)w * L.oi_stride;
    double* vG = m.vec + V_G * L.Rpad;
    double* vSC = m.vec + V_SC * L.Rpad;
    double* vDG = m.vec + V_DG * L.Rpad;
    double* vGT = m.vec + V_GT * L.Rpad;
    double* vGN = m.vec + V_GN * L.Rpad;
    double* vU = m.vec + V_U * L.Rpad;
    double* vY = m.vec + V_Y * L.Rpad;
    double* vT = m.vec + V_T * L.Rpad;
    double* sl = c.sc + L.so_sl;
    double* dgl = c.sc + L.so_dg + L.Rpad;
    double* gtl = c.sc + L.so_gt + L.Rpad;
    double* gnl = c.sc + L.so_gn + L.Rpad;
    double* yl = c.sc + L.so_yl;

    PROF_DECL;
    __syncthreads();

    bool assembled = false;
    bool fresh_point = false;              // a new current point whose gradient has to be tested
    if (s.pending) {
        const bool acc = judge_candidate(s, cs_part, L, out, iout, c.tid);
        if (acc) {
            if (!(s.cost == s.cost)) { s.status = VG_ERR_NUMERIC; s.term = VG_TERM_FAILURE; }
            fresh_point = true;
        }
    } else if (!s.scaled) {
        // round 0: cost of the initial point
        s.cost = 0.5 * cs_part;
        s.init_cost = s.cost;
        s.x_norm = sqrt(block_sum(m.red, SV_NW, c.lane, c.wave, state_sqnorm_share(L, c.sc + L.so_x + s.cur * L.nst,
                                                                                     c.sc + L.so_lam + s.cur * L.Lcap, nL, c.tid, SV_NT)));
        if (!(s.cost == s.cost) || !(s.cost < 1e300)) { s.status = VG_ERR_NUMERIC; s.term = VG_TERM_FAILURE; }
        fresh_point = true;
    }
    PROF_ADD(PF_JUDGE);
    const double* buf = lin_buf(c, s.cur);
    const double* hh = buf + L.bo_h;
    const double* bb = buf + L.bo_b;
    if (fresh_point && s.term == VG_TERM_NO_CONVERGENCE && s.status == VG_OK) {
        ctl_uniform(s);
        assemble_small(pa_c, pa_m, buf);
        assembled = true;
        if (!s.scaled) {
            // Jacobi scaling from the first Jacobian, fixed for the solve: 1 / (1 + ||J_col||)
            for (int k = c.tid; k < R; k += SV_NT) c.sc[L.so_sc + k] = 1.0 / (1.0 + sqrt(hess_diag(L, m, k)));
            for (int k = c.tid; k < nL; k += SV_NT) sl[k] = 1.0 / (1.0 + sqrt(hh[k]));
            s.scaled = 1;
            __syncthreads();
        }
        double mx = 0.0;
        for (int k = c.tid; k < R; k += SV_NT) mx = fmax(mx, fabs(vG[k]));
        for (int k = c.tid; k < nL; k += SV_NT) mx = fmax(mx, fabs(bb[k]));
        if (block_max(m.red, SV_NW, c.lane, c.wave, mx) <= 1e-10) s.term = VG_TERM_CONVERGENCE;
    }
    for (int k = c.tid; k < R; k += SV_NT) vSC[k] = c.sc[L.so_sc + k];
    __syncthreads();
    PROF_ADD(PF_ASM);

    const double* x = c.sc + L.so_x + s.cur * L.nst;
    const double* lam = c.sc + L.so_lam + s.cur * L.Lcap;
    double* xc = c.sc + L.so_x + (s.cur ^ 1) * L.nst;
    double* lamc = c.sc + L.so_lam + (s.cur ^ 1) * L.Lcap;
    const double min_mu = 1e-8, max_mu = 1.0;
    (void)min_mu;
    // SOLVER_TIME cap (estimator.cpp:812-815): Ceres tests it before every iteration; past it the solve ends NO_CONVERGENCE
    const bool timed_out = time_is_up(ctlp, c.di[L.do_par + P_MAXTIME], m.red + 30, c.tid);

    while (s.term == VG_TERM_NO_CONVERGENCE && s.status == VG_OK && s.it < max_iters && !timed_out) {
        ++s.it;
        const int slot = s.it - 1;
        bool ok = true;
        if (!s.reuse) {
            s.reuse = 1;
            ctl_uniform(s);
            if (!assembled) { assemble_small(pa_c, pa_m, buf); assembled = true; }
            PROF_ADD(PF_ASM);
            // Dg, gt (scaled gradient / Dg), t = gt / Dg
            for (int k = c.tid; k < R; k += SV_NT) {
                double d2 = vSC[k] * vSC[k] * hess_diag(L, m, k);
                d2 = d2 < 1e-6 ? 1e-6 : d2;               // std::min(std::max(.)) of dogleg_strategy.cc: NaN propagates
                d2 = 1e32 < d2 ? 1e32 : d2;
                const double d = sqrt(d2);
                vDG[k] = d;
                vGT[k] = vSC[k] * vG[k] / d;
                vT[k] = vGT[k] / d;
            }
            for (int k = c.tid; k < nL; k += SV_NT) {
                double d2 = sl[k] * sl[k] * hh[k];
                d2 = d2 < 1e-6 ? 1e-6 : d2;
                d2 = 1e32 < d2 ? 1e32 : d2;
                const double d = sqrt(d2);
                dgl[k] = d;
                gtl[k] = sl[k] * bb[k] / d;
            }
            __syncthreads();
            {
                double sq = 0.0;
                for (int k = c.tid; k < R; k += SV_NT) sq += vGT[k] * vGT[k];
                for (int l = c.tid; l < nL; l += SV_NT) sq += gtl[l] * gtl[l];
                s.gtn2 = block_sum(m.red, SV_NW, c.lane, c.wave, sq);
            }
            PROF_ADD(PF_DG);
            // Gauss-Newton step, increasing mu on failure (DoglegStrategy::ComputeGaussNewtonStep)
            bool solved = false;
            while (s.mu < max_mu) {
                if (!assembled) { assemble_small(pa_c, pa_m, buf); assembled = true; PROF_ADD(PF_ASM); }
                ctl_uniform(s);
                DBG_DUMP(0);
                double q = build_scaled<false>(pa_c, pa_m, s.mu);
                assembled = false;
                __syncthreads();
                DBG_DUMP(1);
                PROF_ADD(PF_BUILD);
                
#if SV_NW >= 8
                // (8-wavefront build: chain elimination on wavefronts 0 .. 3 and landmark columns on 4 .. 7 side by side, tiles owned as in the 4-wavefront kernel)
                q += L.RcPad <= 80 ? chain_schur_split<4>(pa_c, pa_m, buf, s.mu) : chain_schur_split<6>(pa_c, pa_m, buf, s.mu);
#else
                q += L.RcPad <= 80 ? chain_schur<(15 + SV_NW - 1) / SV_NW>(pa_c, pa_m, buf, s.mu) : chain_schur<(21 + SV_NW - 1) / SV_NW>(pa_c, pa_m, buf, s.mu);
#endif
                        // (the coupling rows' share of t^T H~ t)
                bool cok = *(const int*)(m.red + 24) != 0;                // (uniform: read behind the phase's last barrier)
                DBG_DUMP(2);
                PROF_ADD(PF_CHAIN);
                s.qcam = block_sum(m.red, SV_NW, c.lane, c.wave, q);      // t^T H~ t without the landmark part
                s.alpha = -1.0;                  // Cauchy step length |gt|^2 / |J~ (gt/Dg)|^2: completed on demand below
                PROF_ADD(PF_SCHUR);
                if (cok) cok = cholesky_aug(pa_c, pa_m, Rc);
                DBG_DUMP(3);
                PROF_ADD(PF_CHOL);
                if (cok) {
                    back_substitute(pa_c, pa_m, Rc);
                    PROF_ADD(PF_BACK);
                    chain_back_substitute<false>(pa_c, pa_m);
                    PROF_ADD(PF_CBACK);
                    // landmarks: y_l = (bt_l - wt_l . y_cam) / ht_l
                    // four threads per landmark (rows k = q, q + 4, ...; consecutive lanes = consecutive landmarks: coalesced),
                    // eight loads in flight per thread, partial sums combined through LDS in a fixed order
                    const double* Wt = buf + L.bo_Wt;
                    for (int k = c.tid; k < Rc; k += SV_NT) vU[k] = vSC[k] * vY[k];
                    __syncthreads();
                    // (every pass of 64 landmarks has its own SV_NT partial sums in the scratch: ONE barrier for all passes, and the
                    //  loads of a pass are not fenced behind the previous pass's reduction)
                    const int npass = (nL + SV_NT / 4 - 1) / (SV_NT / 4);
                    const bool own_slots = npass * SV_NT <= 36 * L.ldc;
                    for (int ps = 0; ps < npass; ++ps) {
                        const int l0 = ps * (SV_NT / 4);
                        const int l = l0 + (c.tid & (SV_NT / 4 - 1)), qd = c.tid / (SV_NT / 4);
                        double acc = 0.0;
                        if (l < nL) {
                            const double* wp = Wt + l;
                            int k = qd;
                            for (; k + 28 < Rc; k += 32) {
                                double wv[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) wv[u] = wp[(size_t)(k + 4 * u) * L.Lcap];
#pragma unroll
                                for (int u = 0; u < 8; ++u) acc += wv[u] * vU[k + 4 * u];
                            }
                            for (; k < Rc; k += 4) acc += wp[(size_t)k * L.Lcap] * vU[k];
                        }
                        double* slot = m.wd + (own_slots ? ps * SV_NT : 0);
                        slot[c.tid] = acc;
                        if (!own_slots || ps == npass - 1) {
                            __syncthreads();
                            for (int p2 = own_slots ? 0 : ps; p2 <= ps; ++p2) {
                                const double* sl2 = m.wd + (own_slots ? p2 * SV_NT : 0);
                                const int l2 = p2 * (SV_NT / 4) + (c.tid & (SV_NT / 4 - 1));
                                if (qd == 0 && l2 < nL) {
                                    const double a4 = (sl2[c.tid] + sl2[c.tid + SV_NT / 4]) + (sl2[c.tid + SV_NT / 2] + sl2[c.tid + 3 * SV_NT / 4]);
                                    const double ht = sl[l2] * sl[l2] * hh[l2] + s.mu * dgl[l2] * dgl[l2];
                                    yl[l2] = (sl[l2] * bb[l2] - sl[l2] * a4) / ht;
                                }
                            }
                            __syncthreads();
                        }
                    }
                    double fin = 0.0;
                    for (int k = c.tid; k < R; k += SV_NT) fin += (vY[k] == vY[k] && fabs(vY[k]) < 1e300) ? 0.0 : 1.0;
                    for (int k = c.tid; k < nL; k += SV_NT) fin += (yl[k] == yl[k] && fabs(yl[k]) < 1e300) ? 0.0 : 1.0;
                    const bool finite = block_sum(m.red, SV_NW, c.lane, c.wave, fin) == 0.0;
                    PROF_ADD(PF_LMY);
                    if (finite) { solved = true; s.mu_solved = s.mu; break; }
                }
                s.mu *= 10.0;
            }
            if (!solved) ok = false;
            else {
                double s1 = 0.0, s2 = 0.0;
                for (int k = c.tid; k < R; k += SV_NT) {
                    vGN[k] = -vY[k] * vDG[k];
                    s1 += vGN[k] * vGN[k];
                    s2 += vGN[k] * vGT[k];
                }
                for (int k = c.tid; k < nL; k += SV_NT) {
                    gnl[k] = -yl[k] * dgl[k];
                    s1 += gnl[k] * gnl[k];
                    s2 += gnl[k] * gtl[k];
                }
                block_sum2(m.red, SV_NW, c.lane, c.wave, s1, s2);
                s.gnn2 = s1; s.gtgn = s2;
                // keep Dg, gt, gn of this point for step reuse after a rejection (DoglegStrategy keeps them as members)
                for (int k = c.tid; k < R; k += SV_NT) {
                    c.sc[L.so_dg + k] = vDG[k]; c.sc[L.so_gt + k] = vGT[k]; c.sc[L.so_gn + k] = vGN[k];
                }
            }
        } else {
            for (int k = c.tid; k < R; k += SV_NT) {
                vDG[k] = c.sc[L.so_dg + k]; vGT[k] = c.sc[L.so_gt + k]; vGN[k] = c.sc[L.so_gn + k];
            }
            __syncthreads();
        }
        PROF_ADD(PF_NORMS);
        double model_change = 0.0;
        double c_gt = 0.0, c_gn = 0.0;
        if (ok) {
            // DoglegStrategy::ComputeTraditionalDoglegStep
            const double gtn = sqrt(s.gtn2), gnn = sqrt(s.gnn2);
            if (!(gnn <= s.radius) && s.alpha < 0.0) {
                const double ql = block_sum(m.red, SV_NW, c.lane, c.wave, cauchy_landmark_term(pa_c, pa_m, buf));
                s.alpha = s.gtn2 / (s.qcam + ql);
            }
            if (gnn <= s.radius) { c_gt = 0.0; c_gn = 1.0; s.dnorm = gnn; }
            else if (gtn * s.alpha >= s.radius) { c_gt = -(s.radius / gtn); c_gn = 0.0; s.dnorm = s.radius; }
            else {
                const double b_dot_a = -s.alpha * s.gtgn;
                const double a_sq = (s.alpha * gtn) * (s.alpha * gtn);
                const double bma_sq = a_sq - 2 * b_dot_a + s.gnn2;
                const double cc = b_dot_a - a_sq;
                const double dd = sqrt(cc * cc + bma_sq * (s.radius * s.radius - a_sq));
                const double beta = (cc <= 0) ? (dd - cc) / bma_sq : (s.radius * s.radius - a_sq) / (dd + cc);
                c_gt = -s.alpha * (1.0 - beta); c_gn = beta;
                s.dnorm = sqrt(c_gt * c_gt * s.gtn2 + 2 * c_gt * c_gn * s.gtgn + c_gn * c_gn * s.gnn2);
            }
            __syncthreads();
            // delta = scale .* (s ./ Dg)
            for (int k = c.tid; k < R; k += SV_NT) vU[k] = vSC[k] * ((c_gt * vGT[k] + c_gn * vGN[k]) / vDG[k]);
            __syncthreads();
            // model cost change  -(J~ s)^T (r + J~ s / 2)  with s = c_gt a + c_gn b  (a = gt/Dg, b = gn/Dg = -y):
            //   a.g~ = |gt|^2, b.g~ = gt.gn, a^T H~ a = |gt|^2 / alpha, and from (H~ + mu Dg^2) y = g~ :
            //   b^T H~ b = -gt.gn - mu |gn|^2,  a^T H~ b = -|gt|^2 - mu gt.gn      (no pass over the factors)
            const double q11 = c_gt != 0.0 ? s.gtn2 / s.alpha : 0.0;      // (alpha is only evaluated when the step has a gradient part)
            const double q12 = -s.gtn2 - s.mu_solved * s.gtgn;
            const double q22 = -s.gtgn - s.mu_solved * s.gnn2;
            model_change = -(c_gt * s.gtn2 + c_gn * s.gtgn) - 0.5 * (c_gt * c_gt * q11 + 2.0 * c_gt * c_gn * q12 + c_gn * c_gn * q22);
        }
        if (c.tid == 0) {
            out[L.oo_trace + 0 * VG_MAX_ITERS + slot] = s.cost;
            out[L.oo_trace + 3 * VG_MAX_ITERS + slot] = s.radius;
        }
        if (!ok || !(model_change > 0.0)) {
            if (c.tid == 0) {
                out[L.oo_trace + 1 * VG_MAX_ITERS + slot] = 0.0;
                out[L.oo_trace + 2 * VG_MAX_ITERS + slot] = model_change;
                out[L.oo_trace + 4 * VG_MAX_ITERS + slot] = 0.0;
                iout[4 + slot] = 0;
            }
            ++s.ninv;
            if (s.ninv >= 5) { s.term = VG_TERM_FAILURE; break; }
            s.mu *= 10.0;
            s.reuse = 0;
            continue;                   // the linearisation of the unchanged point is still in HBM: next trip re-assembles it
        }
        s.ninv = 0;
        s.model = model_change;
        // ---- candidate = x (+) delta into the other state copy, |x_c - x| and |x_c| in the same pass (one HBM round trip: the
        //      values are summed where they are produced instead of being read back)
        {
            const glb_d* xg = AS_GLB_C(x);
            glb_d* xcg = AS_GLB(xc);
            const glb_d* lamg = AS_GLB_C(lam);
            glb_d* lamcg = AS_GLB(lamc);
            const glb_d* slg = AS_GLB_C(sl);
            const glb_d* gtlg = AS_GLB_C(gtl);
            const glb_d* gnlg = AS_GLB_C(gnl);
            const glb_d* dglg = AS_GLB_C(dgl);
            double sd = 0.0, sn = 0.0;
            // poses: threads 0 .. Kp-1; the extrinsic pose + td: thread 64 (another wavefront)
            const bool isex = c.tid == 64;
            for (int i = isex ? L.Kp : c.tid; i < L.Kp + (isex ? 1 : 0); i += SV_NT) {
                const int xo = isex ? 7 * L.Kp + 9 * L.K : 7 * i;
                double xi[8], o[7], dl[6];
#pragma unroll
                for (int k = 0; k < 8; ++k) xi[k] = xg[xo + (k < 7 || isex ? k : 0)];
                if (!isex || L.e) {
                    const int co = isex ? col_ex(L) : col_pose(L, i);
#pragma unroll
                    for (int k = 0; k < 6; ++k) dl[k] = vU[co + k];
                    pose_plus(xi, dl, o);
#pragma unroll
                    for (int k = 0; k < 7; ++k) { const double d = xi[k] - o[k]; sd += d * d; sn += o[k] * o[k]; }
                } else {
#pragma unroll
                    for (int k = 0; k < 7; ++k) o[k] = xi[k];
                }
#pragma unroll
                for (int k = 0; k < 7; ++k) xcg[xo + k] = o[k];
                if (isex) {
                    const double o7 = L.t ? xi[7] + vU[col_td(L)] : xi[7];
                    xcg[xo + 7] = o7;
                    if (L.t) { const double d = xi[7] - o7; sd += d * d; sn += o7 * o7; }
                }
            }
            // speed-bias entries: from the last thread downwards (the first wavefront has the poses)
            for (int k = SV_NT - 1 - c.tid; k < 9 * L.K; k += SV_NT) {
                const double xv = xg[7 * L.Kp + k];
                const double o = xv + vU[col_sb(L, k / 9) + k % 9];
                xcg[7 * L.Kp + k] = o;
                const double d = xv - o;
                sd += d * d; sn += o * o;
            }
            for (int k = c.tid; k < nL; k += SV_NT) {
                const double lv = lamg[k];
                const double o = lv + slg[k] * ((c_gt * gtlg[k] + c_gn * gnlg[k]) / dglg[k]);
                lamcg[k] = o;
                const double d = lv - o;
                sd += d * d; sn += o * o;
            }
            block_sum2(m.red, SV_NW, c.lane, c.wave, sd, sn);
            s.step_norm = sqrt(sd);
            s.x_norm_c = sqrt(sn);
        }
        if (c.tid == 0) {
            out[L.oo_trace + 2 * VG_MAX_ITERS + slot] = model_change;
            out[L.oo_trace + 4 * VG_MAX_ITERS + slot] = s.dnorm;
        }
        s.pending = 1;
        PROF_ADD(PF_CAND);
        break;
    }
    if (!s.pending) s.done = 1;
    __syncthreads();
    if (c.tid == 0) ctl_store(s, ctlp);
    PROF_ADD(PF_TAIL);
}
#ifndef BA_SOLVE_W8_TU
extern "C" __global__ __launch_bounds__(SV_NT, SV_WG_PER_CU) void ba_solve_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) { solve_body(Lp, P); }
#else
extern "C" __global__ __launch_bounds__(SV_NT, SV_WG_PER_CU) void ba_solve_w8_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) { solve_body(Lp, P); }
}   // namespace vg_w8
#endif
#ifndef BA_SOLVE_W8_TU                  // (from here to the end of the file: the main translation unit only)

// ================================================================================================
// Large-window path (BaLayout::big): windows whose camera part does not fit the LDS carve of ba_solve_kernel (BASELINE
// configs[4]: 31 frames x 2000 landmarks, Rc = 193, R = 472) and windows whose landmarks are SHARDED over ranks.
//
// Landmarks are conditionally independent given the frames, so a rank that holds a contiguous shard of the landmarks
// (and all frames, IMU factors and the prior, replicated) can form its share of the reduced camera system on its own:
//     Sp, gp  = J^T J, J^T r of its projection factors (accumulation kernel)
//     T       = sum_l omega_l [W_l; b_l] [W_l; b_l]^T,   omega_l = sl^2 / (sl^2 h_l + mu Dg_l^2)      (ba_big_schur_kernel, MFMA)
// One all-reduce of [Sp | gp | T | scalars] (RCCL over xGMI, issued by the caller's hook between two launches; ~300 KB for
// Rc = 193) gives every rank the complete reduced system; every rank then factorises it redundantly (no broadcast),
// back-substitutes its own landmarks, and a second, 4-double all-reduce completes |gn|^2, gt.gn and the Cauchy term for
// the dogleg step.  The camera scaling commutes with the Schur complement (diag(sc) T diag(sc)), so T is reduced
// unscaled; the landmark scaling and damping are local to a landmark.  With a pending candidate the Schur kernel runs
// speculatively at the candidate's linearisation with the mu an accepted step would leave (max(1e-8, mu / 5)): a
// rejected step reuses the previous Gauss-Newton step and needs no linear solve, exactly as DoglegStrategy does.
//
//   per round:  linearize_imu, linearize_proj, accumulate, big_schur | all-reduce 1 | solve_big, big_landmark | all-reduce 2 | big_step
//
// Trust-region semantics are those of ba_solve_kernel (one iteration per round; a failed factorisation retries the same
// iteration with mu x 10 in the next round, because the new T needs the collective).
// ================================================================================================

// grid (nts + 1 + BA_BIG_ZERO_BLOCKS, nwin), 256 threads.  Workgroups 0 .. nts-1: one lower 16x16 tile of T each, the four wavefronts take
// interleaved groups of 16 landmarks (a lane loads 4 consecutive landmarks of its row: 32-byte loads, 128 contiguous
// bytes per row), partial tiles are summed through LDS in a fixed order.  Workgroup nts: per-landmark Dg, gt (and the
// landmark scaling in round 0), the scalar partial sums, and the copy of Sp / gp into the reduce buffer.  The last
// BA_BIG_ZERO_BLOCKS workgroups clear the chain blocks of the reduced system (HBM) for the solve kernel's assembly.
extern "C" __global__ __launch_bounds__(256) void ba_big_schur_kernel(const BaLayout* __restrict__ Lp, BaPtrs P, int cost_only) {
    const BaLayout& L = *Lp;
    Ctx c;
    const int w = blockIdx.y;
    ctx_init(c, Lp, P, w);
    const double* ctl = c.sc + L.so_ctl;
    if (ctl[C_DONE] != 0.0) return;
    const bool pending = ctl[C_PENDING] != 0.0;
    const int cur = (int)ctl[C_CUR];
    const int which = cur ^ (pending ? 1 : 0);
    const double mu = pending ? fmax(1e-8, 2.0 * ctl[C_MU] / 10.0) : ctl[C_MU];
    const bool scaled = ctl[C_SCALED] != 0.0;
    const double* buf = lin_buf(c, which);
    const double* hh = buf + L.bo_h;
    const double* bb = buf + L.bo_b;
    double* sl = c.sc + L.so_sl;
    double* rb = P.rb1 + (size_t)w * L.rb1_len;
    const int Rc = L.Rc, nL = c.nL;
    __shared__ double sh[4 * 256];
    if ((int)blockIdx.x == L.nts) {
        const int ntri = Rc * (Rc + 1) / 2;
        if (!cost_only) {
            for (int k = c.tid; k < ntri; k += 256) rb[k] = buf[L.bo_Sp + k];
            for (int k = c.tid; k < Rc; k += 256) rb[ntri + k] = buf[L.bo_gp + k];
        }
        double* dgl = c.sc + L.so_dgl + which * L.Lcap;
        double* gtl = c.sc + L.so_gtl + which * L.Lcap;
        const double* lam = c.sc + L.so_lam + which * L.Lcap;
        const double* lam0 = c.sc + L.so_lam + cur * L.Lcap;
        double cost = 0.0, gt2 = 0.0, lam2 = 0.0, st2 = 0.0, nbig = 0.0;
        for (int b = c.tid; b < L.nbf; b += 256) cost += c.sc[L.so_part + b];
        for (int l = c.tid; l < nL; l += 256) {
            lam2 += lam[l] * lam[l];
            if (pending) { const double d = lam0[l] - lam[l]; st2 += d * d; }
            if (cost_only) continue;
            const double h = hh[l];
            const double s = scaled ? sl[l] : 1.0 / (1.0 + sqrt(h));
            if (!scaled) sl[l] = s;
            double d2 = s * s * h;
            d2 = d2 < 1e-6 ? 1e-6 : d2;
            d2 = 1e32 < d2 ? 1e32 : d2;
            const double d = sqrt(d2);
            dgl[l] = d;
            const double g = s * bb[l] / d;
            gtl[l] = g;
            gt2 += g * g;
            nbig += fabs(bb[l]) > 1e-10 ? 1.0 : 0.0;
        }
        cost = block_sum(sh, 4, c.lane, c.wave, cost);
        block_sum2(sh, 4, c.lane, c.wave, gt2, lam2);
        block_sum2(sh, 4, c.lane, c.wave, st2, nbig);
        if (c.tid == 0) {
            double* sc = rb + L.rb1_scal;
            sc[RB1_COST] = cost; sc[RB1_GTL2] = gt2; sc[RB1_LAM2] = lam2; sc[RB1_STEP2] = st2; sc[RB1_NBIG] = nbig;
            sc[5] = 0.0; sc[6] = 0.0; sc[7] = 0.0;
        }
        return;
    }
    if (cost_only) return;
    if ((int)blockIdx.x > L.nts) {
        // clear the chain blocks XC | D | E (contiguous in the HBM carve) for the assembly of this round
        double* z = c.sc + L.so_bigm + L.l_XC;
        const int n = L.l_dinv - L.l_XC;
        for (int k = ((int)blockIdx.x - L.nts - 1) * 256 + c.tid; k < n; k += BA_BIG_ZERO_BLOCKS * 256) z[k] = 0.0;
        return;
    }
    int tm, tn;
    tri_decode(blockIdx.x, tm, tn);
    const double* Wt = buf + L.bo_Wt;
    const int r = c.lane & 15, kk = c.lane >> 4;
    const double* pa = Wt + (size_t)(tm * 16 + r) * L.Lcap + 4 * kk;
    const double* pb = Wt + (size_t)(tn * 16 + r) * L.Lcap + 4 * kk;
    double4_t acc = {0, 0, 0, 0};
    // Round 6: BS_PF groups of 16 landmarks per trip -- the loads of all of them (operands, h, landmark scaling) are requested before the
    // first weight is formed.  The loop used to wait for a group's eight operand loads, form its four weights (a square root and a
    // division each) and only then ask for the next group: 31 dependent HBM round trips per wavefront, 68 us per launch for 2000
    // landmarks.  Same weights, same products, the MFMAs in the same order: the same T bit for bit.
    constexpr int BS_PF = 4;
    auto omega = [&](double h, double s0) {
        const double s = scaled ? s0 : 1.0 / (1.0 + sqrt(h));
        double d2 = s * s * h;
        d2 = d2 < 1e-6 ? 1e-6 : d2;
        d2 = 1e32 < d2 ? 1e32 : d2;
        return s * s / (s * s * h + mu * d2);
    };
    for (int l0 = 16 * c.wave; l0 < nL; l0 += 64 * BS_PF) {
        double a[BS_PF][4], bq[BS_PF][4], hv[BS_PF][4], sv[BS_PF][4];
#pragma unroll
        for (int u = 0; u < BS_PF; ++u) {
            const int lu = l0 + 64 * u;
            const bool on = lu < nL;                 // (uniform per wavefront; a group past the end contributes exact zeros)
            const int lc = on ? lu : l0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const double av = pa[lc + q], bv = pb[lc + q];      // (Lcap is a multiple of 16 and columns nL .. Lcap-1 of Wt are zero)
                a[u][q] = on ? av : 0.0;
                bq[u][q] = on ? bv : 0.0;
                const int l = lc + 4 * kk + q;
                const bool in = on && l < nL;
                const double h = hh[in ? l : 0], s0 = sl[in ? l : 0];
                hv[u][q] = in ? h : -1.0;            // (-1: no such landmark)
                sv[u][q] = s0;
            }
        }
#pragma unroll
        for (int u = 0; u < BS_PF; ++u) {
            double om[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) om[q] = hv[u][q] != -1.0 ? omega(hv[u][q], sv[u][q]) : 0.0;
            if (l0 + 64 * u < nL) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][q] * om[q], bq[u][q], acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) sh[c.wave * 256 + reg * 64 + c.lane] = acc[reg];
    __syncthreads();
    {
        // element e = reg * 64 + lane of the tile: D[row = (lane >> 4) + 4 reg][col = lane & 15]
        const int e = c.tid, reg = e >> 6, lane = e & 63;
        const double v = (sh[e] + sh[256 + e]) + (sh[512 + e] + sh[768 + e]);
        const int row = tm * 16 + (lane >> 4) + 4 * reg, col = tn * 16 + (lane & 15);
        if (row <= Rc && col <= row) rb[L.rb1_T + tri(row, col)] = v;
    }
}

// The chain elimination of the large-window path, where XC lives in HBM: the same elimination from both ends as chain_schur (storage
// conventions afterwards, so schur_chain_big / chain_back_substitute do not care), organised so that every element of XC is
// loaded once and stored once.  Thread `id` of the first / second half of the workgroup owns column id of [C_k | g_k] for
// the top / bottom sweep and keeps the solved column of the block it eliminated last in registers: the update a block
// receives from its neighbour only involves that column and the neighbour's 9x9 coupling block, which travels through LDS
// (cz: L and 1/L_rr of the two blocks of the step, coupling blocks double-buffered by block parity).
NOINL bool chain_eliminate_big(const Ctx& c_in, const SolveLds& m_in, double* cz_) {
    PHASE_ENTER(true);
    const int K = L.K, Rc = L.Rc, ldc = m.ldc;
    const int mid = K / 2;
    const int nstep = (K - 1 - mid) > mid ? (K - 1 - mid) : mid;
    lds_d* cz = AS_LDS(cz_);
    glb_d* const XC = AS_GLB(m.XC);
    glb_d* const D = AS_GLB(m.D);
    glb_d* const E = AS_GLB(m.E);
    glb_d* const dinv = AS_GLB(m.dinv);
    lds_d* Lt = cz;            lds_d* dit = cz + 81;
    lds_d* Lb = cz + 96;       lds_d* dib = cz + 96 + 81;
    lds_d* XeB = cz + 192;     // [2][96]: Xe of top block k in buffer k & 1
    lds_d* XuB = cz + 384;     // [2][96]: slot E_k (rows solved by the bottom block k-1) in buffer k & 1
    lds_i* flag = (lds_i*)(m.red + 24);
    if (c.tid == 0) *flag = 1;
    __syncthreads();
    const int half = c.tid >> 8, id = c.tid & 255;
    double xprev[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) xprev[r] = 0.0;
    for (int t = 0; t <= nstep; ++t) {
        const bool last = t == nstep;
        const int kt = last ? mid : K - 1 - t, kb = last ? -1 : t;
        const bool has_t = last || kt > mid, has_b = !last && kb < mid;
        const bool upd_t_from_above = has_t && kt + 1 <= K - 1 && (last ? (K - 1 > mid) : t > 0);
        const bool upd_mid_from_below = last && mid > 0;
        const bool upd_b = has_b && t > 0;
        const lds_d* XeP = XeB + 96 * ((kt + 1) & 1);          // coupling block of the upper neighbour kt + 1
        const lds_d* XuM = XuB + 96 * (kt & 1);                // (last step) slot E_mid, solved by the bottom block mid - 1
        const lds_d* XuP = XuB + 96 * (kb & 1);                // slot E_kb, solved by the bottom block kb - 1
        // ---- (round 6) what this step reads from HBM is requested BEFORE the diagonal blocks are factored: a thread's column of
        //      [C_k | g_k] (or its column / row of the coupling block), the two lanes' entries of D_k -- none of it depends on the step's
        //      own results, and the round trip used to start behind the barrier of (A)
        double xq[9], dq0 = 0.0, dq1 = 0.0;
        {
            const bool colT = half == 0 && has_t && id <= Rc, colB = half == 1 && has_b && id <= Rc;
            const bool eT = half == 0 && has_t && !last && id > Rc && id < Rc + 10, eB = half == 1 && has_b && id > Rc && id < Rc + 10;
            const int cc = id - Rc - 1;
            const glb_d* src = colT ? XC + (size_t)9 * kt * ldc + id : (colB ? XC + (size_t)9 * kb * ldc + id
                               : (eT ? E + 81 * kt + cc : (eB ? E + 81 * (kb + 1) + 9 * cc : XC)));
            const size_t st = (colT || colB) ? (size_t)ldc : (eT ? 9 : (eB ? 1 : 0));
#pragma unroll
            for (int r = 0; r < 9; ++r) xq[r] = src[(size_t)r * st];
            if (c.wave == 0 && has_t) { dq0 = D[81 * kt + c.lane]; dq1 = D[81 * kt + (c.lane + 64 < 81 ? c.lane + 64 : 0)]; }
            else if (c.wave == 4 && has_b) { dq0 = D[81 * kb + c.lane]; dq1 = D[81 * kb + (c.lane + 64 < 81 ? c.lane + 64 : 0)]; }
        }
        // ---- (A) diagonal blocks: update + 9x9 Cholesky in LDS, factor back to HBM for the back substitution
        if (c.wave == 0 && has_t) {
            for (int e = c.lane; e < 81; e += 64) {
                const int r = e / 9, cc = e - 9 * r;
                double s = 0.0;
                if (upd_t_from_above) {
#pragma unroll
                    for (int p = 0; p < 9; ++p) s += XeP[9 * p + r] * XeP[9 * p + cc];
                }
                if (upd_mid_from_below) {
#pragma unroll
                    for (int p = 0; p < 9; ++p) s += XuM[9 * r + p] * XuM[9 * cc + p];
                }
                Lt[e] = (e < 64 ? dq0 : dq1) - s;
            }
            __builtin_amdgcn_wave_barrier();
            if (!chain_factor(c.lane, Lt, dit, 0, (const lds_d*)nullptr, (const lds_d*)nullptr) && c.lane == 0) *flag = 0;
            __builtin_amdgcn_wave_barrier();
            for (int e = c.lane; e < 81; e += 64) D[81 * kt + e] = Lt[e];
            if (c.lane < 9) dinv[9 * kt + c.lane] = dit[c.lane];
        } else if (c.wave == 4 && has_b) {
            for (int e = c.lane; e < 81; e += 64) {
                const int r = e / 9, cc = e - 9 * r;
                double s = 0.0;
                if (upd_b) {
#pragma unroll
                    for (int p = 0; p < 9; ++p) s += XuP[9 * r + p] * XuP[9 * cc + p];
                }
                Lb[e] = (e < 64 ? dq0 : dq1) - s;
            }
            __builtin_amdgcn_wave_barrier();
            if (!chain_factor(c.lane, Lb, dib, 0, (const lds_d*)nullptr, (const lds_d*)nullptr) && c.lane == 0) *flag = 0;
            __builtin_amdgcn_wave_barrier();
            for (int e = c.lane; e < 81; e += 64) D[81 * kb + e] = Lb[e];
            if (c.lane < 9) dinv[9 * kb + c.lane] = dib[c.lane];
        }
        __syncthreads();
        if (*flag == 0) break;
        // ---- (B) columns
        if (half == 0 && has_t) {
            if (id <= Rc) {
                glb_d* col = XC + (size_t)9 * kt * ldc + id;
                double x[9];
#pragma unroll
                for (int r = 0; r < 9; ++r) x[r] = xq[r];
                if (upd_t_from_above) {
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        double s = 0.0;
#pragma unroll
                        for (int p = 0; p < 9; ++p) s += XeP[9 * p + r] * xprev[p];
                        x[r] -= s;
                    }
                }
                if (upd_mid_from_below) {
                    const glb_d* cb = XC + (size_t)9 * (kt - 1) * ldc + id;
                    double xb[9];
#pragma unroll
                    for (int p = 0; p < 9; ++p) xb[p] = cb[(size_t)p * ldc];
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        double s = 0.0;
#pragma unroll
                        for (int p = 0; p < 9; ++p) s += XuM[9 * r + p] * xb[p];
                        x[r] -= s;
                    }
                }
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    double s = x[r];
#pragma unroll
                    for (int q = 0; q < r; ++q) s -= Lt[9 * r + q] * x[q];
                    x[r] = s * dit[r];
                }
#pragma unroll
                for (int r = 0; r < 9; ++r) { col[(size_t)r * ldc] = x[r]; xprev[r] = x[r]; }
            } else if (!last && id < Rc + 10) {
                // column cc of E_kt -> Xe_kt[p][cc]
                const int cc = id - Rc - 1;
                glb_d* e = E + 81 * kt + cc;
                lds_d* xe = XeB + 96 * (kt & 1) + cc;
                double x[9];
#pragma unroll
                for (int r = 0; r < 9; ++r) x[r] = xq[r];
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    double s = x[r];
#pragma unroll
                    for (int q = 0; q < r; ++q) s -= Lt[9 * r + q] * x[q];
                    x[r] = s * dit[r];
                }
#pragma unroll
                for (int r = 0; r < 9; ++r) { e[9 * r] = x[r]; xe[9 * r] = x[r]; }
            }
        } else if (half == 1 && has_b) {
            if (id <= Rc) {
                glb_d* col = XC + (size_t)9 * kb * ldc + id;
                double x[9];
#pragma unroll
                for (int r = 0; r < 9; ++r) x[r] = xq[r];
                if (upd_b) {
#pragma unroll
                    for (int r = 0; r < 9; ++r) {
                        double s = 0.0;
#pragma unroll
                        for (int p = 0; p < 9; ++p) s += XuP[9 * r + p] * xprev[p];
                        x[r] -= s;
                    }
                }
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    double s = x[r];
#pragma unroll
                    for (int q = 0; q < r; ++q) s -= Lb[9 * r + q] * x[q];
                    x[r] = s * dib[r];
                }
#pragma unroll
                for (int r = 0; r < 9; ++r) { col[(size_t)r * ldc] = x[r]; xprev[r] = x[r]; }
            } else if (id < Rc + 10) {
                // row cc of E_kb+1 (as [c][p]) -> XuT of the pair (kb + 1, kb)
                const int cc = id - Rc - 1;
                glb_d* e = E + 81 * (kb + 1) + 9 * cc;
                lds_d* xu = XuB + 96 * ((kb + 1) & 1) + 9 * cc;
                double x[9];
#pragma unroll
                for (int r = 0; r < 9; ++r) x[r] = xq[r];
#pragma unroll
                for (int r = 0; r < 9; ++r) {
                    double s = x[r];
#pragma unroll
                    for (int q = 0; q < r; ++q) s -= Lb[9 * r + q] * x[q];
                    x[r] = s * dib[r];
                }
#pragma unroll
                for (int r = 0; r < 9; ++r) { e[r] = x[r]; xu[r] = x[r]; }
            }
        }
        __syncthreads();
    }
    const bool ok = *flag != 0;
    __syncthreads();
    return ok;
}

// The Schur update of the large-window path on many CUs (round 6; between the two launches of ba_solve_big_kernel, which hands S over
// in HBM): S -= X^T X over the 9K eliminated chain rows (XC, HBM / L2) and S -= diag(sc) T diag(sc).  grid (lower 16 x 16 tiles,
// windows), ONE wavefront per tile: it walks the rows in the order the single-workgroup form of rounds 2-5 did (schur_chain_big: two k-steps of
// four rows per trip, trips in ascending order), so the sums are the same; eight trips' operands are requested together.
extern "C" __global__ __launch_bounds__(64) void ba_big_chain_schur_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) {
    const BaLayout& L = *Lp;
    Ctx c;
    const int w = blockIdx.y;
    ctx_init(c, Lp, P, w);
    const double* ctl = c.sc + L.so_ctl;
    if (ctl[C_DONE] != 0.0 || (int)ctl[C_PHASE] != 4) return;
    int tm, tn;
    tri_decode(blockIdx.x, tm, tn);
    const int Rc = L.Rc, ldc = L.ldc;
    const glb_d* XC = AS_GLB_C(c.sc + L.so_bigm + L.l_XC);
    const glb_d* sc = AS_GLB_C(c.sc + L.so_bigm + L.l_vec + V_SC * L.Rpad);
    const glb_d* T = AS_GLB_C(P.rb1 + (size_t)w * L.rb1_len + L.rb1_T);
    glb_d* Sg = AS_GLB(c.sc + L.so_bigm + L.l_Sg);
    const int nk2 = (9 * L.K + 7) / 8;             // trips of two k-steps (4 rows each); XC has up(9K, 8) rows (zero padded)
    const int lane = threadIdx.x;
    const glb_d* xr = XC + (size_t)(lane >> 4) * ldc + (lane & 15);
    double4_t acc = (double4_t){0, 0, 0, 0};
    for (int k0 = 0; k0 < nk2; k0 += 8) {
        double a0[8], a1[8], b0[8], b1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k2 = k0 + u < nk2 ? k0 + u : nk2 - 1;
            const glb_d* r0 = xr + (size_t)k2 * 8 * ldc;
            const glb_d* r1 = r0 + (size_t)4 * ldc;
            a0[u] = r0[tm * 16]; a1[u] = r1[tm * 16];
            b0[u] = r0[tn * 16]; b1[u] = r1[tn * 16];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (k0 + u < nk2) {                     // (uniform)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[u], b0[u], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const int row = tm * 16 + (lane >> 4) + 4 * reg, col = tn * 16 + (lane & 15);
        if (row <= Rc && col <= row && col < Rc) {
            const double sr = row < Rc ? sc[row] : 1.0;
            Sg[tri(row, col)] -= acc[reg] + sr * sc[col] * T[tri(row, col)];
        }
    }
}

// back substitution L^T y = (row R of S) for R <= 64 NR, one wavefront, lane owns entries lane + 64 q of the running rhs
template <int NR>
NOINL void back_substitute_n(const Ctx& c_in, const SolveLds& m_in, int R) {
    PHASE_ENTER(true);
    const lds_d* S = AS_LDS_C(m.S);
    const lds_d* dinvv = AS_LDS_C(m.di);
    glb_d* y = AS_GLB(m.vec + V_Y * L.Rpad);
    __syncthreads();
    if (c.wave == 0) {
        const lds_d* rowR = S + tri(R, 0);
        double v[NR];
#pragma unroll
        for (int q = 0; q < NR; ++q) { const int l = c.lane + 64 * q; v[q] = l < R ? rowR[l] : 0.0; }
#pragma unroll
        for (int qo = NR - 1; qo >= 0; --qo) {
            const int jhi = R - 1 < 64 * qo + 63 ? R - 1 : 64 * qo + 63;
            for (int j = jhi; j >= 64 * qo; --j) {
                const lds_d* rj = S + tri(j, 0);
                const int own = j & 63;
                const double xj = readlane_d(v[qo], own) * dinvv[j];
#pragma unroll
                for (int q = 0; q < NR; ++q) {
                    if (q < qo) v[q] -= rj[c.lane + 64 * q] * xj;
                    else if (q == qo) {
                        const int l = c.lane + 64 * q;
                        const double rv = l < j ? rj[l] : 0.0;
                        v[q] = c.lane == own ? xj : v[q] - rv * xj;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) { const int l = c.lane + 64 * q; if (l < R) y[l] = v[q]; }
    }
    __syncthreads();
}

// A factorisation (or a non-finite step) failed at damping s.mu: DoglegStrategy::ComputeGaussNewtonStep raises mu tenfold and
// tries again; once mu has left [min_mu, max_mu) the iteration is an invalid step (TrustRegionMinimizer: StepIsInvalid).
// `tried`: a solve at s.mu was attempted (false: mu had already left the range).  Returns through s.phase: 3 = same
// iteration, solve again next round; 0 = the iteration ended as invalid.
DEV void big_solve_failed(Ctl& s, const BaLayout& L, double* out, int* iout, int tid, bool tried) {
    if (tried) {
        s.mu *= 10.0;
        if (s.mu < 1.0) { s.phase = 3; return; }
    }
    const int slot = s.it - 1;
    if (tid == 0) {
        out[L.oo_trace + 0 * VG_MAX_ITERS + slot] = s.cost;
        out[L.oo_trace + 3 * VG_MAX_ITERS + slot] = s.radius;
        out[L.oo_trace + 1 * VG_MAX_ITERS + slot] = 0.0;
        out[L.oo_trace + 2 * VG_MAX_ITERS + slot] = 0.0;
        out[L.oo_trace + 4 * VG_MAX_ITERS + slot] = 0.0;
        iout[4 + slot] = 0;
    }
    ++s.ninv;
    if (s.ninv >= 5) s.term = VG_TERM_FAILURE;
    s.mu *= 10.0;
    s.reuse = 0;
    s.phase = 0;
}

// 1 workgroup / window, BA_NT threads, LDS: S (packed, rhs row), reduction scratch, 1/L_jj.
// Round 6: the kernel runs as TWO launches per round (ba_solve_big_kernel = stage 0, ba_solve_big_tail_kernel = stage 1) with
// ba_big_chain_schur_kernel between them.  Stage 0 ends behind the
// chain elimination: the reduced system leaves LDS for HBM (l_Sg) and the control block says "phase 4: Schur update pending";
// the update S -= X^T X + sc T sc then runs one wavefront per 16 x 16 tile on as many CUs as there are tiles (it was 222K of this
// kernel's 954K cycles per round: 40 wavefront tasks of 35 L2 round trips each on ONE CU, bound by that CU's L1 bandwidth);
// stage 1 reloads S and continues with the Cholesky factorisation.  A round that solves nothing (step reuse, failed damping,
// finished window) ends in stage 0 as before and stage 1 returns at its first branch.  Same operations in the same order.
// (two kernels with the SAME explicit arguments: the phase functions find them at a fixed distance in front of the hidden ones)
template <int stage>
__device__ __forceinline__ void solve_big_body(const BaLayout* __restrict__ Lp, const BaPtrs& P) {
    const BaLayout L = layout_load(Lp);
    Ctx c;
    const int w = blockIdx.x;
    ctx_init(c, Lp, P, w);
    PHASE_SELF_CHECK(c);
    double* ctlp = c.sc + L.so_ctl;
    Ctl s;
    ctl_load(s, ctlp);
    if (s.done) return;
    if (stage == 1 && s.phase != 4) return;          // (uniform) nothing was handed over by stage 0
    SolveLds m;
    big_carve(L, c.sc, m);
    const int max_iters = c.hdr[H_MAXIT];
    const int R = L.R, Rc = L.Rc, nL = c.nL;
    double* out = P.out + (size_t)w * L.ostride;
    int* iout = P.iout + (size_t)w * L.oi_stride;
    const double* rb1 = P.rb1 + (size_t)w * L.rb1_len;
    const double* scal = rb1 + L.rb1_scal;
    double* vG = m.vec + V_G * L.Rpad;
    double* vSC = m.vec + V_SC * L.Rpad;
    double* vDG = m.vec + V_DG * L.Rpad;
    double* vGT = m.vec + V_GT * L.Rpad;
    double* vGN = m.vec + V_GN * L.Rpad;
    double* vU = m.vec + V_U * L.Rpad;
    double* vY = m.vec + V_Y * L.Rpad;
    double* vT = m.vec + V_T * L.Rpad;
    const double* sl = c.sc + L.so_sl;
    double* gnl = c.sc + L.so_gn + L.Rpad;
    double* yl = c.sc + L.so_yl;
    PROF_DECL;
    const bool resume = stage == 1;                  // (uniform) second half of a round whose first half built and eliminated
    glb_d* Sg = AS_GLB(c.sc + L.so_bigm + L.l_Sg);
    const int nS = (Rc + 1) * (Rc + 2) / 2;
    if (resume) {
        // the reduced system after ba_big_chain_schur_kernel, back into LDS (the loads of a thread requested together)
        for (int k0 = c.tid; k0 < nS; k0 += 8 * BA_NT) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int k = k0 + u * BA_NT; v[u] = k < nS ? Sg[k] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int k = k0 + u * BA_NT; if (k < nS) m.S[k] = v[u]; }
        }
        s.phase = 0;
    }
    if (!resume && c.nprior) {
        const int* kind = c.ia + L.io_pb_kind;
        const int* off = c.ia + L.io_pb_off;
        const int* pcol = c.ia + L.io_pb_col;
        for (int blk = c.tid; blk < c.nblk; blk += BA_NT) {
            const int sz = (kind[blk] == VG_BLK_SPEEDBIAS) ? 9 : (kind[blk] == VG_BLK_TD ? 1 : 6);
            for (int k = 0; k < sz; ++k) m.pmap[off[blk] + k] = pcol[blk] >= 0 ? pcol[blk] + k : -1;
        }
    }
    __syncthreads();
    const int phase_in = resume ? 0 : s.phase;
    if (!resume) s.phase = 0;
    // cost of the point the linearisation kernels just evaluated: projection factors of all ranks + this rank's copy of the
    // (replicated) IMU / prior factors
    const double cs = scal[RB1_COST] + sum_partials(c.sc + L.so_part + L.nbf, L.nig + L.nprw);
    bool fresh_point = false;
    if (resume) {
    } else if (s.pending) {
        s.step_norm = sqrt(s.step2c + scal[RB1_STEP2]);
        s.x_norm_c = sqrt(s.xn2c + scal[RB1_LAM2]);
        const bool acc = judge_candidate(s, cs, L, out, iout, c.tid);
        if (acc) {
            if (!(s.cost == s.cost)) { s.status = VG_ERR_NUMERIC; s.term = VG_TERM_FAILURE; }
            fresh_point = true;
        }
    } else if (!s.scaled) {
        s.cost = 0.5 * cs;
        s.init_cost = s.cost;
        const double xc2 = block_sum(m.red, BA_NW, c.lane, c.wave, state_sqnorm_share(L, c.sc + L.so_x + s.cur * L.nst, nullptr, 0, c.tid, BA_NT));
        s.x_norm = sqrt(xc2 + scal[RB1_LAM2]);
        if (!(s.cost == s.cost) || !(s.cost < 1e300)) { s.status = VG_ERR_NUMERIC; s.term = VG_TERM_FAILURE; }
        fresh_point = true;
    }
    PROF_ADD(PF_JUDGE);
    const double* buf = lin_buf(c, s.cur);
    const double* hh = buf + L.bo_h;
    const double* bb = buf + L.bo_b;
    const bool running = s.term == VG_TERM_NO_CONVERGENCE && s.status == VG_OK;
    // (stage 1 takes the time test of stage 0: one decision per round)
    const bool timed_out = resume ? uni(ctlp[C_TIMEDOUT]) != 0.0 : time_is_up(ctlp, c.di[L.do_par + P_MAXTIME], m.red + 30, c.tid);
    const bool trip = resume || (running && !timed_out && (phase_in == 3 || s.it < max_iters));
    const bool need_system = !resume && running && (fresh_point || (trip && !s.reuse));
    if (need_system) {
        const int ntri = Rc * (Rc + 1) / 2;
        assemble_big(c, m, buf, rb1, rb1 + ntri);
        if (!s.scaled) {
            for (int k = c.tid; k < R; k += BA_NT) c.sc[L.so_sc + k] = 1.0 / (1.0 + sqrt(hess_diag(L, m, k)));
            s.scaled = 1;
            __syncthreads();
        }
        if (fresh_point) {
            double mx = 0.0;
            for (int k = c.tid; k < R; k += BA_NT) mx = fmax(mx, fabs(vG[k]));
            if (block_max(m.red, BA_NW, c.lane, c.wave, mx) <= 1e-10 && scal[RB1_NBIG] == 0.0) s.term = VG_TERM_CONVERGENCE;
        }
    }
    for (int k = c.tid; k < R; k += BA_NT) vSC[k] = c.sc[L.so_sc + k];
    __syncthreads();
    PROF_ADD(PF_ASM);
    if (trip && s.term == VG_TERM_NO_CONVERGENCE) {
        if (!resume && phase_in != 3) ++s.it;
        if (!resume && s.reuse) s.phase = 2;
        else if (!resume && !(s.mu < 1.0)) big_solve_failed(s, L, out, iout, c.tid, false);
        else {
            bool cok = true;
            if (!resume) {
            const double* dgl = c.sc + L.so_dgl + s.cur * L.Lcap;
            const double* gtl = c.sc + L.so_gtl + s.cur * L.Lcap;
            for (int k = c.tid; k < R; k += BA_NT) {
                double d2 = vSC[k] * vSC[k] * hess_diag(L, m, k);
                d2 = d2 < 1e-6 ? 1e-6 : d2;
                d2 = 1e32 < d2 ? 1e32 : d2;
                const double d = sqrt(d2);
                vDG[k] = d;
                vGT[k] = vSC[k] * vG[k] / d;
                vT[k] = vGT[k] / d;
            }
            __syncthreads();
            {
                double sq = 0.0;
                for (int k = c.tid; k < R; k += BA_NT) sq += vGT[k] * vGT[k];
                s.gtn2 = block_sum(m.red, BA_NW, c.lane, c.wave, sq) + scal[RB1_GTL2];
            }
            PROF_ADD(PF_DG);
            const double q = build_scaled<true>(c, m, s.mu);
            __syncthreads();
            PROF_ADD(PF_BUILD);
            cok = chain_eliminate_big(c, m, LDSB + L.l_cz);
            PROF_ADD(PF_CHAIN);
            s.qcam = block_sum(m.red, BA_NW, c.lane, c.wave, q);
            if (cok) {
                // ---- hand-over: S to HBM, "phase 4", the round continues in ba_big_chain_schur_kernel and in stage 1
                for (int k = c.tid; k < nS; k += BA_NT) Sg[k] = m.S[k];
                s.phase = 4;
                __syncthreads();
                if (c.tid == 0) { ctl_store(s, ctlp); ctlp[C_TIMEDOUT] = timed_out ? 1.0 : 0.0; }
                return;
            }
            }   // !resume
            PROF_ADD(PF_SCHUR);
            if (cok) cok = cholesky_aug(c, m, Rc);
            PROF_ADD(PF_CHOL);
            if (cok) {
                back_substitute_n<4>(c, m, Rc);
                PROF_ADD(PF_BACK);
                chain_back_substitute<true>(c, m);
                PROF_ADD(PF_CBACK);
                double fin = 0.0, s1 = 0.0, s2 = 0.0;
                for (int k = c.tid; k < R; k += BA_NT) {
                    fin += (vY[k] == vY[k] && fabs(vY[k]) < 1e300) ? 0.0 : 1.0;
                    vGN[k] = -vY[k] * vDG[k];
                    s1 += vGN[k] * vGN[k];
                    s2 += vGN[k] * vGT[k];
                }
                // operands of ba_big_landmark_kernel: sc .* y and sc .* t on the camera columns (t = gt / Dg)
                for (int k = c.tid; k < Rc; k += BA_NT) { vU[k] = vSC[k] * vY[k]; vT[k] = vSC[k] * vT[k]; }
                block_sum2(m.red, BA_NW, c.lane, c.wave, s1, s2);
                fin = block_sum(m.red, BA_NW, c.lane, c.wave, fin);
                cok = fin == 0.0;
                PROF_ADD(PF_LMY);
                if (cok) {
                    s.gnn2c = s1; s.gtgnc = s2;
                    s.mu_solved = s.mu;
                    s.reuse = 1;
                    s.phase = 1;
                    for (int k = c.tid; k < R; k += BA_NT) {
                        c.sc[L.so_dg + k] = vDG[k]; c.sc[L.so_gt + k] = vGT[k]; c.sc[L.so_gn + k] = vGN[k];
                    }
                }
            }
            if (!cok) big_solve_failed(s, L, out, iout, c.tid, true);
        }
    }
    if (s.term != VG_TERM_NO_CONVERGENCE || s.status != VG_OK || (s.phase == 0 && (s.it >= max_iters || timed_out))) s.done = 1;
    __syncthreads();
    if (c.tid == 0) ctl_store(s, ctlp);
}
extern "C" __global__ __launch_bounds__(BA_NT) void ba_solve_big_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) { solve_big_body<0>(Lp, P); }
extern "C" __global__ __launch_bounds__(BA_NT) void ba_solve_big_tail_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) { solve_big_body<1>(Lp, P); }

// Landmark part of the Gauss-Newton step (this rank's landmarks): y_l = (bt_l - wt_l . y_cam) / ht_l, and their shares of
// |gn|^2, gt.gn and the Cauchy-point term  sum_l [ h~_l t_l^2 + 2 t_l (w~_l . t_cam) ]  (t = gt / Dg).  grid
// (BA_BIG_LM_BLOCKS, nwin), 256 threads, thread per landmark (grid-stride), the two camera vectors staged in LDS; one group
// of four partial sums per workgroup into reduce buffer 2.
extern "C" __global__ __launch_bounds__(256) void ba_big_landmark_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) {
    const BaLayout& L = *Lp;
    Ctx c;
    const int w = blockIdx.y;
    ctx_init(c, Lp, P, w);
    const double* ctl = c.sc + L.so_ctl;
    if (ctl[C_DONE] != 0.0 || (int)ctl[C_PHASE] != 1) return;
    const int cur = (int)ctl[C_CUR];
    const double mu = ctl[C_MUSOLVED];
    const int Rc = L.Rc, nL = c.nL;
    const double* vec = c.sc + L.so_bigm + L.l_vec;
    const double* buf = lin_buf(c, cur);
    const double* hh = buf + L.bo_h;
    const double* bb = buf + L.bo_b;
    const double* Wt = buf + L.bo_Wt;
    const double* sl = c.sc + L.so_sl;
    const double* dgl = c.sc + L.so_dgl + cur * L.Lcap;
    const double* gtl = c.sc + L.so_gtl + cur * L.Lcap;
    double* gnl = c.sc + L.so_gn + L.Rpad;
    double* yl = c.sc + L.so_yl;
    __shared__ double su[256], st[256], red[16];
    for (int k = c.tid; k < Rc; k += 256) { su[k] = vec[V_U * L.Rpad + k]; st[k] = vec[V_T * L.Rpad + k]; }
    __syncthreads();
    double p0 = 0.0, p1 = 0.0, p2 = 0.0, fin = 0.0;
    for (int l = blockIdx.x * 256 + c.tid; l < nL; l += BA_BIG_LM_BLOCKS * 256) {
        const double ht = sl[l] * sl[l] * hh[l] + mu * dgl[l] * dgl[l];
        double acc = 0.0, wdot = 0.0;
        const double* wp = Wt + l;
        for (int k = 0; k < Rc; ++k) {
            const double wv = wp[(size_t)k * L.Lcap];
            acc += wv * su[k];
            wdot += wv * st[k];
        }
        const double y = (sl[l] * bb[l] - sl[l] * acc) / ht;
        yl[l] = y;
        fin += (y == y && fabs(y) < 1e300) ? 0.0 : 1.0;
        const double g = -y * dgl[l];
        gnl[l] = g;
        p0 += g * g;
        p1 += g * gtl[l];
        const double tl = gtl[l] / dgl[l];
        p2 += sl[l] * sl[l] * hh[l] * tl * tl + 2.0 * tl * sl[l] * wdot;
    }
    block_sum2(red, 4, c.lane, c.wave, p0, p1);
    block_sum2(red, 4, c.lane, c.wave, p2, fin);
    if (c.tid == 0) {
        double* rb2 = P.rb2 + (size_t)w * RB2_LEN + 4 * blockIdx.x;
        rb2[RB2_GNN2] = p0; rb2[RB2_GTGN] = p1; rb2[RB2_QL] = p2; rb2[RB2_NONFIN] = fin;
    }
}

// Dogleg step, model cost change and the candidate state from the completed norms (after all-reduce 2).  1 workgroup per
// window, no dynamic LDS.
extern "C" __global__ __launch_bounds__(BA_NT) void ba_big_step_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) {
    const BaLayout& L = *Lp;
    Ctx c;
    const int w = blockIdx.x;
    ctx_init(c, Lp, P, w);
    double* ctlp = c.sc + L.so_ctl;
    Ctl s;
    ctl_load(s, ctlp);
    if (s.done || s.phase == 0 || s.phase == 3) return;
    __shared__ double red[32];
    const int max_iters = c.hdr[H_MAXIT];
    const int R = L.R, nL = c.nL;
    double* out = P.out + (size_t)w * L.ostride;
    int* iout = P.iout + (size_t)w * L.oi_stride;
    const double* rb2 = P.rb2 + (size_t)w * RB2_LEN;
    double* vec = c.sc + L.so_bigm + L.l_vec;
    const double* vSC = vec + V_SC * L.Rpad;
    double* vU = vec + V_U * L.Rpad;
    const double* vDG = c.sc + L.so_dg;
    const double* vGT = c.sc + L.so_gt;
    const double* vGN = c.sc + L.so_gn;
    const double* sl = c.sc + L.so_sl;
    const double* dgl = c.sc + L.so_dgl + s.cur * L.Lcap;
    const double* gtl = c.sc + L.so_gtl + s.cur * L.Lcap;
    const double* gnl = c.sc + L.so_gn + L.Rpad;
    const int slot = s.it - 1;
    double r_gnn2 = 0.0, r_gtgn = 0.0, r_ql = 0.0, r_nonfin = 0.0;
    for (int b = 0; b < BA_BIG_LM_BLOCKS; ++b) {
        r_gnn2 += rb2[4 * b + RB2_GNN2]; r_gtgn += rb2[4 * b + RB2_GTGN]; r_ql += rb2[4 * b + RB2_QL]; r_nonfin += rb2[4 * b + RB2_NONFIN];
    }
    if (s.phase == 1) {
        if (r_nonfin != 0.0) {
            big_solve_failed(s, L, out, iout, c.tid, true);
            if (s.term != VG_TERM_NO_CONVERGENCE || (s.phase == 0 && s.it >= max_iters)) s.done = 1;
            __syncthreads();
            if (c.tid == 0) ctl_store(s, ctlp);
            return;
        }
        s.gnn2 = s.gnn2c + r_gnn2;
        s.gtgn = s.gtgnc + r_gtgn;
        s.alpha = s.gtn2 / (s.qcam + r_ql);
    }
    s.phase = 0;
    // DoglegStrategy::ComputeTraditionalDoglegStep (same arithmetic as ba_solve_kernel)
    double c_gt, c_gn;
    const double gtn = sqrt(s.gtn2), gnn = sqrt(s.gnn2);
    if (gnn <= s.radius) { c_gt = 0.0; c_gn = 1.0; s.dnorm = gnn; }
    else if (gtn * s.alpha >= s.radius) { c_gt = -(s.radius / gtn); c_gn = 0.0; s.dnorm = s.radius; }
    else {
        const double b_dot_a = -s.alpha * s.gtgn;
        const double a_sq = (s.alpha * gtn) * (s.alpha * gtn);
        const double bma_sq = a_sq - 2 * b_dot_a + s.gnn2;
        const double cc = b_dot_a - a_sq;
        const double dd = sqrt(cc * cc + bma_sq * (s.radius * s.radius - a_sq));
        const double beta = (cc <= 0) ? (dd - cc) / bma_sq : (s.radius * s.radius - a_sq) / (dd + cc);
        c_gt = -s.alpha * (1.0 - beta); c_gn = beta;
        s.dnorm = sqrt(c_gt * c_gt * s.gtn2 + 2 * c_gt * c_gn * s.gtgn + c_gn * c_gn * s.gnn2);
    }
    const double q11 = c_gt != 0.0 ? s.gtn2 / s.alpha : 0.0;
    const double q12 = -s.gtn2 - s.mu_solved * s.gtgn;
    const double q22 = -s.gtgn - s.mu_solved * s.gnn2;
    const double model_change = -(c_gt * s.gtn2 + c_gn * s.gtgn) - 0.5 * (c_gt * c_gt * q11 + 2.0 * c_gt * c_gn * q12 + c_gn * c_gn * q22);
    if (c.tid == 0) {
        out[L.oo_trace + 0 * VG_MAX_ITERS + slot] = s.cost;
        out[L.oo_trace + 3 * VG_MAX_ITERS + slot] = s.radius;
    }
    if (!(model_change > 0.0)) {
        if (c.tid == 0) {
            out[L.oo_trace + 1 * VG_MAX_ITERS + slot] = 0.0;
            out[L.oo_trace + 2 * VG_MAX_ITERS + slot] = model_change;
            out[L.oo_trace + 4 * VG_MAX_ITERS + slot] = 0.0;
            iout[4 + slot] = 0;
        }
        ++s.ninv;
        if (s.ninv >= 5) s.term = VG_TERM_FAILURE;
        s.mu *= 10.0;
        s.reuse = 0;
        if (s.term != VG_TERM_NO_CONVERGENCE || s.it >= max_iters) s.done = 1;
        __syncthreads();
        if (c.tid == 0) ctl_store(s, ctlp);
        return;
    }
    s.ninv = 0;
    s.model = model_change;
    const double* x = c.sc + L.so_x + s.cur * L.nst;
    const double* lam = c.sc + L.so_lam + s.cur * L.Lcap;
    double* xc = c.sc + L.so_x + (s.cur ^ 1) * L.nst;
    double* lamc = c.sc + L.so_lam + (s.cur ^ 1) * L.Lcap;
    for (int k = c.tid; k < R; k += BA_NT) vU[k] = vSC[k] * ((c_gt * vGT[k] + c_gn * vGN[k]) / vDG[k]);
    __syncthreads();
    for (int i = c.tid; i < L.Kp; i += BA_NT) pose_plus(x + 7 * i, vU + col_pose(L, i), xc + 7 * i);
    for (int k = c.tid; k < 9 * L.K; k += BA_NT) xc[7 * L.Kp + k] = x[7 * L.Kp + k] + vU[col_sb(L, k / 9) + k % 9];
    if (c.tid == 0) {
        double* exc = xc + 7 * L.Kp + 9 * L.K;
        const double* exx = x + 7 * L.Kp + 9 * L.K;
        if (L.e) pose_plus(exx, vU + col_ex(L), exc);
        else for (int k = 0; k < 7; ++k) exc[k] = exx[k];
        exc[7] = L.t ? exx[7] + vU[col_td(L)] : exx[7];
    }
    for (int k = c.tid; k < nL; k += BA_NT) lamc[k] = lam[k] + sl[k] * ((c_gt * gtl[k] + c_gn * gnl[k]) / dgl[k]);
    __syncthreads();
    {
        double sd = 0.0;
        const int nx = 7 * L.Kp + 9 * L.K + (L.e ? 7 : 0);
        for (int k = c.tid; k < nx; k += BA_NT) { const double d = x[k] - xc[k]; sd += d * d; }
        if (L.t && c.tid == 0) { const double d = x[7 * L.Kp + 9 * L.K + 7] - xc[7 * L.Kp + 9 * L.K + 7]; sd += d * d; }
        double sn = state_sqnorm_share(L, xc, nullptr, 0, c.tid, BA_NT);
        block_sum2(red, BA_NW, c.lane, c.wave, sd, sn);
        s.step2c = sd;                              // camera / speed-bias shares; the landmark shares are formed by the
        s.xn2c = sn;                                // next Schur kernel and rank-summed with reduce buffer 1
    }
    if (c.tid == 0) {
        out[L.oo_trace + 2 * VG_MAX_ITERS + slot] = model_change;
        out[L.oo_trace + 4 * VG_MAX_ITERS + slot] = s.dnorm;
    }
    s.pending = 1;
    __syncthreads();
    if (c.tid == 0) ctl_store(s, ctlp);
}

// ================================================================================================
// Final kernel: judge the last candidate, then Estimator::double2vector() (estimator.cpp:530-577: yaw / position gauge
// fix) + the vector2double() repack (:486-528) into the output slab.
// ================================================================================================
extern "C" __global__ __launch_bounds__(256) void ba_final_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) {
    const BaLayout& L = *Lp;
    Ctx c;
    const int w = blockIdx.x;
    ctx_init(c, Lp, P, w);
    const int NT = 256;
    double* ctlp = c.sc + L.so_ctl;
    double* out = P.out + (size_t)w * L.ostride;
    int* iout = P.iout + (size_t)w * L.oi_stride;
    Ctl s;
    ctl_load(s, ctlp);
    if (s.pending) {
        double cs;
        if (L.big) {
            // large-window path: projection cost and landmark norms of all ranks through reduce buffer 1 (ba_big_schur_kernel)
            const double* scal = P.rb1 + (size_t)w * L.rb1_len + L.rb1_scal;
            cs = scal[RB1_COST] + sum_partials(c.sc + L.so_part + L.nbf, L.nig + L.nprw);
            s.step_norm = sqrt(s.step2c + scal[RB1_STEP2]);
            s.x_norm_c = sqrt(s.xn2c + scal[RB1_LAM2]);
        } else {
            cs = sum_partials(c.sc + L.so_part, L.nbl);
        }
        const bool acc = judge_candidate(s, cs, L, out, iout, c.tid);
        if (acc && !(s.cost == s.cost)) { s.status = VG_ERR_NUMERIC; s.term = VG_TERM_FAILURE; }
    }
    const double* x = c.sc + L.so_x + s.cur * L.nst;
    const double* lam = c.sc + L.so_lam + s.cur * L.Lcap;
    const double* p0_in = c.di + L.do_pose;          // pre-solve frame 0
    double Rs0[9], R00[9], y0[3], y00[3], rot[9];
    q_to_R(p0_in + 3, Rs0);
    q_to_R(x + 3, R00);
    R2ypr_dev(Rs0, y0);
    R2ypr_dev(R00, y00);
    const double yd = (y0[0] - y00[0]) / 180.0 * M_PI;
    rot[0] = cos(yd); rot[1] = -sin(yd); rot[2] = 0;
    rot[3] = sin(yd); rot[4] = cos(yd);  rot[5] = 0;
    rot[6] = 0;       rot[7] = 0;        rot[8] = 1;
    if (fabs(fabs(y0[1]) - 90) < 1.0 || fabs(fabs(y00[1]) - 90) < 1.0) m3_mul_t(Rs0, R00, rot);
    for (int i = c.tid; i < L.Kp; i += NT) {
        // frames of the window and (i == K) the relocalisation pose: same gauge transform (estimator.cpp:598-603)
        double q[4] = {x[7 * i + 3], x[7 * i + 4], x[7 * i + 5], x[7 * i + 6]};
        q_normalize(q);
        double Rq[9], Ri[9], qo[4], d[3], po[3];
        q_to_R(q, Rq);
        m3_mul(rot, Rq, Ri);
        R_to_q(Ri, qo);
        d[0] = x[7 * i] - x[0]; d[1] = x[7 * i + 1] - x[1]; d[2] = x[7 * i + 2] - x[2];
        m3_vec(rot, d, po);
        double* o = out + L.oo_pose + 7 * i;
        o[0] = po[0] + p0_in[0]; o[1] = po[1] + p0_in[1]; o[2] = po[2] + p0_in[2];
        o[3] = qo[0]; o[4] = qo[1]; o[5] = qo[2]; o[6] = qo[3];
        if (i < L.K) {
            const double* sb = x + 7 * L.Kp + 9 * i;
            double vo[3];
            m3_vec(rot, sb, vo);
            double* os = out + L.oo_sb + 9 * i;
            os[0] = vo[0]; os[1] = vo[1]; os[2] = vo[2];
            for (int k = 3; k < 9; ++k) os[k] = sb[k];
        }
    }
    if (c.tid == 0) {
        const double* exx = x + 7 * L.Kp + 9 * L.K;
        double Rcm[9], qo[4];
        q_to_R(exx + 3, Rcm);
        R_to_q(Rcm, qo);
        double* o = out + L.oo_ex;
        o[0] = exx[0]; o[1] = exx[1]; o[2] = exx[2]; o[3] = qo[0]; o[4] = qo[1]; o[5] = qo[2]; o[6] = qo[3];
        out[L.oo_td] = exx[7];
        out[L.oo_sum + 0] = s.init_cost;
        out[L.oo_sum + 1] = s.cost;
        out[L.oo_sum + 2] = s.radius;
        // the gauge transform itself (estimator.cpp:541-556): x_fixed = rot (x - p0_solved) + P0_before — lets the caller map
        // quantities outside the window (the relocalisation pose when it carries no factor, :598-603)
        for (int k = 0; k < 9; ++k) out[L.oo_sum + 3 + k] = rot[k];
        for (int k = 0; k < 3; ++k) out[L.oo_sum + 12 + k] = x[k];
        iout[0] = s.status; iout[1] = s.term; iout[2] = s.it; iout[3] = s.nacc;
        s.done = 1;
        ctl_store(s, ctlp);
    }
    // setDepth/getDepthVector round trip (feature_manager.cpp:141-200)
    for (int k = c.tid; k < c.nL; k += NT) out[L.oo_lam + k] = 1.0 / (1.0 / lam[k]);
}

// ================================================================================================
// Factor-evaluation kernel for parity tests (vg_ba_eval_factors): raw (no loss) residuals/Jacobians at the input state.
// proj_J [F][2][20] = [pose_i 6 | pose_j 6 | ex 6 | lambda | td];  imu_J [K-1][15][30].  Runs after ba_prologue_kernel
// (which leaves the sqrt_info factors and state copy 0 in scratch).
// ================================================================================================
extern "C" __global__ __launch_bounds__(BA_NT) void ba_eval_factors_kernel(const BaLayout* __restrict__ Lp, BaPtrs P, double* proj_r,
                                                                       double* proj_J, double* imu_r, double* imu_J, double* prior_r) {
    const BaLayout& L = *Lp;
    Ctx c;
    ctx_init(c, Lp, P, 0);
    const double* x = c.sc + L.so_x;
    const double* lam = c.sc + L.so_lam;
    double* buf = lin_buf(c, 0);
    prior_pass(c, x, buf + L.bo_pr, nullptr, LDSB);
    __syncthreads();
    const int nimu = L.K - 1;
    // weighted IMU residual / Jacobian, thread = (factor, column | residual); U = sqrt_info from the prologue
    for (int w = c.tid; w < nimu * 31; w += BA_NT) {
        const int f = w / 31, col = w % 31;
        if (!c.ia[L.io_imu_valid + f]) continue;
        const double* pre = c.di + L.do_imu + f * BA_IMU_STRIDE;
        const double* U = c.sc + L.so_imuU + f * 225;
        ImuCtx ic;
        imu_ctx<true>(pre, st_pose(L, x, f), st_sb(L, x, f), st_pose(L, x, f + 1), st_sb(L, x, f + 1), c.gnorm, ic);
        double raw[15];
        if (col < 30) imu_raw_col(ic, pre, col, raw);
        else { for (int q = 0; q < 15; ++q) raw[q] = ic.r[q]; }
        for (int r = 0; r < 15; ++r) {
            double s = 0.0;
            for (int k = r; k < 15; ++k) s += U[r * 15 + k] * raw[k];
            if (col < 30) { if (imu_J) imu_J[(size_t)f * 450 + r * 30 + col] = s; }
            else if (imu_r) imu_r[f * 15 + r] = s;
        }
    }
    for (int k = c.tid; k < c.nprior; k += BA_NT) if (prior_r) prior_r[k] = buf[L.bo_pr + k];
    const double* ex = st_ex(L, x);
    for (int f = c.tid; f < c.nF; f += BA_NT) {
        ProjIn p;
        proj_fetch(c, f, x, lam, p);
        double r[2], Ji[12], Jj[12], Jex[12], Jl[2], Jtd[2] = {0, 0};
        for (int k = 0; k < 12; ++k) Jex[k] = 0.0;
        if (L.t) proj_eval<true, true, true>(p.pi, p.pj, ex, p.lam, p.oi, p.oj, ex[7], c.focal, c.tr, c.row, r, Ji, Jj, Jex, Jl, Jtd);
        else proj_eval<false, true, true>(p.pi, p.pj, ex, p.lam, p.oi, p.oj, 0.0, c.focal, c.tr, c.row, r, Ji, Jj, Jex, Jl, Jtd);
        if (proj_r) { proj_r[2 * f] = r[0]; proj_r[2 * f + 1] = r[1]; }
        if (proj_J) {
            for (int rr = 0; rr < 2; ++rr) {
                double* o = proj_J + (size_t)f * 40 + rr * 20;
                for (int k = 0; k < 6; ++k) { o[k] = Ji[rr * 6 + k]; o[6 + k] = Jj[rr * 6 + k]; o[12 + k] = Jex[rr * 6 + k]; }
                o[18] = Jl[rr]; o[19] = Jtd[rr];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Host-side launch sequence of one batch solve (rounds = max over the windows of max_iters).
// hipFuncSetAttribute applies to the function object of the CURRENT device: the set-up is tracked per device (a process may hold
// handles on several GPUs) under a mutex (handles may be driven from several host threads).
// (defined by ba_solve_w8.hip: this file compiled once more with SV_NT = 512, see the top of the file)
extern "C" __global__ void ba_solve_w8_kernel(const BaLayout* __restrict__ Lp, BaPtrs P);
static hipError_t set_lds_attrs() {
    static std::mutex mu;
    static unsigned long long done_mask = 0;      // bit d: device d is set up
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (dev >= 0 && dev < 64 && ((done_mask >> dev) & 1ull)) return hipSuccess;
    e = hipFuncSetAttribute((const void*)ba_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ba_solve_w8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ba_solve_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ba_solve_big_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ba_prologue_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ba_linacc_proj_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    // (these two also hold a few statically allocated LDS words: static + dynamic must stay within 160 KB)
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ba_linearize_imu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ba_eval_factors_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    if (e == hipSuccess && dev >= 0 && dev < 64) done_mask |= 1ull << dev;
    return e;
}

// (stream capture must not meet the first-use set-up above: the host calls this before hipStreamBeginCapture)
extern "C" hipError_t ba_prepare_launch() { return set_lds_attrs(); }

// last failing launch of this translation unit (for the error text of the C-ABI)
static const char* g_failed_launch = "";
extern "C" const char* ba_failed_launch() { return g_failed_launch; }
#define LAUNCH(name, grid, block, lds, ...)                                              \
    do {                                                                                 \
        hipLaunchKernelGGL(name, grid, block, lds, stream, __VA_ARGS__);                 \
        const hipError_t _le = hipGetLastError();                                        \
        if (_le != hipSuccess) { g_failed_launch = #name; return _le; }                  \
        if (ev) { const hipError_t _ee = hipEventRecord(ev[nev++], stream); if (_ee != hipSuccess) return _ee; } \
    } while (0)

// Launch sequence of one batch solve.  The IMU / prior linearisation of a round does not depend on the projection factors:
// with `aux` != nullptr it is forked onto that second stream (event fork -> aux; join before the solve kernel), so the two
// linearisation kernels run side by side.  ev != nullptr (profiling): everything stays on `stream`, an event is recorded
// after every launch (ev[0] before the first), kinds[i] = kernel class of the launch that ends at ev[i + 1] (0 prologue,
// 1 linearize, 2 accumulate, 3 solve, 4 final); the caller provides 4 * rounds + 5 events.
struct BaFork { hipStream_t aux; hipEvent_t fork, join; };
// (A merged kernel -- factor phases and solve phases of a round in ONE launch -- was measured in round 4 and lost: 280 us per round
//  against 103 + 149 us, profiles/r04h_bench_{merged,two}.json; it is gone since the solve kernel runs 4 wavefronts per window.)
extern "C" hipError_t ba_launch_solve(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, int rounds, hipStream_t stream,
                                      hipEvent_t* ev, int* kinds, int* n_launches, const BaFork* fk) {
    hipError_t e = set_lds_attrs();
    if (e != hipSuccess) { g_failed_launch = "hipFuncSetAttribute"; return e; }
    int nev = 0;
    if (ev) { e = hipEventRecord(ev[nev++], stream); if (e != hipSuccess) return e; }
    const bool forked = fk && fk->aux && !ev && !L.la_on;
    int nk = 0;
    LAUNCH(ba_prologue_kernel, dim3(L.nwin, L.pro_split ? 1 + (L.K - 1 + BA_NW - 1) / BA_NW + 2 : 1), dim3(BA_NT), L.lds_pro, dL, P);
    if (kinds) kinds[nk++] = 0;
    for (int r = 0; r <= rounds; ++r) {
        const int cost_only = r == rounds ? 1 : 0;
        const bool fused = L.la_on && !cost_only;         // IMU + prior + projection factors linearised AND accumulated by one kernel
        if (fused) {
            // (nothing: ba_linacc_proj_kernel below does it)
        } else if (forked) {
            if ((e = hipEventRecord(fk->fork, stream)) != hipSuccess) return e;
            if ((e = hipStreamWaitEvent(fk->aux, fk->fork, 0)) != hipSuccess) return e;
            hipLaunchKernelGGL(ba_linearize_imu_kernel, dim3(L.nig + L.nprw, L.nwin), dim3(BA_NT), L.lds_lin, fk->aux, dL, P, cost_only);
            if ((e = hipGetLastError()) != hipSuccess) { g_failed_launch = "ba_linearize_imu_kernel"; return e; }
            if ((e = hipEventRecord(fk->join, fk->aux)) != hipSuccess) return e;
        } else {
            LAUNCH(ba_linearize_imu_kernel, dim3(L.nig + L.nprw, L.nwin), dim3(BA_NT), L.lds_lin, dL, P, cost_only);
        }
        if (fused) LAUNCH(ba_linacc_proj_kernel, dim3(L.nwin), dim3(LA_NT), L.lds_linacc, dL, P);
        else LAUNCH(ba_linearize_proj_kernel, dim3(L.nbf, L.nwin), dim3(BA_LIN_NT), 0, dL, P, cost_only);
        if (kinds) { if (!forked && !fused) kinds[nk++] = 1; kinds[nk++] = fused ? 2 : 1; }
        if (r == rounds) {
            if (forked && (e = hipStreamWaitEvent(stream, fk->join, 0)) != hipSuccess) return e;
            break;
        }
        if (!fused) LAUNCH(ba_accumulate_kernel, dim3(L.nba * ((L.nwin + 7) / 8) * 8), dim3(BA_ACC_NT), 0, dL, P);
        if (forked && (e = hipStreamWaitEvent(stream, fk->join, 0)) != hipSuccess) return e;
        if (L.sv_w8) LAUNCH(ba_solve_w8_kernel, dim3(L.nwin), dim3(2 * SV_NT), L.lds_solve_w8, dL, P);     // few windows: 8 wavefronts per window
        else LAUNCH(ba_solve_kernel, dim3(L.nwin), dim3(SV_NT), L.lds_solve, dL, P);
        if (kinds) { if (!fused) kinds[nk++] = 2; kinds[nk++] = 3; }
    }
    LAUNCH(ba_final_kernel, dim3(L.nwin), dim3(256), 0, dL, P);
    if (kinds) kinds[nk++] = 4;
    if (n_launches) *n_launches = nk;
    return hipSuccess;
}

// Launch sequence of the large-window path.  `allreduce` (may be nullptr: single rank) is called twice per round between
// launches with a device buffer that has to be summed over the ranks IN PLACE, stream-ordered on `stream` (RCCL:
// ncclAllReduce(buf, buf, count, ncclDouble, ncclSum, comm, stream)); a non-zero return aborts the sequence.
// `rounds` = max_iters + slack: a failed factorisation retries its iteration in the next round (see above).
typedef int (*BaAllReduce)(void* user, double* buf, size_t count, void* stream);
extern "C" hipError_t ba_launch_solve_big(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, int rounds, int slack, hipStream_t stream,
                                          BaAllReduce allreduce, void* user, int* hook_rc, hipEvent_t* ev, int* kinds, int* n_launches) {
    hipError_t e = set_lds_attrs();
    if (e != hipSuccess) { g_failed_launch = "hipFuncSetAttribute"; return e; }
    int nev = 0, nk = 0;
    if (ev) { e = hipEventRecord(ev[nev++], stream); if (e != hipSuccess) return e; }
    if (hook_rc) *hook_rc = 0;
    const size_t n1 = (size_t)L.nwin * L.rb1_len, n2 = (size_t)L.nwin * RB2_LEN;
#define REDUCE(buf, n)                                                                   \
    do {                                                                                 \
        if (allreduce) {                                                                 \
            const int _rc = allreduce(user, buf, n, (void*)stream);                      \
            if (_rc) { if (hook_rc) *hook_rc = _rc; g_failed_launch = "all-reduce hook"; return hipErrorInvalidValue; } \
        }                                                                                \
    } while (0)
#define KIND(k) do { if (kinds) kinds[nk++] = (k); } while (0)
    // (profiling: the all-reduce sits in the gap before the kernel that consumes it and is counted with that kernel)
    LAUNCH(ba_prologue_kernel, dim3(L.nwin, L.pro_split ? 1 + (L.K - 1 + BA_NW - 1) / BA_NW + 2 : 1), dim3(BA_NT), L.lds_pro, dL, P); KIND(0);
    // `rounds` = max_iters + slack: a failed factorisation retries in the NEXT round (the new T needs the collective), so a solve
    // can need more rounds than iterations.  Usually it does not: once the nominal number of rounds has been issued the host
    // looks at the windows' DONE flags (one strided 8-byte-per-window copy + a stream synchronisation) before every further
    // round and stops issuing when every window has finished -- rounds of kernels that return at their first instruction and, on
    // several ranks, two all-reduces each.  The flags are functions of the reduced (rank-identical) data: every rank stops at the
    // same round.
    std::vector<double> done_flags;
    for (int r = 0; r <= rounds; ++r) {
        const int cost_only = r == rounds ? 1 : 0;
        if (slack > 0 && r >= rounds - slack) {
            done_flags.assign(L.nwin, 0.0);
            e = hipMemcpy2DAsync(done_flags.data(), sizeof(double), P.scr + L.so_ctl + C_DONE, (size_t)L.sstride * sizeof(double), sizeof(double),
                                 L.nwin, hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) { g_failed_launch = "read-back of the DONE flags"; return e; }
            bool all_done = true;
            for (double v : done_flags) all_done = all_done && v != 0.0;
            if (all_done) break;             // (the cost-only pass would return at its first instruction too)
        }
        LAUNCH(ba_linearize_imu_kernel, dim3(L.nig + L.nprw, L.nwin), dim3(BA_NT), L.lds_lin, dL, P, cost_only); KIND(1);
        LAUNCH(ba_linearize_proj_kernel, dim3(L.nbf, L.nwin), dim3(BA_LIN_NT), 0, dL, P, cost_only); KIND(1);
        if (!cost_only) { LAUNCH(ba_accumulate_kernel, dim3(L.nba * ((L.nwin + 7) / 8) * 8), dim3(BA_ACC_NT), 0, dL, P); KIND(2); }
        LAUNCH(ba_big_schur_kernel, dim3(L.nts + 1 + BA_BIG_ZERO_BLOCKS, L.nwin), dim3(256), 0, dL, P, cost_only); KIND(6);
        REDUCE(P.rb1, n1);
        if (cost_only) break;
        LAUNCH(ba_solve_big_kernel, dim3(L.nwin), dim3(BA_NT), L.lds_solve, dL, P); KIND(7);
        { const int nt16 = L.RcPad / 16; LAUNCH(ba_big_chain_schur_kernel, dim3(nt16 * (nt16 + 1) / 2, L.nwin), dim3(64), 0, dL, P); KIND(7 | 0x100); }
        LAUNCH(ba_solve_big_tail_kernel, dim3(L.nwin), dim3(BA_NT), L.lds_solve, dL, P); KIND(7 | 0x100);
        LAUNCH(ba_big_landmark_kernel, dim3(BA_BIG_LM_BLOCKS, L.nwin), dim3(256), 0, dL, P); KIND(8);
        REDUCE(P.rb2, n2);
        LAUNCH(ba_big_step_kernel, dim3(L.nwin), dim3(BA_NT), 0, dL, P); KIND(8);
    }
    LAUNCH(ba_final_kernel, dim3(L.nwin), dim3(256), 0, dL, P); KIND(4);
#undef REDUCE
#undef KIND
    if (n_launches) *n_launches = nk;
    return hipSuccess;
}

extern "C" hipError_t ba_launch_eval_factors(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, double* proj_r, double* proj_J,
                                            double* imu_r, double* imu_J, double* prior_r, hipStream_t stream) {
    hipError_t e = set_lds_attrs();
    if (e != hipSuccess) { g_failed_launch = "hipFuncSetAttribute"; return e; }
    hipEvent_t* ev = nullptr;
    int nev = 0;
    (void)nev;
    LAUNCH(ba_prologue_kernel, dim3(1), dim3(BA_NT), L.lds_pro, dL, P);
    LAUNCH(ba_eval_factors_kernel, dim3(1), dim3(BA_NT), L.lds_lin, dL, P, proj_r, proj_J, imu_r, imu_J, prior_r);
    return hipSuccess;
}
#endif   // BA_SOLVE_W8_TU
