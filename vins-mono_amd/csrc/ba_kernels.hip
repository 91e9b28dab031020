// ba_kernels.hip — sliding-window bundle adjustment on gfx950 (CDNA4), one workgroup per window.
//
// Replaces the arithmetic under Estimator::optimization() (vins_estimator/src/estimator.cpp:670-1003):
// factor evaluation (factor/*.cpp,h), the Ceres trust-region solve it configures at :803-818
// (DENSE_SCHUR + DOGLEG + Jacobi scaling + Cauchy loss corrector; third-party behaviour restated in
// oracle/ASSUMPTIONS.md) and the gauge fix of double2vector() (:530-619).
//
// Design: the whole 8-iteration solve runs inside ONE launch; the reduced camera
// system S (R = 6Kp + 9K [+6 +1] <= 184) lives in LDS as a packed lower triangle for its entire life
// (assembly -> Jacobi scaling -> landmark Schur complement -> Cholesky -> back-substitution); the
// per-factor Jacobian records of the current chunk are staged in the part of LDS that S's speed-bias
// rows will occupy later, and "owner" lanes (one wavefront per 6x6 block, one lane per entry) sum
// them into S without atomics, so the result is bit-reproducible.  The landmark Schur complement
// S -= W diag(1/h) W^T is the only GEMM-shaped step and runs on v_mfma_f64_16x16x4_f64.
#include <hip/hip_runtime.h>
#include "ba_layout.h"
#include "ba_factors.h"
#include "../../include/vinsgpu.h"

#define NOINL __device__ __noinline__
extern __shared__ __attribute__((aligned(16))) char ba_smem[];
#define LDSB ((double*)ba_smem)
typedef double double4_t __attribute__((vector_size(32)));      // v_mfma_f64_16x16x4 accumulator (4 VGPR pairs)

// optional phase timers (build with -DBA_PROFILE): lane 0 accumulates s_memtime deltas per phase
#ifdef BA_PROFILE
#define PROF_DECL long long _pt = 0
#define PROF_T0() do { if (c.tid == 0) _pt = clock64(); } while (0)
#define PROF_ADD(id) do { if (c.tid == 0) { const long long _n = clock64(); LDSB[c.Lp->l_misc + (id)] += (double)(_n - _pt); _pt = _n; } } while (0)
#else
#define PROF_DECL
#define PROF_T0()
#define PROF_ADD(id)
#endif
enum { PF_PRO = 0, PF_IMU, PF_PRIOR, PF_PROJ, PF_ACC, PF_LMACC, PF_IMUACC, PF_PRACC, PF_JVEC, PF_BUILD, PF_SCHUR, PF_CHOL, PF_BACK, PF_CAND, PF_MISC, PF_TOTAL };

// R-vectors kept in LDS
enum { V_G = 0, V_SC, V_DG, V_GT, V_GN, V_U, V_Y, V_T, V_DI, V_NVEC };

// ------------------------------------------------------------------------------------------------
struct Ctx {
    const BaLayout* Lp;      // layout lives in device memory: uniform scalar loads on demand, no SGPR hoarding
    const int* hdr;
    const int* ia;       // int arrays of this window
    const double* di;    // double inputs
    double* sc;          // scratch
    int tid, lane, wave;
    int nL, nF, nprior, nblk, nchunk;
    double focal, tr, row, gnorm;
};

DEV double wave_sum(double v) { return wave_sum_all(v); }     // DPP network, result in every lane (ba_math.h)
// deterministic block-wide sum, result uniform in every thread (2 barriers)
DEV double block_sum(const Ctx& c, double v) {
    double* red = LDSB + c.Lp->l_red;
    v = wave_sum(v);
    __syncthreads();
    if (c.lane == 0) red[c.wave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < BA_NW; ++w) s += red[w];
    return s;
}
DEV void block_sum2(const Ctx& c, double& a, double& b) {
    double* red = LDSB + c.Lp->l_red;
    a = wave_sum(a);
    b = wave_sum(b);
    __syncthreads();
    if (c.lane == 0) { red[c.wave] = a; red[BA_NW + c.wave] = b; }
    __syncthreads();
    double s = 0.0, t = 0.0;
#pragma unroll
    for (int w = 0; w < BA_NW; ++w) { s += red[w]; t += red[BA_NW + w]; }
    a = s; b = t;
}
DEV double block_max(const Ctx& c, double v) {
    double* red = LDSB + c.Lp->l_red;
    v = wave_max_all(v);
    __syncthreads();
    if (c.lane == 0) red[c.wave] = v;
    __syncthreads();
    double s = red[0];
#pragma unroll
    for (int w = 1; w < BA_NW; ++w) s = fmax(s, red[w]);
    return s;
}

DEV int col_pose(const BaLayout& L, int i) { return 6 * i; }
DEV int col_ex(const BaLayout& L) { return 6 * L.Kp; }
DEV int col_td(const BaLayout& L) { return 6 * L.Kp + 6 * L.e; }
DEV int col_sb(const BaLayout& L, int i) { return L.Rc + 9 * i; }
DEV int tri(int i, int j) { return i * (i + 1) / 2 + j; }   // packed lower, j <= i

// state block in LDS: [pose Kp*7 | sb K*9 | ex 7 | td 1]
DEV const double* st_pose(const BaLayout& L, const double* x, int i) { return x + 7 * i; }
DEV const double* st_sb(const BaLayout& L, const double* x, int i) { return x + 7 * L.Kp + 9 * i; }
DEV const double* st_ex(const BaLayout& L, const double* x) { return x + 7 * L.Kp + 9 * L.K; }
DEV int st_size(const BaLayout& L) { return 7 * L.Kp + 9 * L.K + 8; }

// ================================================================================================
// IMU: sqrt_info = U^-1 where covariance = U U^T (U upper) — this is exactly
// LLT(covariance^-1).matrixL().transpose() of imu_factor.h:64 (the Cholesky factor of the inverse is
// unique) but never forms the badly conditioned inverse.  One wavefront per factor, matrix in LDS.
// ================================================================================================
NOINL void imu_sqrt_info(const Ctx& c) {
    const BaLayout& L = *c.Lp;
    double* A = LDSB + L.l_stage + c.wave * 256;     // 15x15 scratch per wave (stage region is free)
    const int nimu = L.K - 1;
    const int* valid = c.ia + L.io_imu_valid;
    for (int base = 0; base < nimu; base += BA_NW) {
        const int f = base + c.wave;
        const bool act = f < nimu && valid[f];
        const double* cov = c.di + L.do_imu + f * BA_IMU_STRIDE + IM_COV;
        if (act)
            for (int k = c.lane; k < 225; k += 64) A[k] = cov[k];
        __syncthreads();
        // UL factorisation, columns from the last to the first: A = U U^T
        for (int j = 14; j >= 0; --j) {
            const double d = act ? sqrt(A[j * 15 + j]) : 1.0;
            __syncthreads();
            if (act && c.lane < j) A[c.lane * 15 + j] /= d;
            if (act && c.lane == j) A[j * 15 + j] = d;
            __syncthreads();
            if (act)
                for (int k = c.lane; k < j * j; k += 64) {
                    const int i = k / j, kk = k % j;
                    if (kk >= i) A[i * 15 + kk] -= A[i * 15 + j] * A[kk * 15 + j];
                }
            __syncthreads();
        }
        // X = U^-1 (upper): lane = column j, back-substitute upward
        double* Uo = c.sc + L.so_imuU + f * 225;
        if (act && c.lane < 15) {
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
        __syncthreads();
    }
}

// All IMU factors at state x, one (half-)wavefront per factor, all factors in one round:
//   lanes 0..29 = Jacobian columns, lane 30 = the residual "column"; every lane evaluates the (cheap) factor context,
//   weights its column with the upper-triangular U = sqrt_info read from an LDS copy, parks it in LDS, and the
//   wavefront then forms the factor's 30x30 Hessian block and J^T r from LDS — so the later accumulation into S is
//   pure adds.  The LDS used here (U copy + one 15x32 panel per wavefront) is the S / staging area, idle in this phase.
// JAC = false: residual only (candidate cost).  Returns this thread's share of sum r^2.
//   global scratch: so_imuR [f][15] weighted residual (JAC only), so_imuJ [f][512]: 465 lower Hessian entries + 30 g.
template <bool JAC>
NOINL double imu_pass(const Ctx& c, const double* x) {
    const BaLayout& L = *c.Lp;
    const int nimu = L.K - 1;
    const int* valid = c.ia + L.io_imu_valid;
    double* Us = LDSB + L.l_S;                              // [nimu][225]
    double* panels = Us + ((nimu * 225 + 1) & ~1);          // [nimu][15][32]
    double cost = 0.0;
    __syncthreads();
    for (int k = c.tid; k < nimu * 225; k += BA_NT) Us[k] = c.sc[L.so_imuU + k];
    __syncthreads();
    // phase A: one HALF-wavefront per factor when there are more factors than wavefronts (lanes 0-30 / 32-62), so all
    // factors are evaluated in one round; phase B: the Hessian entries of all factors are spread over all threads.
    const int per = nimu > BA_NW ? 2 : 1;
    const int half = c.lane >> 5, hl = c.lane & 31;
    const int f = c.wave * per + half;
    const bool act = f < nimu && half < per && valid[f];
    if (act) {
        const double* pre = c.di + L.do_imu + f * BA_IMU_STRIDE;
        const double* U = Us + f * 225;
        double* panel = panels + f * 480;
        ImuCtx ic;
        imu_ctx<JAC>(pre, st_pose(L, x, f), st_sb(L, x, f), st_pose(L, x, f + 1), st_sb(L, x, f + 1), c.gnorm, ic);
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
                if (hl == 30) { cost += s * s; if (JAC) c.sc[L.so_imuR + f * 15 + r] = s; }
            }
        }
    }
    __syncthreads();
    if (JAC) {
        for (int w = c.tid; w < nimu * 495; w += BA_NT) {
            const int ff = w / 495, e = w - 495 * ff;
            if (!valid[ff]) continue;
            const double* panel = panels + ff * 480;
            int a, b;
            if (e < 465) tri_decode(e, a, b);
            else { a = e - 465; b = 30; }
            double s = 0.0;
#pragma unroll
            for (int r = 0; r < 15; ++r) s += panel[r * 32 + a] * panel[r * 32 + b];
            c.sc[L.so_imuJ + ff * 512 + e] = s;
        }
    }
    __syncthreads();
    return cost;
}

// local column (0..29) of IMU factor f -> reduced column
DEV int imu_col(const BaLayout& L, int f, int lc) {
    if (lc < 6) return col_pose(L, f) + lc;
    if (lc < 15) return col_sb(L, f) + lc - 6;
    if (lc < 21) return col_pose(L, f + 1) + lc - 15;
    return col_sb(L, f + 1) + lc - 21;
}

// ================================================================================================
// Prior (MarginalizationFactor::Evaluate, marginalization_factor.cpp:333-381)
// ================================================================================================
DEV const double* state_block(const Ctx& c, const double* x, int kind, int idx) {
    const BaLayout& L = *c.Lp;
    if (kind == VG_BLK_POSE) return st_pose(L, x, idx);
    if (kind == VG_BLK_SPEEDBIAS) return st_sb(L, x, idx);
    if (kind == VG_BLK_EXPOSE) return st_ex(L, x);
    return st_ex(L, x) + 7;     // td
}
// dx into scratch so_pu (n doubles); then r = r0 + J0 dx into so_pr.  Returns share of sum r^2.
NOINL double prior_pass(const Ctx& c, const double* x, double* pr) {
    const BaLayout& L = *c.Lp;
    if (c.nprior == 0) return 0.0;
    double* dx = c.sc + L.so_pu;
    const int* kind = c.ia + L.io_pb_kind;
    const int* idx = c.ia + L.io_pb_idx;
    const int* off = c.ia + L.io_pb_off;
    const int* x0off = c.ia + L.io_pb_x0off;
    __syncthreads();
    for (int b = c.tid; b < c.nblk; b += BA_NT) {
        const double* xb = state_block(c, x, kind[b], idx[b]);
        const double* x0 = c.di + L.do_px0 + x0off[b];
        double* d = dx + off[b];
        if (kind[b] == VG_BLK_SPEEDBIAS) {
#pragma unroll
            for (int k = 0; k < 9; ++k) d[k] = xb[k] - x0[k];
        } else if (kind[b] == VG_BLK_TD) {
            d[0] = xb[0] - x0[0];
        } else {
            d[0] = xb[0] - x0[0]; d[1] = xb[1] - x0[1]; d[2] = xb[2] - x0[2];
            double qi[4], dq[4];
            q_inv(x0 + 3, qi);
            q_mul(qi, xb + 3, dq);
            const double sgn = (dq[3] >= 0) ? 2.0 : -2.0;
            d[3] = sgn * dq[0]; d[4] = sgn * dq[1]; d[5] = sgn * dq[2];
        }
    }
    __syncthreads();
    const int n = c.nprior;
    const double* J0t = c.di + L.do_pJ0t;     // J0t[c*Ncap + r] = J0[r][c]  (coalesced over r)
    const double* r0 = c.di + L.do_pr0;
    double* part = LDSB + L.l_wd;            // 4 x Ncap partial sums (the Schur tile is idle here)
    double cost = 0.0;
    {
        // r = r0 + J0 dx, the n-term dot product of every row split over 4 threads
        const int r = c.tid % L.Ncap, q = c.tid / L.Ncap;
        if (q < 4 && r < n) {
            double s = 0.0;
            for (int k = q; k < n; k += 4) s += J0t[k * L.Ncap + r] * dx[k];
            part[q * L.Ncap + r] = s;
        }
    }
    __syncthreads();
    for (int r = c.tid; r < n; r += BA_NT) {
        const double s = r0[r] + ((part[r] + part[L.Ncap + r]) + (part[2 * L.Ncap + r] + part[3 * L.Ncap + r]));
        pr[r] = s;
        cost += s * s;
    }
    return cost;
}

// ================================================================================================
// Projection factors
// ================================================================================================
struct ProjIn {
    const double *pi, *pj, *oi, *oj;
    double lam;
    int i, j, l;
};
DEV void proj_fetch(const Ctx& c, int f, const double* x, const double* lam, ProjIn& p) {
    const BaLayout& L = *c.Lp;
    p.i = c.ia[L.io_fac_i + f];
    p.j = c.ia[L.io_fac_j + f];
    p.l = c.ia[L.io_fac_lm + f];
    p.pi = st_pose(L, x, p.i);
    p.pj = st_pose(L, x, p.j);
    p.oi = c.di + L.do_obs + c.ia[L.io_fac_oi + f] * BA_OBS_STRIDE;
    p.oj = c.di + L.do_obs + c.ia[L.io_fac_oj + f] * BA_OBS_STRIDE;
    p.lam = lam[p.l];
}

// cost-only pass: returns this thread's share of sum rho(|r|^2)  (CauchyLoss(1.0): rho = log(1+s))
NOINL double proj_cost_pass(const Ctx& c, const double* x, const double* lam) {
    const BaLayout& L = *c.Lp;
    const double* ex = st_ex(L, x);
    double cost = 0.0;
    for (int f = c.tid; f < c.nF; f += BA_NT) {
        ProjIn p;
        proj_fetch(c, f, x, lam, p);
        double r[2];
        if (L.t) proj_eval<true, false, false>(p.pi, p.pj, ex, p.lam, p.oi, p.oj, ex[7], c.focal, c.tr, c.row, r, 0, 0, 0, 0, 0);
        else proj_eval<false, false, false>(p.pi, p.pj, ex, p.lam, p.oi, p.oj, 0.0, c.focal, c.tr, c.row, r, 0, 0, 0, 0, 0);
        cost += log1p(r[0] * r[0] + r[1] * r[1]);
    }
    return cost;
}

DEV void proj_jac(const Ctx& c, const ProjIn& p, const double* ex, double* r, double* Ji, double* Jj, double* Jex,
                  double* Jl, double* Jtd) {
    const BaLayout& L = *c.Lp;
    if (L.t) {
        if (L.e) proj_eval<true, true, true>(p.pi, p.pj, ex, p.lam, p.oi, p.oj, ex[7], c.focal, c.tr, c.row, r, Ji, Jj, Jex, Jl, Jtd);
        else proj_eval<true, true, false>(p.pi, p.pj, ex, p.lam, p.oi, p.oj, ex[7], c.focal, c.tr, c.row, r, Ji, Jj, Jex, Jl, Jtd);
    } else {
        if (L.e) proj_eval<false, true, true>(p.pi, p.pj, ex, p.lam, p.oi, p.oj, 0.0, c.focal, c.tr, c.row, r, Ji, Jj, Jex, Jl, Jtd);
        else proj_eval<false, true, false>(p.pi, p.pj, ex, p.lam, p.oi, p.oj, 0.0, c.focal, c.tr, c.row, r, Ji, Jj, Jex, Jl, Jtd);
    }
}

// ---- linearisation of one chunk: thread per factor -> loss-corrected record in LDS staging
// record (REC doubles): [0..11] Ji as (row0,row1) pairs per column | [12..23] Jj | [24,25] Jl | [26,27] r |
//                       [28..39] Jex | [40,41] Jtd
NOINL double proj_linearize_chunk(const Ctx& c, int ch, const double* x, const double* lam) {
    const BaLayout& L = *c.Lp;
    const double* ex = st_ex(L, x);
    const int fb = c.ia[L.io_chunk_fbeg + ch], fe = c.ia[L.io_chunk_fbeg + ch + 1];
    double* stage = LDSB + L.l_stage;
    double cost = 0.0;
    for (int f = fb + c.tid; f < fe; f += BA_NT) {
        ProjIn p;
        proj_fetch(c, f, x, lam, p);
        double r[2], Ji[12], Jj[12], Jex[12], Jl[2], Jtd[2];
        proj_jac(c, p, ex, r, Ji, Jj, Jex, Jl, Jtd);
        const double s = r[0] * r[0] + r[1] * r[1];
        cost += log1p(s);
        const double sq = sqrt(1.0 / (1.0 + s));
        double* rec = stage + c.ia[L.io_fac_slot + f] * L.REC;
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
    }
    return cost;
}

// sum over slots [b,e) of  rec[offA..+1] . rec[offB..+1]
DEV double seg_dot(const double* stage, int REC, int b, int e, int offA, int offB) {
    double acc = 0.0;
    for (int s = b; s < e; ++s) {
        const double2 a2 = *(const double2*)(stage + s * REC + offA);
        const double2 b2 = *(const double2*)(stage + s * REC + offB);
        acc += a2.x * b2.x + a2.y * b2.y;
    }
    return acc;
}

// Owner accumulation of the camera part of S and g from the staged chunk: one wavefront per block
// task, lane = entry (p,q) of the (<=6)x(<=6) block; lanes 36..41 of diagonal tasks own the gradient.
NOINL void proj_accumulate_chunk(const Ctx& c, int ch) {
    const BaLayout& L = *c.Lp;
    const int Kp = L.Kp, REC = L.REC;
    const double* stage = LDSB + L.l_stage;
    double* S = LDSB + L.l_S;
    double* g = LDSB + L.l_vec + V_G * L.Rpad;
    const int* ptr = c.ia + L.io_pair_ptr + ch * (Kp * Kp + 1);
    const int nb = Kp + L.e + L.t;                 // block rows: poses, ex, td
    const int ntask = nb * (nb + 1) / 2;
    const int offEx = 28, offTd = 28 + 12 * L.e;
    const int nslots = ptr[Kp * Kp];
    for (int task = c.wave; task < ntask; task += BA_NW) {
        // task -> (br >= bc)
        int br, bc;
        tri_decode(task, br, bc);
        const int kr = br < Kp ? 0 : (br == Kp && L.e ? 1 : 2);     // 0 pose, 1 ex, 2 td
        const int kc = bc < Kp ? 0 : (bc == Kp && L.e ? 1 : 2);
        const int dr = kr == 2 ? 1 : 6, dc = kc == 2 ? 1 : 6;
        const int rowbase = kr == 0 ? 6 * br : (kr == 1 ? col_ex(L) : col_td(L));
        const int colbase = kc == 0 ? 6 * bc : (kc == 1 ? col_ex(L) : col_td(L));
        const bool diag = br == bc;
        if (kr == 0 && kc == 0 && diag) {
            // diagonal pose block: it visits every factor anchored at or targeting frame a (~10x the visits of an
            // off-diagonal block), so its 21 lower-triangle entries + 6 gradient entries are spread over TWO lane
            // segments (lanes 0-26 / 27-53) that each take half of every slot range; fixed split -> deterministic
            const int a = br;
            const int seg = c.lane >= 27 ? 1 : 0, e = c.lane - 27 * seg;
            const bool on = c.lane < 54;
            const bool isg2 = e >= 21;
            int p2 = 0, q2 = 0;
            if (!isg2) tri_decode(e, p2, q2); else p2 = e - 21;
            double part = 0.0;
            if (on) {
                const int oB_i = isg2 ? 26 : 2 * q2, oB_j = isg2 ? 26 : 12 + 2 * q2;
                {
                    const int s0 = ptr[a * Kp], s1 = ptr[(a + 1) * Kp], mid = (s0 + s1) >> 1;
                    part += seg_dot(stage, REC, seg ? mid : s0, seg ? s1 : mid, 2 * p2, oB_i);
                }
                for (int a2 = 0; a2 < a; ++a2) {
                    const int s0 = ptr[a2 * Kp + a], s1 = ptr[a2 * Kp + a + 1], mid = (s0 + s1) >> 1;
                    part += seg_dot(stage, REC, seg ? mid : s0, seg ? s1 : mid, 12 + 2 * p2, oB_j);
                }
            }
            const double other = __shfl_down(part, 27, 64);
            if (c.lane < 27) {
                const double tot = part + other;
                if (isg2) g[rowbase + p2] += tot;
                else S[tri(rowbase + p2, colbase + q2)] += tot;
            }
            continue;
        }
        const bool isg = diag && c.lane >= 36 && c.lane < 36 + dr;
        const int p = isg ? c.lane - 36 : c.lane / 6, q = isg ? 0 : c.lane % 6;
        const bool act = isg || (c.lane < 36 && p < dr && q < dc && (!diag || q <= p));
        if (!act) continue;
        double acc = 0.0;
        // role offsets of the row / column block inside a record, as anchor (i) or target (j)
        if (kr == 0 && kc == 0) {
            // row block br = target j, column block bc = anchor i
            acc += seg_dot(stage, REC, ptr[bc * Kp + br], ptr[bc * Kp + br + 1], 12 + 2 * p, 2 * q);
        } else {
            const int oA = (kr == 1 ? offEx : offTd) + 2 * p;
            if (kc == 0) {
                const int a = bc;
                acc += seg_dot(stage, REC, ptr[a * Kp], ptr[(a + 1) * Kp], oA, 2 * q);
                for (int a2 = 0; a2 < a; ++a2)
                    acc += seg_dot(stage, REC, ptr[a2 * Kp + a], ptr[a2 * Kp + a + 1], oA, 12 + 2 * q);
            } else {
                const int oB = isg ? 26 : (kc == 1 ? offEx : offTd) + 2 * q;
                acc += seg_dot(stage, REC, 0, nslots, oA, oB);
            }
        }
        if (isg) g[rowbase + p] += acc;
        else S[tri(rowbase + p, colbase + q)] += acc;
    }
}

// per-landmark sums from the staged chunk: h = sum Jl.Jl, b = sum Jl.r, W column -> Wt[col][l]
NOINL void landmark_accumulate_chunk(const Ctx& c, int ch) {
    const BaLayout& L = *c.Lp;
    const int REC = L.REC;
    const double* stage = LDSB + L.l_stage;
    const int lb = c.ia[L.io_chunk_lbeg + ch], le = c.ia[L.io_chunk_lbeg + ch + 1];
    double* Wt = c.sc + L.so_Wt;
    for (int l = lb + c.tid; l < le; l += BA_NT) {
        const int fb = c.ia[L.io_lm_fbeg + l], fe = c.ia[L.io_lm_fbeg + l + 1];
        double h = 0.0, b = 0.0, wi[6] = {0, 0, 0, 0, 0, 0}, wex[6] = {0, 0, 0, 0, 0, 0}, wtd = 0.0;
        int anchor = -1;
        for (int f = fb; f < fe; ++f) {
            const double* rec = stage + c.ia[L.io_fac_slot + f] * REC;
            const double l0 = rec[24], l1 = rec[25];
            h += l0 * l0 + l1 * l1;
            b += l0 * rec[26] + l1 * rec[27];
            anchor = c.ia[L.io_fac_i + f];
            const int j = c.ia[L.io_fac_j + f];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                wi[k] += rec[2 * k] * l0 + rec[2 * k + 1] * l1;
                Wt[(col_pose(L, j) + k) * L.Lcap + l] = rec[12 + 2 * k] * l0 + rec[12 + 2 * k + 1] * l1;
            }
            if (L.e) {
#pragma unroll
                for (int k = 0; k < 6; ++k) wex[k] += rec[28 + 2 * k] * l0 + rec[28 + 2 * k + 1] * l1;
            }
            if (L.t) wtd += rec[28 + 12 * L.e] * l0 + rec[29 + 12 * L.e] * l1;
        }
        if (anchor >= 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) Wt[(col_pose(L, anchor) + k) * L.Lcap + l] = wi[k];
            if (L.e) {
#pragma unroll
                for (int k = 0; k < 6; ++k) Wt[(col_ex(L) + k) * L.Lcap + l] = wex[k];
            }
            if (L.t) Wt[col_td(L) * L.Lcap + l] = wtd;
        }
        c.sc[L.so_h + l] = h;
        c.sc[L.so_b + l] = b;
    }
}

// ================================================================================================
// Linearisation at x: fills S (unscaled, pre-Schur), g, h, b, Wt; returns the cost (uniform).
// ================================================================================================
DEV double linearize(const Ctx& c, const double* x, const double* lam) {
    const BaLayout& L = *c.Lp;
    double* S = LDSB + L.l_S;
    double* g = LDSB + L.l_vec + V_G * L.Rpad;
    const int R = L.R, Rc = L.Rc;
    const int camtri = Rc * (Rc + 1) / 2;
    const int fulltri = (R + 1) * (R + 2) / 2;
    __syncthreads();
    PROF_DECL;
    PROF_T0();
    double cost2 = imu_pass<true>(c, x);            // sum r^2 share (IMU + prior, no loss)
    PROF_ADD(PF_IMU);
    cost2 += prior_pass(c, x, c.sc + L.so_pr);
    PROF_ADD(PF_PRIOR);
    double costrho = 0.0;                           // sum rho share (projection)
    for (int k = c.tid; k < camtri; k += BA_NT) S[k] = 0.0;
    for (int k = c.tid; k < L.Rpad; k += BA_NT) g[k] = 0.0;
    {
        double* Wt = c.sc + L.so_Wt;
        const int n = L.RcPad * L.Lcap;
        for (int k = c.tid; k < n; k += BA_NT) Wt[k] = 0.0;
    }
    __syncthreads();
    for (int ch = 0; ch < c.nchunk; ++ch) {
        costrho += proj_linearize_chunk(c, ch, x, lam);
        __syncthreads();
        PROF_ADD(PF_PROJ);
        proj_accumulate_chunk(c, ch);
        __syncthreads();
        PROF_ADD(PF_ACC);
        landmark_accumulate_chunk(c, ch);
        __syncthreads();
        PROF_ADD(PF_LMACC);
    }
    for (int k = camtri + c.tid; k < fulltri; k += BA_NT) S[k] = 0.0;
    __syncthreads();
    // ---- IMU Hessian blocks (precomputed by imu_pass): factors k and k+1 share the blocks of frame k+1, so even and
    //      odd factors are added in two rounds (inside a round every S entry has exactly one writer)
    {
        const int nimu = L.K - 1;
        const int* valid = c.ia + L.io_imu_valid;
        for (int par = 0; par < 2; ++par) {
            const int nf = (nimu - par + 1) / 2;
            for (int w = c.tid; w < nf * 512; w += BA_NT) {
                const int f = 2 * (w >> 9) + par, e = w & 511;
                if (e >= 495 || !valid[f]) continue;
                const double v = c.sc[L.so_imuJ + f * 512 + e];
                if (e < 465) {
                    int a, b;
                    tri_decode(e, a, b);
                    const int ca = imu_col(L, f, a), cb = imu_col(L, f, b);
                    S[ca >= cb ? tri(ca, cb) : tri(cb, ca)] += v;
                } else {
                    g[imu_col(L, f, e - 465)] += v;
                }
            }
            __syncthreads();
        }
    }
    PROF_ADD(PF_IMUACC);
    // ---- prior: S += J0^T J0 (precomputed Hp), g += J0^T r   (pmap: prior column -> reduced column or -1)
    if (c.nprior) {
        const int n = c.nprior;
        const double* Hp = c.sc + L.so_Hp;
        const double* J0 = c.di + L.do_pJ0;       // row-major: J0[r*Ncap + c] coalesced over c
        const double* pr = c.sc + L.so_pr;
        const int* pmap = (const int*)(LDSB + L.l_pmap);
        for (int w = c.tid; w < n * (n + 1) / 2 + n; w += BA_NT) {
            const bool isg = w >= n * (n + 1) / 2;
            if (isg) {
                const int a = w - n * (n + 1) / 2;
                const int ca = pmap[a];
                if (ca >= 0) {
                    double s0 = 0.0, s1 = 0.0;
                    int r = 0;
                    for (; r + 1 < n; r += 2) { s0 += J0[r * L.Ncap + a] * pr[r]; s1 += J0[(r + 1) * L.Ncap + a] * pr[r + 1]; }
                    if (r < n) s0 += J0[r * L.Ncap + a] * pr[r];
                    g[ca] += s0 + s1;
                }
            } else {
                int a, b;
                tri_decode(w, a, b);
                const int ca = pmap[a], cb = pmap[b];
                if (ca >= 0 && cb >= 0) S[ca >= cb ? tri(ca, cb) : tri(cb, ca)] += Hp[a * L.Ncap + b];
            }
        }
    }
    const double tot = block_sum(c, 0.5 * (cost2 + costrho));
    PROF_ADD(PF_PRACC);
    return tot;
}

// cost only at (x, lam)
DEV double cost_only(const Ctx& c, const double* x, const double* lam) {
    __syncthreads();
    PROF_DECL;
    PROF_T0();
    double c2 = imu_pass<false>(c, x);
    c2 += prior_pass(c, x, c.sc + c.Lp->so_prc);
    const double cr = proj_cost_pass(c, x, lam);
    const double tot = block_sum(c, 0.5 * (c2 + cr));
    PROF_ADD(PF_CAND);
    return tot;
}

// ================================================================================================
// Scaled, damped reduced system + landmark Schur complement (MFMA) + Cholesky
// ================================================================================================
// S <- diag(sc) S diag(sc) + mu*Dg^2 ; augmented row R <- sc .* g  (the rhs).
// Also returns this thread's share of  t^T H~ t  over the reduced block (H~ = scaled, UN-damped Hessian,
// t = V_T = gt / Dg): the Cauchy-point denominator |J~ t|^2 of DoglegStrategy::ComputeCauchyPoint without a
// second pass over the factors.
NOINL double build_scaled(const Ctx& c, double mu) {
    const BaLayout& L = *c.Lp;
    double* S = LDSB + L.l_S;
    const double* g = LDSB + L.l_vec + V_G * L.Rpad;
    const double* sc = LDSB + L.l_vec + V_SC * L.Rpad;
    const double* dg = LDSB + L.l_vec + V_DG * L.Rpad;
    const double* tv = LDSB + L.l_vec + V_T * L.Rpad;
    const int R = L.R;
    const int n = R * (R + 1) / 2;
    double q = 0.0;
    for (int w = c.tid; w < n; w += BA_NT) {
        int a, b;
        tri_decode(w, a, b);
        double v = S[w] * sc[a] * sc[b];
        q += v * tv[a] * tv[b] * (a == b ? 1.0 : 2.0);
        if (a == b) v += mu * dg[a] * dg[a];
        S[w] = v;
    }
    for (int k = c.tid; k < R; k += BA_NT) S[tri(R, k)] = sc[k] * g[k];
    if (c.tid == 0) S[tri(R, R)] = 0.0;
    return q;
}

// Landmark Schur complement on the camera part:  S_cam -= Wd Wd^T,  rhs_cam -= Wd * bd
//   Wd[c][l] = sc[c] * Wt[c][l] * sl[l] / sqrt(ht[l]),  bd[l] = sl[l]*b[l]/sqrt(ht[l]),
//   ht[l] = sl[l]^2 h[l] + mu*dgl[l]^2.
// 16x16 tiles of Wd Wd^T are accumulated with v_mfma_f64_16x16x4_f64 over chunks of 16 landmarks.
NOINL void schur_mfma(const Ctx& c, double mu) {
    const BaLayout& L = *c.Lp;
    double* S = LDSB + L.l_S;
    double* wd = LDSB + L.l_wd;                   // [RcPad][17] + bd[16]
    const double* sc = LDSB + L.l_vec + V_SC * L.Rpad;
    const double* Wt = c.sc + L.so_Wt;
    const double* h = c.sc + L.so_h;
    const double* b = c.sc + L.so_b;
    const double* sl = c.sc + L.so_sl;
    const double* dgl = c.sc + L.so_dgl;
    const int Rc = L.Rc, RcPad = L.RcPad, R = L.R;
    const int nt = RcPad / 16;
    const int ntile = nt * (nt + 1) / 2;
    double4_t acc[2];                              // up to 2 tiles per wavefront (15 tiles / 8 waves)
    double racc = 0.0;                             // rhs accumulation: thread tid < Rc owns row tid
    acc[0] = (double4_t){0, 0, 0, 0};
    acc[1] = (double4_t){0, 0, 0, 0};
    int tm[2], tn[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int t = c.wave + s * BA_NW;
        int a, bq;
        tri_decode(t, a, bq);
        tm[s] = a; tn[s] = bq;
    }
    double* bd = wd + RcPad * 17;
    double* lsc = c.sc + L.so_yl;                 // sl / sqrt(h~): yl is free until the back substitution
    for (int l = c.tid; l < c.nL; l += BA_NT) lsc[l] = sl[l] / sqrt(sl[l] * sl[l] * h[l] + mu * dgl[l] * dgl[l]);
    for (int l0 = 0; l0 < c.nL; l0 += 16) {
        __syncthreads();
        for (int w = c.tid; w < RcPad * 16; w += BA_NT) {
            const int row = w / 16, k = w % 16, l = l0 + k;
            const bool in = row < Rc && l < c.nL;
            const double wv = Wt[(in ? row : 0) * L.Lcap + (in ? l : 0)];
            const double v = in ? sc[row] * wv * lsc[l] : 0.0;
            wd[row * 17 + k] = v;
        }
        if (c.tid < 16) {
            const int l = l0 + c.tid;
            bd[c.tid] = (l < c.nL) ? b[l] * lsc[l] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (c.wave + s * BA_NW < ntile) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double a = wd[(tm[s] * 16 + (c.lane & 15)) * 17 + kk * 4 + (c.lane >> 4)];
                    const double bb = wd[(tn[s] * 16 + (c.lane & 15)) * 17 + kk * 4 + (c.lane >> 4)];
                    acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[s], 0, 0, 0);
                }
            }
        }
        if (c.tid < Rc) {
#pragma unroll
            for (int k = 0; k < 16; ++k) racc += wd[c.tid * 17 + k] * bd[k];
        }
    }
    __syncthreads();
    // D[row = (lane>>4) + 4*reg][col = lane&15]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        if (c.wave + s * BA_NW < ntile) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int row = tm[s] * 16 + (c.lane >> 4) + 4 * reg;
                const int col = tn[s] * 16 + (c.lane & 15);
                if (row < Rc && col <= row) S[tri(row, col)] -= acc[s][reg];
            }
        }
    }
    if (c.tid < Rc) S[tri(R, c.tid)] -= racc;
    __syncthreads();
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

// Blocked right-looking Cholesky (NB = 16) of the packed lower triangle S (R x R) held in LDS, with the rhs as
// augmented row R (so row R of the factor is the forward-substituted rhs):
//   (1) the 16x16 diagonal block is factored by ONE wavefront in registers — lane i holds row i, pivots and
//       multipliers travel through v_readlane, no barrier inside the block;
//   (2) the panel rows below it are solved one thread per row (x L_D^T = a, L_D broadcast from LDS);
//   (3) the trailing matrix gets its rank-16 update tile by tile on v_mfma_f64_16x16x4_f64.
// Three barriers per 16 columns.  1/L_jj goes to V_DI.  Returns false (uniform) on a bad pivot.
NOINL bool cholesky_aug(const Ctx& c) {
    const BaLayout& L = *c.Lp;
    double* S = LDSB + L.l_S;
    double* dinvv = LDSB + L.l_vec + V_DI * L.Rpad;
    int* flag = (int*)(LDSB + L.l_red + 24);
    const int R = L.R;
    const int lane = c.lane;
    if (c.tid == 0) *flag = 1;
    __syncthreads();
    for (int c0 = 0; c0 < R; c0 += 16) {
        const int nb = (R - c0) < 16 ? (R - c0) : 16;
        // ---- (1) diagonal block, wavefront 0
        if (c.wave == 0) {
            double a[16];
            const int i = lane & 15;
#pragma unroll
            for (int m = 0; m < 16; ++m) a[m] = (lane < 16 && i < nb && m <= i) ? S[tri(c0 + i, c0 + m)] : 0.0;
            bool good = true;
#pragma unroll
            for (int jj = 0; jj < 16; ++jj) {
                if (jj < nb) {
                    const double piv = readlane_d(a[jj], jj);
                    if (!(piv > 0.0) || !(piv < 1e300)) good = false;
                    const double dinv = rsqrt_nr(piv);
                    const double l = a[jj] * dinv;              // column jj of row `lane` (lane jj: sqrt(piv))
                    a[jj] = l;
                    if (lane == 0) dinvv[c0 + jj] = dinv;
#pragma unroll
                    for (int k = jj + 1; k < 16; ++k) {
                        const double lk = readlane_d(l, k);
                        a[k] -= l * lk;
                    }
                }
            }
            if (lane < 16 && i < nb) {
#pragma unroll
                for (int m = 0; m < 16; ++m) if (m <= i) S[tri(c0 + i, c0 + m)] = a[m];
            }
            if (!good && lane == 0) *flag = 0;
        }
        __syncthreads();
        if (*flag == 0) break;
        // ---- (2) panel: rows i > block, x_c = (a_c - sum_{m<c} x_m L_D[c][m]) / L_D[c][c]
        const int r1 = c0 + nb;
        if (nb == 16) {
            // full block: straight-line code, every LDS read unconditional (the compiler batches them)
            for (int i = r1 + c.tid; i <= R; i += BA_NT) {
                double x[16];
                double* row = S + tri(i, c0);
#pragma unroll
                for (int m = 0; m < 16; ++m) x[m] = row[m];
#pragma unroll
                for (int cc = 0; cc < 16; ++cc) {
                    const double* ld = S + tri(c0 + cc, c0);
                    double sacc = x[cc];
#pragma unroll
                    for (int m = 0; m < cc; ++m) sacc -= x[m] * ld[m];
                    x[cc] = sacc * dinvv[c0 + cc];
                }
#pragma unroll
                for (int m = 0; m < 16; ++m) row[m] = x[m];
            }
        } else {
            for (int i = r1 + c.tid; i <= R; i += BA_NT) {
                double* row = S + tri(i, c0);
                for (int cc = 0; cc < nb; ++cc) {
                    const double* ld = S + tri(c0 + cc, c0);
                    double sacc = row[cc];
                    for (int m = 0; m < cc; ++m) sacc -= row[m] * ld[m];
                    row[cc] = sacc * dinvv[c0 + cc];
                }
            }
        }
        __syncthreads();
        // ---- (3) trailing update (only full blocks have anything right of them)
        if (nb == 16 && r1 < R) {
            const int t0 = r1 >> 4;
            const int nt = (R >> 4) + 1;                 // tile rows covering rows 0..R
            const int m = nt - t0;
            const int ntile = m * (m + 1) / 2;
            for (int t = c.wave; t < ntile; t += BA_NW) {
                int tr_, tc_;
                tri_decode(t, tr_, tc_);
                const int ti = t0 + tr_, tk = t0 + tc_;
                const int arow = 16 * ti + (lane & 15), brow = 16 * tk + (lane & 15);
                const bool interior = (16 * ti + 15 <= R) && (ti != tk);     // wave-uniform
                const double* pa = S + tri(arow <= R ? arow : R, c0) + (lane >> 4);
                const double* pb = S + tri(brow <= R ? brow : R, c0) + (lane >> 4);
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
                    double* q0 = S + tri(row0, colw);
                    double* q1 = S + tri(row0 + 4, colw);
                    double* q2 = S + tri(row0 + 8, colw);
                    double* q3 = S + tri(row0 + 12, colw);
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
        __syncthreads();
    }
    const bool ok = (*flag != 0);
    __syncthreads();
    return ok;
}


// y <- solve L^T y = (row R of S) by one wavefront (lane owns entries lane, lane+64, lane+128 of the running
// rhs in registers; the pivot value travels through v_readlane, the next row of L is prefetched); result in V_Y.
NOINL void back_substitute(const Ctx& c) {
    const BaLayout& L = *c.Lp;
    const double* S = LDSB + L.l_S;
    const double* dinvv = LDSB + L.l_vec + V_DI * L.Rpad;
    double* y = LDSB + L.l_vec + V_Y * L.Rpad;
    const int R = L.R;
    __syncthreads();
    if (c.wave == 0) {
        const int l0 = c.lane, l1 = c.lane + 64, l2 = c.lane + 128;
        const double* rowR = S + tri(R, 0);
        double v0 = rowR[l0 < R ? l0 : 0], v1 = rowR[l1 < R ? l1 : 0], v2 = rowR[l2 < R ? l2 : 0];
        v0 = l0 < R ? v0 : 0.0; v1 = l1 < R ? v1 : 0.0; v2 = l2 < R ? v2 : 0.0;
        // A lone wave pays ~8 cycles per FP64 instruction whatever the dependencies, so the row loop is specialised
        // by the 64-column slot that holds the pivot: rows j < 128 have no entries in columns >= 128 (v2 is final),
        // rows j < 64 none in columns >= 64 (v1 final) -> fewer loads, FMAs and selects per row.
        int j = R - 1;
        {   // ---- slot 2: pivot in v2, entries in all three registers
            const int jlo = 128;
            if (j >= jlo) {
                const double* rj = S + tri(j, 0);
                double r0 = rj[l0], r1 = rj[l1], r2 = rj[l2 < j ? l2 : 0];
                r2 = l2 < j ? r2 : 0.0;
                double di = dinvv[j];
                for (; j >= jlo; --j) {
                    const int jp = j > jlo ? j - 1 : j;
                    const double* rn = S + tri(jp, 0);
                    const double n0 = rn[l0], n1 = rn[l1], n2 = rn[l2 < jp ? l2 : 0];
                    const double dn = dinvv[jp];
                    const int own = j & 63;
                    const double xj = readlane_d(v2, own) * di;
                    v2 = c.lane == own ? xj : v2 - r2 * xj;
                    v1 -= r1 * xj;
                    v0 -= r0 * xj;
                    r0 = n0; r1 = n1; r2 = l2 < jp ? n2 : 0.0;
                    di = dn;
                }
            }
        }
        {   // ---- slot 1: pivot in v1
            const int jlo = 64;
            if (j >= jlo) {
                const double* rj = S + tri(j, 0);
                double r0 = rj[l0], r1 = rj[l1 < j ? l1 : 0];
                r1 = l1 < j ? r1 : 0.0;
                double di = dinvv[j];
                for (; j >= jlo; --j) {
                    const int jp = j > jlo ? j - 1 : j;
                    const double* rn = S + tri(jp, 0);
                    const double n0 = rn[l0], n1 = rn[l1 < jp ? l1 : 0];
                    const double dn = dinvv[jp];
                    const int own = j & 63;
                    const double xj = readlane_d(v1, own) * di;
                    v1 = c.lane == own ? xj : v1 - r1 * xj;
                    v0 -= r0 * xj;
                    r0 = n0; r1 = l1 < jp ? n1 : 0.0;
                    di = dn;
                }
            }
        }
        if (j >= 0) {   // ---- slot 0: pivot in v0
            const double* rj = S + tri(j, 0);
            double r0 = rj[l0 < j ? l0 : 0];
            r0 = l0 < j ? r0 : 0.0;
            double di = dinvv[j];
            for (; j >= 0; --j) {
                const int jp = j > 0 ? j - 1 : 0;
                const double* rn = S + tri(jp, 0);
                const double n0 = rn[l0 < jp ? l0 : 0];
                const double dn = dinvv[jp];
                const double xj = readlane_d(v0, j) * di;
                v0 = c.lane == j ? xj : v0 - r0 * xj;
                r0 = l0 < jp ? n0 : 0.0;
                di = dn;
            }
        }
        if (c.lane < R) y[c.lane] = v0;
        if (c.lane + 64 < R) y[c.lane + 64] = v1;
        if (c.lane + 128 < R) y[c.lane + 128] = v2;
    }
    __syncthreads();
}

// ================================================================================================
// gauge fix of Estimator::double2vector() (estimator.cpp:530-577) + vector2double repack (:486-528)
// ================================================================================================
DEV void R2ypr_dev(const double* R, double* ypr) {
    const double n0 = R[0], n1 = R[3], n2 = R[6];
    const double o0 = R[1], o1 = R[4];
    const double a0 = R[2], a1 = R[5];
    const double y = atan2(n1, n0);
    const double p = atan2(-n2, n0 * cos(y) + n1 * sin(y));
    const double r = atan2(a0 * sin(y) - a1 * cos(y), -o0 * sin(y) + o1 * cos(y));
    ypr[0] = y / M_PI * 180.0; ypr[1] = p / M_PI * 180.0; ypr[2] = r / M_PI * 180.0;
}

// ================================================================================================
// THE SOLVE KERNEL
// ================================================================================================
extern "C" __global__ __launch_bounds__(BA_NT, 2) void ba_solve_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) {
    const BaLayout& L = *Lp;
    Ctx c;
    c.Lp = Lp;
    const int w = blockIdx.x;
    c.ia = P.iarr + (size_t)w * L.istride;
    c.hdr = c.ia + L.io_hdr;
    c.di = P.din + (size_t)w * L.dstride;
    c.sc = P.scr + (size_t)w * L.sstride;
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6;
    c.nL = c.hdr[H_L]; c.nF = c.hdr[H_F]; c.nprior = c.hdr[H_NPRIOR]; c.nblk = c.hdr[H_NBLK];
    c.nchunk = c.hdr[H_NCHUNK];
    c.focal = c.di[L.do_par + P_FOCAL]; c.tr = c.di[L.do_par + P_TR]; c.row = c.di[L.do_par + P_ROW];
    c.gnorm = c.di[L.do_par + P_GNORM];
    const int max_iters = c.hdr[H_MAXIT];
    const int R = L.R, nL = c.nL;
    double* out = P.out + (size_t)w * L.ostride;
    int* iout = P.iout + (size_t)w * L.oi_stride;

    double* x = LDSB + L.l_x;        // current state
    double* xc = LDSB + L.l_xc;      // candidate
    double* lam = c.sc + L.so_lam;
    double* lamc = c.sc + L.so_lamc;
    double* vG = LDSB + L.l_vec + V_G * L.Rpad;
    double* vSC = LDSB + L.l_vec + V_SC * L.Rpad;
    double* vDG = LDSB + L.l_vec + V_DG * L.Rpad;
    double* vGT = LDSB + L.l_vec + V_GT * L.Rpad;
    double* vGN = LDSB + L.l_vec + V_GN * L.Rpad;
    double* vU = LDSB + L.l_vec + V_U * L.Rpad;
    double* vY = LDSB + L.l_vec + V_Y * L.Rpad;
    double* S = LDSB + L.l_S;
    double* h = c.sc + L.so_h; double* b = c.sc + L.so_b; double* sl = c.sc + L.so_sl;
    double* dgl = c.sc + L.so_dgl; double* gtl = c.sc + L.so_gtl; double* gnl = c.sc + L.so_gnl;
    double* ul = c.sc + L.so_ul; double* yl = c.sc + L.so_yl;

    // ---- load state
    const int nst = st_size(L);
    for (int k = c.tid; k < 7 * L.Kp; k += BA_NT) x[k] = c.di[L.do_pose + k];
    for (int k = c.tid; k < 9 * L.K; k += BA_NT) x[7 * L.Kp + k] = c.di[L.do_sb + k];
    if (c.tid < 7) x[7 * L.Kp + 9 * L.K + c.tid] = c.di[L.do_ex + c.tid];
    if (c.tid == 7) x[7 * L.Kp + 9 * L.K + 7] = c.di[L.do_td];
    for (int k = c.tid; k < nL; k += BA_NT) lam[k] = c.di[L.do_lam + k];
    __syncthreads();

    PROF_DECL;
#ifdef BA_PROFILE
    if (c.tid < 16) LDSB[L.l_misc + c.tid] = 0.0;
    __syncthreads();
    const long long _pstart = clock64();
#endif
    PROF_T0();
    // ---- once per solve: IMU sqrt_info, prior J0^T J0
    imu_sqrt_info(c);
    if (c.nprior) {
        // J0^T J0 once per solve, J0 staged in LDS (the factor staging area is idle here)
        const int n = c.nprior;
        const double* J0 = c.di + L.do_pJ0;
        double* Hp = c.sc + L.so_Hp;
        double* J0s = LDSB + L.l_stage;          // n x n, row stride n
        for (int wk = c.tid; wk < n * n; wk += BA_NT) J0s[wk] = J0[(wk / n) * L.Ncap + wk % n];
        int* pmap = (int*)(LDSB + L.l_pmap);
        {
            const int* kind = c.ia + L.io_pb_kind;
            const int* off = c.ia + L.io_pb_off;
            const int* pcol = c.ia + L.io_pb_col;
            for (int blk = c.tid; blk < c.nblk; blk += BA_NT) {
                const int sz = (kind[blk] == VG_BLK_SPEEDBIAS) ? 9 : (kind[blk] == VG_BLK_TD ? 1 : 6);
                for (int k = 0; k < sz; ++k) pmap[off[blk] + k] = pcol[blk] >= 0 ? pcol[blk] + k : -1;
            }
        }
        __syncthreads();
        for (int wk = c.tid; wk < n * (n + 1) / 2; wk += BA_NT) {
            int a, bb;
            tri_decode(wk, a, bb);
            double s0 = 0.0, s1 = 0.0;
            int r = 0;
            for (; r + 1 < n; r += 2) { s0 += J0s[r * n + a] * J0s[r * n + bb]; s1 += J0s[(r + 1) * n + a] * J0s[(r + 1) * n + bb]; }
            if (r < n) s0 += J0s[r * n + a] * J0s[r * n + bb];
            Hp[a * L.Ncap + bb] = s0 + s1;
        }
    }
    __syncthreads();
    PROF_ADD(PF_PRO);
    // ---- iteration 0
    double cost = linearize(c, x, lam);
    __syncthreads();
    for (int k = c.tid; k < R; k += BA_NT) vSC[k] = 1.0 / (1.0 + sqrt(S[tri(k, k)]));
    for (int k = c.tid; k < nL; k += BA_NT) sl[k] = 1.0 / (1.0 + sqrt(h[k]));
    __syncthreads();
    const double initial_cost = cost;
    double gmax;
    {
        double m = 0.0;
        for (int k = c.tid; k < R; k += BA_NT) m = fmax(m, fabs(vG[k]));
        for (int k = c.tid; k < nL; k += BA_NT) m = fmax(m, fabs(b[k]));
        gmax = block_max(c, m);
    }
    int termination = VG_TERM_NO_CONVERGENCE, status = VG_OK;
    int it = 0, num_accepted = 0, num_invalid = 0;
    double radius = 1e4, mu = 1e-8;
    const double min_mu = 1e-8, max_mu = 1.0;
    bool reuse = false;
    double alpha = 0.0, gtn2 = 0.0, gnn2 = 0.0, gtgn = 0.0, dogleg_norm = 0.0, mu_solved = 1e-8;
    double x_norm;
    {
        double s = 0.0;
        const int nx = 7 * L.Kp + 9 * L.K + (L.e ? 7 : 0);
        for (int k = c.tid; k < nx; k += BA_NT) s += x[k] * x[k];
        if (L.t && c.tid == 0) s += x[7 * L.Kp + 9 * L.K + 7] * x[7 * L.Kp + 9 * L.K + 7];
        for (int k = c.tid; k < nL; k += BA_NT) s += lam[k] * lam[k];
        x_norm = sqrt(block_sum(c, s));
    }
    if (!(cost == cost) || !(cost < 1e300)) { status = VG_ERR_NUMERIC; termination = VG_TERM_FAILURE; }
    else if (gmax <= 1e-10) termination = VG_TERM_CONVERGENCE;

    while (termination == VG_TERM_NO_CONVERGENCE && status == VG_OK && it < max_iters) {
        ++it;
        const int slot = it - 1;
        bool ok = true;
        if (!reuse) {
            reuse = true;
            // Dg, gt (scaled gradient / Dg), t = gt / Dg
            __syncthreads();
            double* vT = LDSB + L.l_vec + V_T * L.Rpad;
            for (int k = c.tid; k < R; k += BA_NT) {
                const double d2 = vSC[k] * vSC[k] * S[tri(k, k)];
                const double d = sqrt(fmin(fmax(d2, 1e-6), 1e32));
                vDG[k] = d;
                vGT[k] = vSC[k] * vG[k] / d;
                vT[k] = vGT[k] / d;
            }
            for (int k = c.tid; k < nL; k += BA_NT) {
                const double d2 = sl[k] * sl[k] * h[k];
                const double d = sqrt(fmin(fmax(d2, 1e-6), 1e32));
                dgl[k] = d;
                gtl[k] = sl[k] * b[k] / d;
            }
            __syncthreads();
            PROF_T0();
            double qland = 0.0;        // landmark part of t^T H~ t: sum_l [ h~_l t_l^2 + 2 t_l (w~_l . t_cam) ]
            {
                double s = 0.0;
                for (int k = c.tid; k < R; k += BA_NT) s += vGT[k] * vGT[k];
                const double* Wt = c.sc + L.so_Wt;
                for (int l = c.tid; l < nL; l += BA_NT) {
                    s += gtl[l] * gtl[l];
                    const double tl = gtl[l] / dgl[l];
                    double wdot = 0.0;
                    for (int k = 0; k < L.Rc; ++k) wdot += vSC[k] * Wt[k * L.Lcap + l] * vT[k];
                    qland += sl[l] * sl[l] * h[l] * tl * tl + 2.0 * tl * sl[l] * wdot;
                }
                gtn2 = block_sum(c, s);
            }
            PROF_ADD(PF_JVEC);
            // Gauss-Newton step, increasing mu on failure (DoglegStrategy::ComputeGaussNewtonStep)
            bool solved = false;
            bool first = true;
            while (mu < max_mu) {
                if (!first) { cost = linearize(c, x, lam); }     // S was destroyed by the failed attempt (Wt, h unchanged)
                first = false;
                __syncthreads();
                PROF_T0();
                {
                    double q = build_scaled(c, mu) + qland;
                    q = block_sum(c, q);
                    alpha = gtn2 / q;            // |gt|^2 / |J~ (gt/Dg)|^2
                }
                __syncthreads();
                PROF_ADD(PF_BUILD);
                schur_mfma(c, mu);
                PROF_ADD(PF_SCHUR);
                const bool cok = cholesky_aug(c);
                PROF_ADD(PF_CHOL);
                if (cok) {
                    back_substitute(c);
                    PROF_ADD(PF_BACK);
                    // landmarks: y_l = (bt_l - wt_l . y_cam) / ht_l
                    for (int l = c.tid; l < nL; l += BA_NT) {
                        const double ht = sl[l] * sl[l] * h[l] + mu * dgl[l] * dgl[l];
                        double s = sl[l] * b[l];
                        const double* Wt = c.sc + L.so_Wt;
                        for (int k = 0; k < L.Rc; ++k) s -= vSC[k] * Wt[k * L.Lcap + l] * sl[l] * vY[k];
                        yl[l] = s / ht;
                    }
                    __syncthreads();
                    double fin = 0.0;
                    for (int k = c.tid; k < R; k += BA_NT) fin += (vY[k] == vY[k] && fabs(vY[k]) < 1e300) ? 0.0 : 1.0;
                    for (int k = c.tid; k < nL; k += BA_NT) fin += (yl[k] == yl[k] && fabs(yl[k]) < 1e300) ? 0.0 : 1.0;
                    if (block_sum(c, fin) == 0.0) { solved = true; mu_solved = mu; break; }
                }
                mu *= 10.0;
            }
            if (!solved) ok = false;
            else {
                double s1 = 0.0, s2 = 0.0;
                for (int k = c.tid; k < R; k += BA_NT) {
                    vGN[k] = -vY[k] * vDG[k];
                    s1 += vGN[k] * vGN[k];
                    s2 += vGN[k] * vGT[k];
                }
                for (int k = c.tid; k < nL; k += BA_NT) {
                    gnl[k] = -yl[k] * dgl[k];
                    s1 += gnl[k] * gnl[k];
                    s2 += gnl[k] * gtl[k];
                }
                block_sum2(c, s1, s2);
                gnn2 = s1; gtgn = s2;
                // the failed-attempt relinearisation overwrote S; restore an unscaled linearisation
                // only when needed (next accepted step re-linearises anyway)
            }
        }
        double model_change = 0.0;
        double c_gt = 0.0, c_gn = 0.0;
        if (ok) {
            // DoglegStrategy::ComputeTraditionalDoglegStep
            const double gtn = sqrt(gtn2), gnn = sqrt(gnn2);
            if (gnn <= radius) { c_gt = 0.0; c_gn = 1.0; dogleg_norm = gnn; }
            else if (gtn * alpha >= radius) { c_gt = -(radius / gtn); c_gn = 0.0; dogleg_norm = radius; }
            else {
                const double b_dot_a = -alpha * gtgn;
                const double a_sq = (alpha * gtn) * (alpha * gtn);
                const double bma_sq = a_sq - 2 * b_dot_a + gnn2;
                const double cc = b_dot_a - a_sq;
                const double dd = sqrt(cc * cc + bma_sq * (radius * radius - a_sq));
                const double beta = (cc <= 0) ? (dd - cc) / bma_sq : (radius * radius - a_sq) / (dd + cc);
                c_gt = -alpha * (1.0 - beta); c_gn = beta;
                dogleg_norm = sqrt(c_gt * c_gt * gtn2 + 2 * c_gt * c_gn * gtgn + c_gn * c_gn * gnn2);
            }
            __syncthreads();
            // delta = scale .* (s ./ Dg)
            for (int k = c.tid; k < R; k += BA_NT) vU[k] = vSC[k] * ((c_gt * vGT[k] + c_gn * vGN[k]) / vDG[k]);
            for (int k = c.tid; k < nL; k += BA_NT) ul[k] = sl[k] * ((c_gt * gtl[k] + c_gn * gnl[k]) / dgl[k]);
            __syncthreads();
            // model cost change  -(J~ s)^T (r + J~ s / 2)  with s = c_gt a + c_gn b  (a = gt/Dg, b = gn/Dg = -y):
            //   a.g~ = |gt|^2, b.g~ = gt.gn, a^T H~ a = |gt|^2 / alpha, and from (H~ + mu Dg^2) y = g~ :
            //   b^T H~ b = -gt.gn - mu |gn|^2,  a^T H~ b = -|gt|^2 - mu gt.gn      (no pass over the factors)
            {
                const double q11 = gtn2 / alpha;
                const double q12 = -gtn2 - mu_solved * gtgn;
                const double q22 = -gtgn - mu_solved * gnn2;
                model_change = -(c_gt * gtn2 + c_gn * gtgn) - 0.5 * (c_gt * c_gt * q11 + 2.0 * c_gt * c_gn * q12 + c_gn * c_gn * q22);
            }
        }
        if (c.tid == 0) {
            out[L.oo_trace + 0 * VG_MAX_ITERS + slot] = cost;
            out[L.oo_trace + 3 * VG_MAX_ITERS + slot] = radius;
        }
        if (!ok || !(model_change > 0.0)) {
            if (c.tid == 0) {
                out[L.oo_trace + 1 * VG_MAX_ITERS + slot] = 0.0;
                out[L.oo_trace + 2 * VG_MAX_ITERS + slot] = model_change;
                out[L.oo_trace + 4 * VG_MAX_ITERS + slot] = 0.0;
                iout[4 + slot] = 0;
            }
            ++num_invalid;
            if (num_invalid >= 5) { termination = VG_TERM_FAILURE; break; }
            mu *= 10.0;
            reuse = false;
            // S holds a Cholesky factor: rebuild the linearisation at the unchanged x
            cost = linearize(c, x, lam);
            continue;
        }
        num_invalid = 0;
        // ---- candidate
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
        for (int k = c.tid; k < nL; k += BA_NT) lamc[k] = lam[k] + ul[k];
        __syncthreads();
        const double cost_cand = cost_only(c, xc, lamc);
        double step_norm;
        {
            double s = 0.0;
            const int nx = 7 * L.Kp + 9 * L.K + (L.e ? 7 : 0);
            for (int k = c.tid; k < nx; k += BA_NT) { const double d = x[k] - xc[k]; s += d * d; }
            if (L.t && c.tid == 0) { const double d = x[7 * L.Kp + 9 * L.K + 7] - xc[7 * L.Kp + 9 * L.K + 7]; s += d * d; }
            for (int k = c.tid; k < nL; k += BA_NT) { const double d = lam[k] - lamc[k]; s += d * d; }
            step_norm = sqrt(block_sum(c, s));
        }
        if (c.tid == 0) {
            out[L.oo_trace + 1 * VG_MAX_ITERS + slot] = cost_cand;
            out[L.oo_trace + 2 * VG_MAX_ITERS + slot] = model_change;
            out[L.oo_trace + 4 * VG_MAX_ITERS + slot] = dogleg_norm;
        }
        if (step_norm <= 1e-8 * (x_norm + 1e-8)) { if (c.tid == 0) iout[4 + slot] = 1; termination = VG_TERM_CONVERGENCE; break; }
        if (fabs(cost - cost_cand) <= 1e-6 * cost) { if (c.tid == 0) iout[4 + slot] = 1; termination = VG_TERM_CONVERGENCE; break; }
        const double rho = (cost - cost_cand) / model_change;
        if (rho > 1e-3) {
            if (c.tid == 0) iout[4 + slot] = 3;
            ++num_accepted;
            __syncthreads();
            for (int k = c.tid; k < nst; k += BA_NT) x[k] = xc[k];
            for (int k = c.tid; k < nL; k += BA_NT) lam[k] = lamc[k];
            __syncthreads();
            {
                double s = 0.0;
                const int nx = 7 * L.Kp + 9 * L.K + (L.e ? 7 : 0);
                for (int k = c.tid; k < nx; k += BA_NT) s += x[k] * x[k];
                if (L.t && c.tid == 0) s += x[7 * L.Kp + 9 * L.K + 7] * x[7 * L.Kp + 9 * L.K + 7];
                for (int k = c.tid; k < nL; k += BA_NT) s += lam[k] * lam[k];
                x_norm = sqrt(block_sum(c, s));
            }
            cost = linearize(c, x, lam);
            if (!(cost == cost)) { status = VG_ERR_NUMERIC; termination = VG_TERM_FAILURE; break; }
            if (rho < 0.25) radius *= 0.5;
            if (rho > 0.75) radius = fmax(radius, 3.0 * dogleg_norm);
            mu = fmax(min_mu, 2.0 * mu / 10.0);
            reuse = false;
            double m = 0.0;
            __syncthreads();
            for (int k = c.tid; k < R; k += BA_NT) m = fmax(m, fabs(vG[k]));
            for (int k = c.tid; k < nL; k += BA_NT) m = fmax(m, fabs(b[k]));
            if (block_max(c, m) <= 1e-10) { termination = VG_TERM_CONVERGENCE; break; }
        } else {
            if (c.tid == 0) iout[4 + slot] = 1;
            radius *= 0.5;
            reuse = true;
        }
    }

    // ---- outputs: gauge fix (double2vector) + repack
    __syncthreads();
    {
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
        for (int i = c.tid; i < L.K; i += BA_NT) {
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
            const double* sb = x + 7 * L.Kp + 9 * i;
            double vo[3];
            m3_vec(rot, sb, vo);
            double* os = out + L.oo_sb + 9 * i;
            os[0] = vo[0]; os[1] = vo[1]; os[2] = vo[2];
            for (int k = 3; k < 9; ++k) os[k] = sb[k];
        }
        if (c.tid == 0) {
            const double* exx = x + 7 * L.Kp + 9 * L.K;
            double Rc[9], qo[4];
            q_to_R(exx + 3, Rc);
            R_to_q(Rc, qo);
            double* o = out + L.oo_ex;
            o[0] = exx[0]; o[1] = exx[1]; o[2] = exx[2]; o[3] = qo[0]; o[4] = qo[1]; o[5] = qo[2]; o[6] = qo[3];
            out[L.oo_td] = exx[7];
            if (L.Kp > L.K) {
                // relocalisation pose: same gauge transform (estimator.cpp:598-603)
                const int i = L.K;
                double q[4] = {x[7 * i + 3], x[7 * i + 4], x[7 * i + 5], x[7 * i + 6]};
                q_normalize(q);
                double Rq[9], Ri[9], d[3], po[3];
                q_to_R(q, Rq);
                m3_mul(rot, Rq, Ri);
                R_to_q(Ri, qo);
                d[0] = x[7 * i] - x[0]; d[1] = x[7 * i + 1] - x[1]; d[2] = x[7 * i + 2] - x[2];
                m3_vec(rot, d, po);
                double* orp = out + L.oo_pose + 7 * i;
                orp[0] = po[0] + p0_in[0]; orp[1] = po[1] + p0_in[1]; orp[2] = po[2] + p0_in[2];
                orp[3] = qo[0]; orp[4] = qo[1]; orp[5] = qo[2]; orp[6] = qo[3];
            }
            out[L.oo_sum + 0] = initial_cost;
            out[L.oo_sum + 1] = cost;
            out[L.oo_sum + 2] = radius;
            iout[0] = status; iout[1] = termination; iout[2] = it; iout[3] = num_accepted;
        }
        // setDepth/getDepthVector round trip (feature_manager.cpp:141-200)
        for (int k = c.tid; k < nL; k += BA_NT) out[L.oo_lam + k] = 1.0 / (1.0 / lam[k]);
#ifdef BA_PROFILE
        __syncthreads();
        if (c.tid == 0) {
            for (int k = 0; k < 15; ++k) out[L.oo_trace + 5 * VG_MAX_ITERS + k] = LDSB[L.l_misc + k];
            out[L.oo_trace + 5 * VG_MAX_ITERS + 15] = (double)(clock64() - _pstart);
        }
#endif
    }
}

// ================================================================================================
// Factor-evaluation kernel for parity tests (vg_ba_eval_factors): raw (no loss) residuals/Jacobians.
// proj_J [F][2][20] = [pose_i 6 | pose_j 6 | ex 6 | lambda | td];  imu_J [K-1][15][30]
// ================================================================================================
extern "C" __global__ __launch_bounds__(BA_NT, 2) void ba_eval_factors_kernel(const BaLayout* __restrict__ Lp, BaPtrs P, double* proj_r,
                                                                          double* proj_J, double* imu_r,
                                                                          double* imu_J, double* prior_r) {
    const BaLayout& L = *Lp;
    Ctx c;
    c.Lp = Lp;
    c.ia = P.iarr; c.hdr = c.ia + L.io_hdr; c.di = P.din; c.sc = P.scr;
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6;
    c.nL = c.hdr[H_L]; c.nF = c.hdr[H_F]; c.nprior = c.hdr[H_NPRIOR]; c.nblk = c.hdr[H_NBLK];
    c.nchunk = c.hdr[H_NCHUNK];
    c.focal = c.di[L.do_par + P_FOCAL]; c.tr = c.di[L.do_par + P_TR]; c.row = c.di[L.do_par + P_ROW];
    c.gnorm = c.di[L.do_par + P_GNORM];
    double* x = LDSB + L.l_x;
    for (int k = c.tid; k < 7 * L.Kp; k += BA_NT) x[k] = c.di[L.do_pose + k];
    for (int k = c.tid; k < 9 * L.K; k += BA_NT) x[7 * L.Kp + k] = c.di[L.do_sb + k];
    if (c.tid < 7) x[7 * L.Kp + 9 * L.K + c.tid] = c.di[L.do_ex + c.tid];
    if (c.tid == 7) x[7 * L.Kp + 9 * L.K + 7] = c.di[L.do_td];
    __syncthreads();
    imu_sqrt_info(c);
    __syncthreads();
    prior_pass(c, x, c.sc + L.so_pr);
    __syncthreads();
    const int nimu = L.K - 1;
    // weighted IMU residual / Jacobian, thread = (factor, column | residual); U = sqrt_info from imu_sqrt_info
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
    for (int k = c.tid; k < c.nprior; k += BA_NT) if (prior_r) prior_r[k] = c.sc[L.so_pr + k];
    const double* ex = st_ex(L, x);
    const double* lam = c.di + L.do_lam;
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
extern "C" hipError_t ba_launch_solve(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)ba_solve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)ba_eval_factors_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(ba_solve_kernel, dim3(L.nwin), dim3(BA_NT), L.lds_bytes, stream, dL, P);
    return hipGetLastError();
}

extern "C" hipError_t ba_launch_eval_factors(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, double* proj_r, double* proj_J,
                                            double* imu_r, double* imu_J, double* prior_r, hipStream_t stream) {
    hipError_t e = hipFuncSetAttribute((const void*)ba_eval_factors_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ba_eval_factors_kernel, dim3(1), dim3(BA_NT), L.lds_bytes, stream, dL, P, proj_r, proj_J, imu_r, imu_J, prior_r);
    return hipGetLastError();
}
