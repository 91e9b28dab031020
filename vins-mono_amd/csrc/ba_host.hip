// ba_host.hip — host side of the BA path behind the C-ABI (include/vinsgpu.h): packs
// vg_ba_problem windows into the device layout of ba_layout.h, launches the kernels of
// ba_kernels.hip / ba_marg.hip on the handle's stream and unpacks the results.
//
// The packing replaces the ceres::Problem construction of Estimator::optimization()
// (estimator.cpp:672-764, :769-801): instead of `new`-ing one cost-function object per residual it
// writes structure-of-arrays tables (factor list, per-chunk pair-sorted slot table, prior block map).
#include "vg_range.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "ba_layout.h"
#include "vg_handle.h"
#include "../../include/vinsgpu.h"

struct BaFork { hipStream_t aux; hipEvent_t fork, join; };
extern "C" hipError_t ba_launch_solve(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, int rounds, hipStream_t stream,
                                      hipEvent_t* ev, int* kinds, int* n_launches, const BaFork* fk);
typedef int (*BaAllReduce)(void* user, double* buf, size_t count, void* stream);
extern "C" hipError_t ba_launch_solve_big(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, int rounds, int slack, hipStream_t stream,
                                          BaAllReduce allreduce, void* user, int* hook_rc, hipEvent_t* ev, int* kinds, int* n_launches);
extern "C" hipError_t ba_launch_eval_factors(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, double* proj_r, double* proj_J,
                                            double* imu_r, double* imu_J, double* prior_r, hipStream_t stream);
extern "C" const char* ba_failed_launch();
extern "C" hipError_t ba_prepare_launch();
extern "C" hipError_t ba_launch_marg(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, hipStream_t stream);
extern "C" hipError_t ba_launch_carry_prior(int nwin, const double* mout, const int* miout, int mo_J0, int mo_r0, int mo_x0, int mo_stride,
                                            int mi_stride, int mcap, int x0cap, double* pri, int po_x0, int po_r0, int po_J0, int pld,
                                            int pstride, hipStream_t stream);

#define HIPCHK(h, expr)                                                                            \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                          \
            return VG_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

static inline int up(int v, int m) { return (v + m - 1) / m * m; }

static int blk_lsize(int kind) { return kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 6); }
static int blk_gsize(int kind) { return kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 7); }

static int check_problem(vg_handle* h, const vg_ba_problem* p) {
    if (!p || !p->pose || !p->speedbias || !p->ex_pose || !p->imu) { h->err = "null problem pointer"; return VG_ERR_BAD_ARG; }
    if (p->K < 2 || p->K + 1 > BA_MAX_K_LARGE) { h->err = "K out of range"; return VG_ERR_UNSUPPORTED; }
    if (p->L < 0 || p->n_obs < 0 || (p->L > 0 && (!p->inv_depth || !p->lm_start || !p->lm_nobs || !p->lm_obs_off || !p->obs))) {
        h->err = "bad landmark tables"; return VG_ERR_BAD_ARG;
    }
    for (int l = 0; l < p->L; ++l) {
        if (p->lm_nobs[l] < 2 || p->lm_start[l] < 0 || p->lm_start[l] + p->lm_nobs[l] > p->K ||
            p->lm_obs_off[l] < 0 || p->lm_obs_off[l] + p->lm_nobs[l] > p->n_obs) {
            h->err = "landmark track outside the window"; return VG_ERR_BAD_ARG;
        }
    }
    if (p->prior_n == VG_PRIOR_RESIDENT) {
        // the prior this window slot holds on the device: its block table is checked when it is resolved (vg_ba_batch_upload)
    } else if (p->prior_n < 0 || (p->prior_n > 0 && (!p->prior_block_kind || !p->prior_block_index || !p->prior_J0 || !p->prior_r0 || !p->prior_x0))) {
        h->err = "bad prior"; return VG_ERR_BAD_ARG;
    }
    if (p->prior_n > 0) {
        int n = 0;
        for (int b = 0; b < p->prior_nblocks; ++b) {
            const int k = p->prior_block_kind[b];
            if (k < 0 || k > 3) { h->err = "bad prior block kind"; return VG_ERR_BAD_ARG; }
            if ((k == VG_BLK_POSE || k == VG_BLK_SPEEDBIAS) && (p->prior_block_index[b] < 0 || p->prior_block_index[b] >= p->K)) {
                h->err = "prior block index outside the window"; return VG_ERR_BAD_ARG;
            }
            n += blk_lsize(k);
        }
        if (n != p->prior_n) { h->err = "prior_n != sum of local block sizes"; return VG_ERR_BAD_ARG; }
        // every parameter block at most once; speed-bias blocks of a prior must be chain neighbours (the reference's
        // priors hold sb_0 only: estimator.cpp:836-870, :913-930)
        int sb_lo = 1 << 30, sb_hi = -1, nsb = 0;
        for (int b = 0; b < p->prior_nblocks; ++b) {
            const int k = p->prior_block_kind[b], ix = (k == VG_BLK_POSE || k == VG_BLK_SPEEDBIAS) ? p->prior_block_index[b] : 0;
            for (int b2 = 0; b2 < b; ++b2) {
                const int k2 = p->prior_block_kind[b2];
                const int ix2 = (k2 == VG_BLK_POSE || k2 == VG_BLK_SPEEDBIAS) ? p->prior_block_index[b2] : 0;
                if (k == k2 && ix == ix2) { h->err = "prior lists a parameter block twice"; return VG_ERR_BAD_ARG; }
            }
            if (k == VG_BLK_SPEEDBIAS) { sb_lo = std::min(sb_lo, ix); sb_hi = std::max(sb_hi, ix); ++nsb; }
        }
        if (nsb > 2 || (nsb == 2 && sb_hi - sb_lo != 1)) {
            h->err = "prior couples speed-bias blocks that are not chain neighbours";
            return VG_ERR_UNSUPPORTED;
        }
    }
    if (p->relo_n < 0 || (p->relo_n > 0 && (!p->relo_pose || !p->relo_lm || !p->relo_xy))) { h->err = "bad relo"; return VG_ERR_BAD_ARG; }
    for (int k = 0; k < p->relo_n; ++k)
        if (p->relo_lm[k] < 0 || p->relo_lm[k] >= p->L) { h->err = "relo landmark out of range"; return VG_ERR_BAD_ARG; }
    if (p->max_iters < 0 || p->max_iters > VG_MAX_ITERS) { h->err = "max_iters out of range"; return VG_ERR_BAD_ARG; }
    return VG_OK;
}

// The prior of a window as the packer sees it: the caller's arrays, or (VG_PRIOR_RESIDENT) the block table of what the
// window slot holds on the device.
struct PriorRef {
    int n = 0, nb = 0;
    const int* kind = nullptr;
    const int* idx = nullptr;
    bool resident = false;
};

// ---- gather plan of assemble_small (ba_pipeline.hip): for every stored entry of the unscaled reduced system of the single-workgroup
//      path -- camera part S (packed lower), gradient g, chain blocks D_k / E_k -- where its terms come from.  A function of the
//      layout alone (which IMU factors exist and what the prior holds is decided on the device: the validity bit of the factor
//      index, the column -> prior-index map).  IMU factor f, local columns: 0-5 pose_f, 6-14 sb_f, 15-20 pose_f+1, 21-29 sb_f+1;
//      imuJ[f] = 465 packed lower Hessian entries + 30 gradient entries.
static void build_asm_plan(const BaLayout& L, std::vector<AsmPlanEntry>& plan) {
    plan.clear();
    if (L.big) return;
    const int K = L.K, Rc = L.Rc, R = L.R;
    auto tri = [](int a, int b) { return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a; };
    // the (at most two) IMU terms of an entry, sorted into (even factor, odd factor): the order the scatter rounds of rounds 2-4 added them in
    auto imu = [&](int f0, int i0, int f1, int i1) {
        int ev = 0, od = 0;
        auto put = [&](int f, int i) { if (f >= 0 && f <= K - 2 && i >= 0) ((f & 1) ? od : ev) = f * 512 + i + 1; };
        put(f0, i0); put(f1, i1);
        return ev | (od << 16);
    };
    for (int a = 0; a < Rc; ++a)
        for (int b = 0; b <= a; ++b) {
            AsmPlanEntry e;
            e.dst = L.l_S + a * (a + 1) / 2 + b;
            e.base = L.bo_Sp + a * (a + 1) / 2 + b;
            e.imu = 0;
            const int pa = a / 6, oa = a % 6, pb = b / 6, ob = b % 6;
            if (a < 6 * K) {                                      // frames with IMU factors (not the relocalisation pose, not ex / td)
                if (pa == pb) e.imu = imu(pa, tri(oa, ob), pa - 1, tri(15 + oa, 15 + ob));
                else if (pa == pb + 1) e.imu = imu(pb, tri(15 + oa, ob), -1, -1);
            }
            e.cols = (a << 16) | b;
            plan.push_back(e);
        }
    for (int k = 0; k < L.Rpad; ++k) {
        AsmPlanEntry e;
        e.dst = (L.l_vec + V_G * L.Rpad + k) | (1 << 28);
        e.base = k < Rc ? L.bo_gp + k : -1;
        e.imu = 0;
        if (k < 6 * K) e.imu = imu(k / 6, 465 + k % 6, k / 6 - 1, 465 + 15 + k % 6);
        else if (k >= Rc && k < R) e.imu = imu((k - Rc) / 9, 465 + 6 + (k - Rc) % 9, (k - Rc) / 9 - 1, 465 + 21 + (k - Rc) % 9);
        e.cols = k < R ? k : -1;
        plan.push_back(e);
    }
    for (int k = 0; k < K; ++k)
        for (int ra = 0; ra < 9; ++ra)
            for (int rb = 0; rb < 9; ++rb) {                      // D_k: full 9x9, both triangles
                AsmPlanEntry e;
                e.dst = L.l_D + 81 * k + 9 * ra + rb;
                e.base = -1;
                e.imu = imu(k, tri(6 + ra, 6 + rb), k - 1, tri(21 + ra, 21 + rb));
                e.cols = ((Rc + 9 * k + ra) << 16) | (Rc + 9 * k + rb);
                plan.push_back(e);
            }
    for (int k = 0; k < K; ++k)
        for (int ra = 0; ra < 9; ++ra)
            for (int rb = 0; rb < 9; ++rb) {                      // E_k: rows sb_k, columns sb_k-1 (factor k-1: sb_f+1 x sb_f); E_0 = 0
                AsmPlanEntry e;
                e.dst = L.l_E + 81 * k + 9 * ra + rb;
                e.base = -1;
                e.imu = k > 0 ? imu(k - 1, tri(21 + ra, 6 + rb), -1, -1) : 0;
                e.cols = k > 0 ? ((Rc + 9 * k + ra) << 16) | (Rc + 9 * (k - 1) + rb) : -1;
                plan.push_back(e);
            }
}

// ---- layout -------------------------------------------------------------------------------------
static int build_layout(vg_handle* h, int nwin, const vg_ba_problem* const* in, const PriorRef* pr, BaLayout& L) {
    memset(&L, 0, sizeof(L));
    const vg_ba_problem* p0 = in[0];
    L.nwin = nwin;
    L.K = p0->K;
    bool relo = false;
    int Lmax = 1, Fmax = 1, Omax = 1, Nmax = 0, NBmax = 0;
    for (int w = 0; w < nwin; ++w) {
        const vg_ba_problem* p = in[w];
        int rc = check_problem(h, p);
        if (rc) return rc;
        if (p->K != p0->K || (p->estimate_extrinsic != 0) != (p0->estimate_extrinsic != 0) ||
            (p->estimate_td != 0) != (p0->estimate_td != 0)) {
            h->err = "windows of one batch must share K / estimate_extrinsic / estimate_td";
            return VG_ERR_BAD_ARG;
        }
        relo = relo || p->relo_n > 0;
        int F = p->relo_n;
        for (int l = 0; l < p->L; ++l) F += p->lm_nobs[l] - 1;
        Lmax = std::max(Lmax, p->L);
        Fmax = std::max(Fmax, F);
        Omax = std::max(Omax, p->n_obs + p->relo_n);
        Nmax = std::max(Nmax, pr[w].n);
        NBmax = std::max(NBmax, pr[w].nb);
    }
    Lmax = std::max(Lmax, h->ba.res_L); Fmax = std::max(Fmax, h->ba.res_F); Omax = std::max(Omax, h->ba.res_O);      // vg_ba_reserve
    if (h->ba.res_N > Nmax) { Nmax = h->ba.res_N; NBmax = std::max(NBmax, L.K + 4); }
    L.Kp = L.K + (relo ? 1 : 0);
    L.e = p0->estimate_extrinsic ? 1 : 0;
    L.t = p0->estimate_td ? 1 : 0;
    L.Rc = 6 * L.Kp + 6 * L.e + L.t;
    L.RcPad = up(L.Rc + 1, 16);                  // + the augmented rhs row / column
    L.R = L.Rc + 9 * L.K;
    L.Rpad = up(L.R + 1, 8);
    L.Lcap = up(Lmax, 16);
    L.Fcap = up(Fmax, 16);
    L.Ocap = up(Omax, 8);
    L.Ncap = up(std::max(Nmax, 1), 8);
    L.NBcap = up(std::max(NBmax, 1), 8);
    L.REC = 28 + 12 * L.e + 2 * L.t;
    L.nst = up(7 * L.Kp + 9 * L.K + 8, 2);
    L.imu_info = h->ba.imu_info_mode;
    // one workgroup solves a window out of LDS while the camera part fits its tiling; wider windows (and windows the
    // caller shards over ranks) take the large-window path
    L.big = (L.RcPad > 96 || L.Rc > 127 || L.K + 1 > BA_MAX_K || h->ba.force_large) ? 1 : 0;
    if (L.big && L.Rc + 10 > 256) { h->err = "camera part wider than the large-window solve kernel's tiling"; return VG_ERR_UNSUPPORTED; }
    // ---- workgroups per window
    L.nbf = (L.Fcap + BA_LIN_NT - 1) / BA_LIN_NT;
    // IMU / prior linearisation: spread over workgroups only while all of them fit the chip at once (latency mode)
    {
        const int nsplit = (L.K - 1 + BA_IMU_GROUP - 1) / BA_IMU_GROUP;
        const bool split = (long)nwin * (nsplit + 1) <= 256;
        L.igs = split ? BA_IMU_GROUP : L.K - 1;
        L.nig = split ? nsplit : 1;
        L.nprw = split ? 1 : 0;
        L.pro_split = split ? 1 : 0;                   // (ba_prologue_kernel: its pieces side by side, see the kernel)
    }
    L.nbl = L.nbf + L.nig + L.nprw;
    {
        const int nb = L.Kp + L.e + L.t;
        L.ntask = nb * (nb + 1) / 2;
        const int per = BA_ACC_NT / 64;
        const int ng = L.e + L.t, nglob = ng * (ng + 1) / 2;      // blocks among ex / td: they visit EVERY factor
        // diagonal pose blocks get a workgroup each, the ex / td blocks share one (if present), wavefront tasks after them
        L.nba = L.Kp + (ng ? 1 : 0) + (L.ntask - L.Kp - nglob + (L.Lcap + 63) / 64 + per - 1) / per;
    }
    // ---- LDS carve of the solve kernel (large-window path: only S, red and 1/L_jj are LDS offsets, the others are
    //      offsets into the HBM scratch at so_bigm)
    int bigm_doubles = 0;
    {
        // XC leading dimension: rows p and p+1 of an MFMA operand read must fall on different halves of the 64 banks
        int ldc = L.RcPad;
        if ((2 * ldc) % 64 != 32) ldc += 16;
        L.ldc = ldc;
        int o = 0, ob = 0;
        int& oo = L.big ? ob : o;                    // where the movable arrays are carved from
        L.l_S = o; o += up((L.Rc + 1) * (L.Rc + 2) / 2, 2);
        L.l_XC = 0; L.l_ring = 0; L.l_pinv = 0;
        if (L.big) { L.l_XC = oo; oo += up(9 * L.K, 8) * ldc; }    // (large path: two k-steps per trip of schur_chain_big)
        else {
            // single-workgroup path (round 5): the coupling rows are not held for the whole solve any more (100 rows x ldc = 64 KB of
            // the former 134 KB carve).  Two slots of 18 rows -- the block pair being eliminated and the pair before it -- are all the
            // chain elimination reads; finished rows are parked in HBM (so_xp) for the back substitution.  The staged landmark tile
            // of the Schur sweep (l_wd) and the small scratch uses of `wd` alias the ring: they are only live outside the chain.
            L.l_ring = o; o += 36 * ldc;
        }
        L.l_D = oo; oo += up(81 * L.K, 2);
        L.l_E = oo; oo += up(81 * L.K, 2);
        L.l_dinv = oo; oo += up(9 * L.K, 2);
        L.l_vec = oo; oo += 9 * L.Rpad;
        L.l_red = o; o += 32;
        if (L.big) { L.l_wd = oo; oo += up(std::max(9 * L.K, 162), 2); }
        else L.l_wd = L.l_ring;                                    // [RcPad][32 + 1] <= 36 ldc;  SV_NT, 9 K, 162 doubles of scratch likewise
        L.l_z = oo; oo += up(36 * L.K, 2);
        if (L.big) { L.l_pmap = oo; oo += up(L.Ncap, 4) / 2; }
        else { L.l_pmap = 0; L.l_pinv = o; o += up(L.R, 4) / 2; }
        if (L.big) { L.l_di = o; o += up(L.Rc + 1, 2); L.l_cz = o; o += 6 * 96; }
        L.l_Sg = 0;
        if (L.big) { L.l_Sg = oo; oo += up((L.Rc + 1) * (L.Rc + 2) / 2, 2); }      // S between the halves of the reduced solve
        L.lds_solve = o * 8;
        bigm_doubles = ob;
        if (!L.big && L.Rc + 10 - 64 > 32) { h->err = "solve kernel: more column tasks than three wavefronts"; return VG_ERR_UNSUPPORTED; }
        if (!L.big && 36 * ldc < std::max(std::max(L.RcPad * 33, 2 * SV_NT), std::max(9 * L.K, 162))) { h->err = "solve kernel: scratch does not fit the coupling-row ring"; return VG_ERR_UNSUPPORTED; }
        if (L.lds_solve > 160 * 1024) { h->err = "solve kernel LDS carve exceeds 160 KB"; return VG_ERR_UNSUPPORTED; }
        const int nib = std::min(std::min(L.K - 1, BA_IMU_BATCH), L.igs);
        L.lds_lin = 8 * std::max(up(nib * 225, 2) + nib * 480, 5 * L.Ncap);
        L.lds_pro = 8 * BA_NW * (L.imu_info ? 512 : 256);      // imu_sqrt_info / imu_sqrt_info_ref: scratch per wavefront
        if (8 * L.Ncap * L.Ncap <= 128 * 1024) L.lds_pro = std::max(L.lds_pro, 8 * L.Ncap * L.Ncap);   // J0 staged in LDS when it fits
        if (L.lds_lin > 128 * 1024) { h->err = "prior too large for the linearisation kernel"; return VG_ERR_UNSUPPORTED; }
        // fused projection kernel (ba_linacc_proj_kernel): LDS = staged records [la_chf][33] | pair blocks [Kp (Kp - 1) / 2][90] |
        // keys [la_chf] + chunk starts [64] (ints)
        {
            static const bool env_off = getenv("VG_BA_FUSED") && !strcmp(getenv("VG_BA_FUSED"), "0");       // (development switches:
            const bool off = env_off && !h->ba.no_env;                                                          //  plain vg_create() only)
            // few windows: the chip is empty and the separate kernels spread a window over many workgroups (single window: 28 us per
            // round against 52 us for the one fused workgroup); from a few dozen windows on the fused kernel wins
            static const int env_min = getenv("VG_BA_FUSED_MIN") ? atoi(getenv("VG_BA_FUSED_MIN")) : 32;
            const int la_min_windows = h->ba.fused_min >= 0 ? (h->ba.fused_min ? h->ba.fused_min : 1 << 30) : (env_min ? env_min : 1 << 30);
            const int npair = L.Kp * (L.Kp - 1) / 2;
            // (the per-round kernel reports 8 KB of fixed group segment -- its non-inlined phases -- next to the dynamic LDS: 160 KB in all)
            const int avail = (160 * 1024 - 8192 - 512) / 8;
            const int nstl = up(7 * L.Kp + 9 * L.K + 8, 2);                                   // the staged state (round 6)
            int chf = (int)((avail - nstl - npair * 90 - 34 - (64 + 82 + 8 * 80) / 2) / 34.0);      // per factor: record 33 doubles + key + position (2 ints)
            chf = std::min(chf & ~1, 512);
            L.la_on = (!off && L.nwin >= la_min_windows && !L.big && !L.e && !L.t && L.Kp <= 13 && chf >= 96 && (L.Fcap + chf - 17) / (chf - 16) <= 62) ? 1 : 0;
            L.la_chf = chf; L.la_chq = chf - 16;
            L.la_P = up(chf * 33, 2);
            L.la_key = L.la_P + up(npair * 90, 2);
            L.la_x = up(L.la_key + (2 * chf + 64 + 82 + 8 * 80 + 1) / 2 + 2, 2);
            L.lds_linacc = (L.la_x + nstl) * 8;
        }
        // which build of the per-round solve kernel: 4 wavefronts per window (two windows per CU, the batch) or 8 (ba_solve_w8.hip: a
        // few windows on an otherwise empty chip).  Below 32 windows -- where the factor side is in its latency
        // mode as well -- the second window per CU does not exist and the wavefront slots go to the same window.
        {
            static const int env_max = getenv("VG_BA_SOLVE_W8_BELOW") ? atoi(getenv("VG_BA_SOLVE_W8_BELOW")) : -1;      // (development switch:
            const int below = (env_max >= 0 && !h->ba.no_env) ? env_max : 32;                                             //  plain vg_create() only)
            L.sv_w8 = (!L.big && L.nwin < below) ? 1 : 0;
            // (the 8-wavefront build stages the landmark tile of its Schur phase, [RcPad][32 + 1] doubles, behind the carve -- the ring it
            //  aliases in the 4-wavefront kernel is live beside it there; a shape that leaves no room keeps the 4-wavefront kernel)
            L.l_wd8 = L.lds_solve / 8;
            L.lds_solve_w8 = L.lds_solve + 8 * L.RcPad * 33;
            if (L.lds_solve_w8 > 160 * 1024) L.sv_w8 = 0;
        }
    }
    if (L.big) {
        const int nt = (L.Rc + 1 + 15) / 16;
        L.nts = nt * (nt + 1) / 2;
        const int ntri = L.Rc * (L.Rc + 1) / 2;
        L.rb1_T = up(ntri + L.Rc, 2);
        L.rb1_scal = L.rb1_T + up((L.Rc + 1) * (L.Rc + 2) / 2, 2);
        L.rb1_len = L.rb1_scal + RB1_NSCAL;
    }
    // ---- int arrays
    int o = 0;
    L.io_hdr = o; o += BA_HDR_INTS;
    L.io_lm_start = o; o += L.Lcap;
    L.io_lm_fbeg = o; o += L.Lcap + 8;
    L.io_fac_i = o; o += L.Fcap;
    L.io_fac_j = o; o += L.Fcap;
    L.io_fac_lm = o; o += L.Fcap;
    L.io_fac_oi = o; o += L.Fcap;
    L.io_fac_oj = o; o += L.Fcap;
    L.io_fac_slot = o; o += L.Fcap;
    L.io_pair_ptr = o; o += up(L.Kp * L.Kp + 1, 8);
    L.io_task_list = o; o += up(L.ntask, 8);
    L.io_imu_valid = o; o += up(L.K, 8);
    L.io_pb_kind = o; o += L.NBcap;
    L.io_pb_idx = o; o += L.NBcap;
    L.io_pb_col = o; o += L.NBcap;
    L.io_pb_off = o; o += L.NBcap;
    L.io_pb_x0off = o; o += L.NBcap;
    L.istride = up(o, 8);
    // ---- double inputs
    o = 0;
    L.do_pose = o; o += up(7 * L.Kp, 2);
    L.do_sb = o; o += up(9 * L.K, 2);
    L.do_ex = o; o += 8;
    L.do_td = o; o += 2;
    L.do_lam = o; o += L.Lcap;
    L.do_obs = o; o += L.Ocap * BA_OBS_STRIDE;
    L.do_imu = o; o += (L.K - 1) * BA_IMU_STRIDE;
    L.do_par = o; o += P_NPAR;
    L.dstride = up(o, 8);
    // ---- prior factor: a buffer of its own whose layout depends on K alone as long as the prior is one the marginalization
    //      can have produced (n <= 6K + 25, <= K + 4 blocks) -- so that a prior can stay where it is between frames
    L.pld = std::max(up(6 * L.K + 9 * 2 + 6 + 1, 8), L.Ncap);
    L.po_x0 = 0;
    L.po_r0 = 9 * std::max(up(L.K + 4, 8), L.NBcap);
    L.po_J0 = L.po_r0 + L.pld;
    L.pstride = up(L.po_J0 + L.pld * L.pld, 8);
    // ---- scratch
    o = 0;
    L.so_ctl = o; o += C_NCTL;
    L.so_part = o; o += up(L.nbl, 8);
    L.so_x = o; o += 2 * L.nst;
    L.so_lam = o; o += 2 * L.Lcap;
    L.so_imuU = o; o += up((L.K - 1) * 225, 2);
    L.so_Hp = o; o += L.Ncap * L.Ncap;
    L.so_J0t = o; o += L.Ncap * L.Ncap;
    L.so_rec = o; o += L.Fcap * L.REC;
    L.so_sc = o; o += L.Rpad;
    L.so_sl = o; o += L.Lcap;
    L.so_dg = o; o += L.Rpad + L.Lcap;
    L.so_gt = o; o += L.Rpad + L.Lcap;
    L.so_gn = o; o += L.Rpad + L.Lcap;
    L.so_yl = o; o += L.Lcap;
    L.so_lsc = o; o += L.Lcap;
    L.so_xp = 0;
    if (!L.big) { L.so_xp = o; o += up(9 * L.K, 4) * L.ldc; }
    if (L.big) {
        L.so_bigm = o; o += up(bigm_doubles, 8);
        L.so_dgl = o; o += 2 * L.Lcap;
        L.so_gtl = o; o += 2 * L.Lcap;
    }
    {
        int b = 0;
        L.bo_Sp = b; b += up(L.Rc * (L.Rc + 1) / 2, 2);
        L.bo_gp = b; b += L.RcPad;
        L.bo_h = b; b += L.Lcap;
        L.bo_b = b; b += L.Lcap;
        L.bo_Wt = b; b += L.RcPad * L.Lcap;
        L.bo_imuJ = b; b += (L.K - 1) * 512;       // per factor: 465 lower Hessian entries + 30 gradient entries
        L.bo_pr = b; b += L.Ncap;
        L.bo_gpr = b; b += L.Ncap;
        L.buf_stride = up(b, 8);
    }
    L.so_buf = o; o += 2 * L.buf_stride;
    L.sstride = up(o, 8);
    // gather plan of assemble_small behind the device copy of this struct (build_asm_plan)
    L.pl_off = (int)((sizeof(BaLayout) + 255) / 256 * 256);
    L.pl_n = L.big ? 0 : L.Rc * (L.Rc + 1) / 2 + L.Rpad + 2 * 81 * L.K;
    // ---- outputs
    o = 0;
    L.oo_pose = o; o += up(7 * L.Kp, 2);
    L.oo_sb = o; o += up(9 * L.K, 2);
    L.oo_ex = o; o += 8;
    L.oo_td = o; o += 2;
    L.oo_lam = o; o += L.Lcap;
    L.oo_sum = o; o += BA_SUM_DOUBLES;
    L.oo_trace = o; o += 5 * VG_MAX_ITERS + 16;
    L.ostride = up(o, 8);
    L.oi_stride = up(4 + VG_MAX_ITERS, 8);
    // ---- marginalization: kept dimension <= 6K + 2*9 + 6 + 1; blocks <= K + 4.  Both paths: the kernel's LDS tables are
    //      sized for BA_MAX_K_LARGE frames; a kept block wider than the LDS tile is factored in global memory.
    L.mcap = up(6 * L.K + 9 * 2 + 6 + 1, 8);
    if (L.mcap > 256) { h->err = "kept dimension of the marginalization beyond the kernel's tables"; return VG_ERR_UNSUPPORTED; }
    const int mcap = L.mcap;
    L.mo_J0 = 0;
    L.mo_r0 = mcap * mcap;
    L.mo_x0 = L.mo_r0 + mcap;
    L.mo_stride = up(L.mo_x0 + (L.K + 4) * 9, 8);
    L.mi_stride = up(8 + 2 * (L.K + 4) + 16, 8);   // + 16 phase stamps (BA_PROFILE builds)
    L.mg_posmax = up(mcap + 15 + L.Lcap, 2);
    {
        // LDS of the marginalization kernel: [eigM ld^2][eigV ld^2][cs 2 ld][red 16][state][ints 256]
        const int nstm = up(16 * L.K + 8 + 1, 2);
        L.mg_cs = up(std::max(std::max(2 * mcap, 3 * (L.mg_posmax / 2 + 2)), 2 * (3 * ((mcap + 2) / 2) + 2) + 2), 2);   // generic: one table
                                                     // of 3*half doubles; fast path: two (double-buffered)
        L.mg_cs = std::max(L.mg_cs, std::max(up(5 * mcap + 8, 2), up(L.Lcap + 512, 2)));   // vh_eig: running diagonal, norms, eigenvalues, flags; Amm block elimination: Lcap + 2 x 256
        const int fixed = 32 + nstm + BA_MARG_LDS_INTS / 2 + L.mg_cs;
        int ld = mcap + 1;                           // big enough for the kept part; also used for Amm when m <= ld.  ODD:
                                                     // the symmetric-storage Jacobi walks columns (stride ld doubles) and
                                                     // an even stride folds them onto a few LDS banks (96 -> one bank)
        while (2 * ld * ld + fixed > 160 * 1024 / 8) ld -= 2;
        L.mg_ld = ld;
        L.mg_lds_bytes = (2 * ld * ld + fixed) * 8;
        if (ld < 8) { h->err = "marginalization LDS carve failed"; return VG_ERR_UNSUPPORTED; }
        const long pm = L.mg_posmax;
        long sdoubles = pm * pm + pm + (long)L.Fcap * 42 + 2 * pm * (mcap + 1) + 2 * pm * pm + 2L * mcap * mcap + 2 * L.Ncap + 480 + L.Lcap + 64 + L.Lcap / 2 + 8;
        L.ms_stride = (int)up((int)sdoubles, 8);
    }
    return VG_OK;
}

// ---- packing of one window ------------------------------------------------------------------------
struct FacTmp { int i, j, l, oi, oj; };

static int pack_window(vg_handle* h, const BaLayout& L, const vg_ba_problem* p, const PriorRef& pr, int margin, int* ia, double* di, double* hp) {
    const int K = L.K, Kp = L.Kp;
    int* hdr = ia + L.io_hdr;
    // observations (+ relo rows)
    for (int o = 0; o < p->n_obs; ++o) {
        double* d = di + L.do_obs + (size_t)o * BA_OBS_STRIDE;
        for (int k = 0; k < 7; ++k) d[k] = p->obs[(size_t)o * 7 + k];
        d[7] = 0.0;
    }
    for (int k = 0; k < p->relo_n; ++k) {
        double* d = di + L.do_obs + (size_t)(p->n_obs + k) * BA_OBS_STRIDE;
        memset(d, 0, sizeof(double) * BA_OBS_STRIDE);
        d[0] = p->relo_xy[2 * k];
        d[1] = p->relo_xy[2 * k + 1];
    }
    // factor list, landmark-major; a landmark's relo factor (if any) follows its window factors
    std::vector<FacTmp> fac;
    std::vector<int> relo_of(p->L, -1);
    for (int k = 0; k < p->relo_n; ++k) relo_of[p->relo_lm[k]] = k;
    for (int l = 0; l < p->L; ++l) {
        ia[L.io_lm_start + l] = p->lm_start[l];
        ia[L.io_lm_fbeg + l] = (int)fac.size();
        const int s = p->lm_start[l], o = p->lm_obs_off[l];
        for (int k = 1; k < p->lm_nobs[l]; ++k) fac.push_back({s, s + k, l, o, o + k});
        if (relo_of[l] >= 0) fac.push_back({s, K, l, o, p->n_obs + relo_of[l]});
    }
    ia[L.io_lm_fbeg + p->L] = (int)fac.size();
    const int F = (int)fac.size();
    for (int f = 0; f < F; ++f) {
        ia[L.io_fac_i + f] = fac[f].i; ia[L.io_fac_j + f] = fac[f].j; ia[L.io_fac_lm + f] = fac[f].l;
        ia[L.io_fac_oi + f] = fac[f].oi; ia[L.io_fac_oj + f] = fac[f].oj;
    }
    // slot table: records sorted by (anchor i, target j) pair, so that the owner of an entry of the camera system walks
    // one contiguous slot range per pair
    {
        int* ptr = ia + L.io_pair_ptr;
        std::vector<int> count(Kp * Kp + 1, 0);
        for (int f = 0; f < F; ++f) count[fac[f].i * Kp + fac[f].j + 1]++;
        ptr[0] = 0;
        for (int k = 0; k < Kp * Kp; ++k) ptr[k + 1] = ptr[k] + count[k + 1];
        std::vector<int> cursor(ptr, ptr + Kp * Kp);
        for (int f = 0; f < F; ++f) ia[L.io_fac_slot + f] = cursor[fac[f].i * Kp + fac[f].j]++;
    }
    const int nchunk = 1;
    {
        // wavefront owner tasks: every block except the diagonal pose blocks and the blocks among ex / td, which have
        // workgroups of their own (packed-triangle index of the block)
        const int nb = Kp + L.e + L.t;
        int n = 0;
        for (int br = 0; br < nb; ++br)
            for (int bc = 0; bc <= br; ++bc)
                if (!(br == bc && br < Kp) && !(bc >= Kp)) ia[L.io_task_list + n++] = br * (br + 1) / 2 + bc;
    }
    hdr[H_L] = p->L; hdr[H_F] = F; hdr[H_NPRIOR] = pr.n; hdr[H_NBLK] = pr.n ? pr.nb : 0;
    hdr[H_MAXIT] = p->max_iters; hdr[H_NCHUNK] = nchunk; hdr[H_MARGIN] = margin; hdr[H_STATUS] = 0;
    hdr[H_MARGMODE] = h->ba.marg_mode;
    // state
    memcpy(di + L.do_pose, p->pose, sizeof(double) * 7 * K);
    if (Kp > K) {
        if (p->relo_n > 0) memcpy(di + L.do_pose + 7 * K, p->relo_pose, sizeof(double) * 7);
        else { double id[7] = {0, 0, 0, 0, 0, 0, 1}; memcpy(di + L.do_pose + 7 * K, id, sizeof(id)); }
    }
    memcpy(di + L.do_sb, p->speedbias, sizeof(double) * 9 * K);
    memcpy(di + L.do_ex, p->ex_pose, sizeof(double) * 7);
    di[L.do_td] = p->td;
    for (int l = 0; l < p->L; ++l) di[L.do_lam + l] = p->inv_depth[l];
    for (int k = 0; k < K - 1; ++k) {
        const vg_imu_preint& m = p->imu[k];
        double* d = di + L.do_imu + (size_t)k * BA_IMU_STRIDE;
        d[0] = m.sum_dt;
        memcpy(d + 1, m.delta_p, 24); memcpy(d + 4, m.delta_q, 32); memcpy(d + 8, m.delta_v, 24);
        memcpy(d + 11, m.linearized_ba, 24); memcpy(d + 14, m.linearized_bg, 24);
        memcpy(d + 17, m.jacobian, 225 * 8); memcpy(d + 242, m.covariance, 225 * 8);
        ia[L.io_imu_valid + k] = (m.valid && m.sum_dt <= 10.0) ? 1 : 0;
    }
    // prior: block table (where the rows / columns of J0 sit in the solver's ordering); the factor itself goes to the
    // window's slot of the prior buffer unless it is there already
    if (pr.n > 0) {
        const int n = pr.n;
        int off = 0, x0off = 0;
        for (int b = 0; b < pr.nb; ++b) {
            const int kind = pr.kind[b], idx = pr.idx[b];
            ia[L.io_pb_kind + b] = kind;
            ia[L.io_pb_idx + b] = idx;
            ia[L.io_pb_off + b] = off;
            ia[L.io_pb_x0off + b] = x0off;
            int col = -1;
            if (kind == VG_BLK_POSE) col = 6 * idx;
            else if (kind == VG_BLK_SPEEDBIAS) col = L.Rc + 9 * idx;
            else if (kind == VG_BLK_EXPOSE) col = L.e ? 6 * Kp : -1;
            else col = L.t ? 6 * Kp + 6 * L.e : -1;
            ia[L.io_pb_col + b] = col;
            off += blk_lsize(kind);
            x0off += blk_gsize(kind);
        }
        if (!pr.resident) {
            memcpy(hp + L.po_x0, p->prior_x0, sizeof(double) * x0off);
            memcpy(hp + L.po_r0, p->prior_r0, sizeof(double) * n);
            for (int r = 0; r < n; ++r) memcpy(hp + L.po_J0 + (size_t)r * L.pld, p->prior_J0 + (size_t)r * n, sizeof(double) * n);
        }
    }
    di[L.do_par + P_FOCAL] = p->focal; di[L.do_par + P_TR] = p->tr; di[L.do_par + P_ROW] = p->row;
    di[L.do_par + P_GNORM] = p->g_norm;
    di[L.do_par + P_MAXTIME] = (p->max_solver_time_s > 0.0 && !h->ba.allreduce) ? p->max_solver_time_s : 0.0;
    return VG_OK;
}

// windows are independent: a few host threads share the packing / unpacking of a batch (strided assignment)
static thread_local int tl_pack_cap = 0;      // vg_config::pack_threads of the handle whose call is running on this thread (0: not set)
static int host_threads_for(int nwin) {
    static const int env_cap = [] { const char* e = getenv("VG_PACK_THREADS"); const int v = e ? atoi(e) : 0; return v > 0 ? std::min(v, 64) : 8; }();
    return std::max(1, std::min(tl_pack_cap > 0 ? tl_pack_cap : env_cap, nwin / 16));      // (the handle's setting, else the development variable, else 8)
}
template <typename F>
static void for_windows(int nwin, F&& body) {            // body(thread, window)
    const int nthr = host_threads_for(nwin);
    if (nthr == 1) { for (int w = 0; w < nwin; ++w) body(0, w); return; }
    std::vector<std::thread> pool;
    for (int t = 0; t < nthr; ++t)
        pool.emplace_back([&, t] { for (int w = t; w < nwin; w += nthr) body(t, w); });
    for (auto& th : pool) th.join();
}

// ---- device buffers -------------------------------------------------------------------------------
template <typename T>
static int ensure(vg_handle* h, T*& ptr, size_t& cap, size_t need) {
    if (need <= cap && ptr) return VG_OK;
    if (ptr) HIPCHK(h, hipFree(ptr));
    ptr = nullptr; cap = 0;
    HIPCHK(h, hipMalloc((void**)&ptr, need * sizeof(T)));
    cap = need;
    return VG_OK;
}

extern "C" void ba_seq_release(vg_handle* h);        // (below: windows that stay on the device)

// ---- priors that stay on the device -----------------------------------------------------------------
// The first upload after a run that asks for a resident prior collects the block tables of that run's marginalization
// (a few dozen ints per window) and moves the factors device-to-device from the marginalization output into the prior slots.
static int resolve_resident(vg_handle* h, int nwin, const vg_ba_problem* const* in, std::vector<PriorRef>& pr, bool& carry) {
    BaBatch& B = h->ba;
    carry = false;
    bool any = false;
    for (int w = 0; w < nwin; ++w) any = any || in[w]->prior_n == VG_PRIOR_RESIDENT;
    if (any && B.mout_pending) {
        if (B.run_K != in[0]->K) { h->err = "resident prior: the window size changed since the run that produced it"; return VG_ERR_BAD_ARG; }
        HIPCHK(h, hipStreamSynchronize(h->stream));
        std::vector<int> mi((size_t)B.run_nwin * B.run_mi_stride);
        HIPCHK(h, hipMemcpy(mi.data(), B.P.miout, mi.size() * sizeof(int), hipMemcpyDeviceToHost));
        if ((int)B.slot.size() < B.run_nwin) B.slot.resize(B.run_nwin);
        for (int w = 0; w < B.run_nwin; ++w) {
            const int* m = mi.data() + (size_t)w * B.run_mi_stride;
            if (B.run_margin[w] == VG_MARGIN_NONE || !m[0]) continue;      // no new prior: the slot keeps what it holds
            BaBatch::PriorSlot& s = B.slot[w];
            s.n = m[1]; s.nb = m[3];
            s.kind.assign(m + 8, m + 8 + s.nb);
            s.idx.assign(m + 8 + (B.run_K + 4), m + 8 + (B.run_K + 4) + s.nb);
            carry = true;
        }
    }
    for (int w = 0; w < nwin; ++w) {
        const vg_ba_problem* p = in[w];
        PriorRef& r = pr[w];
        if (p->prior_n != VG_PRIOR_RESIDENT) {
            r.n = p->prior_n; r.nb = p->prior_n ? p->prior_nblocks : 0; r.kind = p->prior_block_kind; r.idx = p->prior_block_index;
            continue;
        }
        if (w >= (int)B.slot.size() || B.slot[w].n <= 0) {
            h->err = "VG_PRIOR_RESIDENT: window " + std::to_string(w) + " holds no prior on the device";
            return VG_ERR_BAD_ARG;
        }
        const BaBatch::PriorSlot& s = B.slot[w];
        r.n = s.n; r.nb = s.nb; r.kind = s.kind.data(); r.idx = s.idx.data(); r.resident = true;
    }
    return VG_OK;
}

extern "C" int vg_ba_batch_upload(vg_handle* h, int nwin, const vg_ba_problem* const* in, const int* margin_flags) {
    VG_RANGE("vg_ba_batch_upload");
    if (!h || nwin <= 0 || !in) return VG_ERR_BAD_ARG;
    for (int w = 0; w < nwin; ++w) if (!in[w]) return VG_ERR_BAD_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    BaBatch& B = h->ba;
    if (B.seq.active) {              // an ordinary upload takes the batch buffers back: the sequence ends here
        HIPCHK(h, hipStreamSynchronize(h->stream));
        ba_seq_release(h);
    }
    B.uploaded = false;              // a failed upload must not leave the previous batch's flags next to the new layout
    B.solved_recorded = false;
    static const bool debug_upload = getenv("VG_DEBUG_UPLOAD") != nullptr;      // phase times of this call on stderr
    const auto t_0 = std::chrono::steady_clock::now();
    auto t_prev = t_0;
    double t_ph[6] = {0, 0, 0, 0, 0, 0};
    auto stamp = [&](int k) { const auto n = std::chrono::steady_clock::now(); t_ph[k] += std::chrono::duration<double, std::micro>(n - t_prev).count(); t_prev = n; };
    std::vector<PriorRef> pr(nwin);
    bool carry = false, any_resident = false, any_host_prior = false;
    int rc = resolve_resident(h, nwin, in, pr, carry);
    stamp(0);
    if (rc) return rc;
    for (int w = 0; w < nwin; ++w) { any_resident = any_resident || pr[w].resident; any_host_prior = any_host_prior || (!pr[w].resident && pr[w].n > 0); }
    BaLayout L;
    rc = build_layout(h, nwin, in, pr.data(), L);
    if (rc) return rc;
    if ((any_resident || carry) && B.cap_pri && (B.slot_K != L.K || B.slot_po_r0 != L.po_r0 || B.slot_pld != L.pld || B.slot_pstride != L.pstride)) {
        h->err = "resident prior: the batch needs a different prior layout (window size or prior capacity changed)";
        return VG_ERR_UNSUPPORTED;
    }
    B.L = L;
    B.nwin = nwin;
    const size_t n_ia = (size_t)nwin * L.istride, n_di = (size_t)nwin * L.dstride, n_pri = (size_t)nwin * L.pstride;
    if (n_ia > B.hcap_ia) {
        if (B.h_ia) HIPCHK(h, hipHostFree(B.h_ia));
        B.h_ia = nullptr; B.hcap_ia = 0;
        HIPCHK(h, hipHostMalloc((void**)&B.h_ia, n_ia * sizeof(int), hipHostMallocDefault));
        B.hcap_ia = n_ia;
    }
    if (n_di > B.hcap_di) {
        if (B.h_di) HIPCHK(h, hipHostFree(B.h_di));
        B.h_di = nullptr; B.hcap_di = 0;
        HIPCHK(h, hipHostMalloc((void**)&B.h_di, n_di * sizeof(double), hipHostMallocDefault));
        B.hcap_di = n_di;
    }
    if (any_host_prior && n_pri > B.hcap_pri) {
        if (B.h_pri) HIPCHK(h, hipHostFree(B.h_pri));
        B.h_pri = nullptr; B.hcap_pri = 0;
        HIPCHK(h, hipHostMalloc((void**)&B.h_pri, n_pri * sizeof(double), hipHostMallocDefault));
        B.hcap_pri = n_pri;
    }
    B.flops = 0.0; B.flops_marg = 0.0; B.bytes_in = 0.0; B.bytes_out = 0.0;
    for (double& v : B.flops_k) v = 0.0;
    {
        // the marginalization scratch of a large window is 3 (Lcap + 6K + 40)^2 doubles: only when one is asked for
        bool any = false;
        for (int w = 0; w < nwin && margin_flags; ++w) any = any || margin_flags[w] != VG_MARGIN_NONE;
        if (!any && L.big) { L.ms_stride = 8; B.L.ms_stride = 8; }
        if (any && B.allreduce) { h->err = "marginalization of a landmark shard: every rank would need all frame-0 landmarks"; return VG_ERR_UNSUPPORTED; }
    }
    B.margin.assign(nwin, VG_MARGIN_NONE);
    B.nL.assign(nwin, 0);
    B.rounds = 0;
    for (int w = 0; w < nwin; ++w) B.rounds = std::max(B.rounds, in[w]->max_iters);
    B.rounds = std::max(B.rounds, 1);          // round 0 also evaluates the initial cost (max_iters = 0 windows)
    // pack: zero-fill + pack of every window's slabs
    {
        std::vector<int> rcs(64, VG_OK);
        tl_pack_cap = h->ba.pack_threads;
        for_windows(nwin, [&](int t, int w) {
            int* ia = B.h_ia + (size_t)w * L.istride;
            double* di = B.h_di + (size_t)w * L.dstride;
            memset(ia, 0, sizeof(int) * L.istride);
            memset(di, 0, sizeof(double) * L.dstride);
            const int mf = margin_flags ? margin_flags[w] : VG_MARGIN_NONE;
            const int r = pack_window(h, L, in[w], pr[w], mf, ia, di, B.h_pri ? B.h_pri + (size_t)w * L.pstride : nullptr);
            if (r != VG_OK && rcs[t] == VG_OK) rcs[t] = r;
        });
        for (int r : rcs) if (r != VG_OK) return r;
    }
    stamp(1);
    for (int w = 0; w < nwin; ++w) {
        const int mf = margin_flags ? margin_flags[w] : VG_MARGIN_NONE;
        B.margin[w] = mf;
        B.nL[w] = in[w]->L;
        // algorithmic flop / byte model of SURVEY.md 8(d)
        const vg_ba_problem* p = in[w];
        double F = p->relo_n, schur = 0.0, sumn = 0.0;
        for (int l = 0; l < p->L; ++l) {
            const double n = p->lm_nobs[l];
            F += n - 1;
            schur += (6 * n) * (6 * n + 1) + 12 * n;
            sumn += n;
        }
        const double np = pr[w].n, R = L.R;
        // per launch class (include/vinsgpu.h VG_BA_KERNEL_*), summed over the max_iters rounds of this window
        const double it = p->max_iters;
        const double f_lin = it * (F * 750 + 10 * 37000.0 + 4 * np * np + F * 145 + 10 * 9000.0);    // factor evaluation (+ the step evaluation it replaces)
        const double f_acc = it * (F * 416);                                                       // J^T J / J^T r of the projection factors
        const double f_sol = it * (schur + R * R * R / 3 + 2 * R * R + 12 * sumn);                  // Schur complement, Cholesky, back substitution
        const double f_pro = 2 * np * np * np;                                                     // prior J0^T J0
        // (fused projection kernel: the factor evaluation moves to the class of the kernel that now does it)
        const double f_move = L.la_on ? it * (F * 750 + 10 * 37000.0 + 4 * np * np) : 0.0;
        B.flops_k[0] += f_pro; B.flops_k[1] += f_lin - f_move;
        B.flops_k[2] += f_acc + f_move;
        if (L.big) {
            // large-window path: the landmark Schur complement has a kernel of its own
            B.flops_k[VG_BA_KERNEL_BIG_SCHUR] += it * schur;
            B.flops_k[VG_BA_KERNEL_BIG_SOLVE] += f_sol - it * schur;
        } else {
            B.flops_k[3] += f_sol;
        }
        double fl = f_lin + f_acc + f_sol + f_pro;
        if (mf == VG_MARGIN_OLD) {
            double m = 15;
            for (int l = 0; l < p->L; ++l) m += (p->lm_start[l] == 0);
            const double n = 6.0 * (L.K - 1) + 9 + 6 + L.t;
            const double fm = 9 * (m * m * m + n * n * n) + 2 * (m * m * n + m * n * n);
            fl += fm;
            B.flops_marg += fm;
            B.flops_k[5] += fm;
        }
        B.flops += fl;
        B.bytes_in += 8.0 * (16 * L.K + 8 + p->L + 7.0 * p->n_obs + (L.K - 1) * 467.0 + np * np + 2 * np) + 12.0 * p->L;   // (a resident prior is still read by the kernels)
        B.bytes_out += 8.0 * (16 * L.K + 8 + p->L) + (mf != VG_MARGIN_NONE ? 8.0 * (75.0 * 75 + 75 + 100) : 0.0);
    }
    stamp(2);
    rc = ensure(h, B.P.iarr, B.cap_ia, (size_t)nwin * L.istride); if (rc) return rc;
    rc = ensure(h, B.P.din, B.cap_di, (size_t)nwin * L.dstride); if (rc) return rc;
    rc = ensure(h, B.P.scr, B.cap_sc, (size_t)nwin * L.sstride); if (rc) return rc;
    rc = ensure(h, B.P.out, B.cap_out, (size_t)nwin * L.ostride); if (rc) return rc;
    rc = ensure(h, B.P.iout, B.cap_iout, (size_t)nwin * L.oi_stride); if (rc) return rc;
    rc = ensure(h, B.P.mscr, B.cap_mscr, (size_t)nwin * L.ms_stride); if (rc) return rc;
    // prior slots: growing keeps what the slots hold (same layout: slot w stays at w * pstride)
    {
        const size_t need = std::max(n_pri, carry ? (size_t)B.run_nwin * L.pstride : (size_t)0);
        if (need > B.cap_pri || !B.P.pri) {
            double* fresh = nullptr;
            HIPCHK(h, hipMalloc((void**)&fresh, need * sizeof(double)));
            if (B.P.pri && B.cap_pri && B.slot_pstride == L.pstride)
                HIPCHK(h, hipMemcpy(fresh, B.P.pri, std::min(B.cap_pri, need) * sizeof(double), hipMemcpyDeviceToDevice));
            if (B.P.pri) HIPCHK(h, hipFree(B.P.pri));
            B.P.pri = fresh; B.cap_pri = need;
        }
    }
    if (B.slot_K != L.K || B.slot_po_r0 != L.po_r0 || B.slot_pld != L.pld || B.slot_pstride != L.pstride) {
        B.slot.clear();              // another layout: nothing the slots held can be addressed any more
        B.slot_K = L.K; B.slot_po_r0 = L.po_r0; B.slot_pld = L.pld; B.slot_pstride = L.pstride;
    }
    if (carry) {
        hipError_t e = ba_launch_carry_prior(B.run_nwin, B.P.mout, B.P.miout, B.run_mo_J0, B.run_mo_r0, B.run_mo_x0, B.run_mo_stride,
                                             B.run_mi_stride, B.run_mcap, 9 * (B.run_K + 4), B.P.pri, L.po_x0, L.po_r0, L.po_J0, L.pld,
                                             L.pstride, h->stream);
        if (e != hipSuccess) { h->err = std::string("launch of ba_carry_prior_kernel: ") + hipGetErrorString(e); return VG_ERR_HIP; }
        HIPCHK(h, hipStreamSynchronize(h->stream));      // mout / miout may be re-allocated and are cleared below
        B.mout_pending = false;
    }
    stamp(3);
    rc = ensure(h, B.P.mout, B.cap_mout, (size_t)nwin * L.mo_stride); if (rc) return rc;
    rc = ensure(h, B.P.miout, B.cap_miout, (size_t)nwin * L.mi_stride); if (rc) return rc;
    if (L.big) {
        rc = ensure(h, B.P.rb1, B.cap_rb1, (size_t)nwin * L.rb1_len); if (rc) return rc;
        rc = ensure(h, B.P.rb2, B.cap_rb2, (size_t)nwin * RB2_LEN); if (rc) return rc;
        HIPCHK(h, hipMemsetAsync(B.P.rb1, 0, (size_t)nwin * L.rb1_len * sizeof(double), h->stream));
        HIPCHK(h, hipMemsetAsync(B.P.rb2, 0, (size_t)nwin * RB2_LEN * sizeof(double), h->stream));
    }
    // device copy of the layout, the gather plan of assemble_small behind it (frame after frame neither changes)
    {
        const size_t pl_off = (size_t)B.L.pl_off;
        const size_t need = pl_off + (size_t)B.L.pl_n * sizeof(AsmPlanEntry);
        if (!B.dL || B.dL_bytes < need) {
            if (B.dL) { HIPCHK(h, hipStreamSynchronize(h->stream)); (void)hipFree(B.dL); B.dL = nullptr; }
            HIPCHK(h, hipMalloc((void**)&B.dL, need));
            B.dL_bytes = need; B.dL_valid = false;
        }
        if (!B.dL_valid || memcmp(&B.dL_host, &B.L, sizeof(BaLayout)) != 0) {
            std::vector<AsmPlanEntry> plan;
            build_asm_plan(B.L, plan);
            if ((int)plan.size() != B.L.pl_n) { h->err = "gather plan size mismatch"; return VG_ERR_UNSUPPORTED; }
            HIPCHK(h, hipMemcpyAsync(B.dL, &B.L, sizeof(BaLayout), hipMemcpyHostToDevice, h->stream));
            if (B.L.pl_n) HIPCHK(h, hipMemcpyAsync((char*)B.dL + pl_off, plan.data(), plan.size() * sizeof(AsmPlanEntry), hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));                        // (pageable sources)
            B.dL_host = B.L; B.dL_valid = true;
        }
    }
    HIPCHK(h, hipMemcpyAsync(B.P.iarr, B.h_ia, n_ia * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(B.P.din, B.h_di, n_di * sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (any_host_prior) {
        // priors that come from the host: the used part of every slot (x0, r0 and the first n rows of J0)
        if (!any_resident) {
            int nmax = 0;
            for (int w = 0; w < nwin; ++w) nmax = std::max(nmax, pr[w].n);
            const size_t used = (size_t)L.po_J0 + (size_t)nmax * L.pld;
            if (nwin == 1) HIPCHK(h, hipMemcpyAsync(B.P.pri, B.h_pri, used * sizeof(double), hipMemcpyHostToDevice, h->stream));
            else HIPCHK(h, hipMemcpy2DAsync(B.P.pri, (size_t)L.pstride * sizeof(double), B.h_pri, (size_t)L.pstride * sizeof(double),
                                            used * sizeof(double), nwin, hipMemcpyHostToDevice, h->stream));
        } else {
            for (int w = 0; w < nwin; ++w)
                if (!pr[w].resident && pr[w].n > 0)
                    HIPCHK(h, hipMemcpyAsync(B.P.pri + (size_t)w * L.pstride, B.h_pri + (size_t)w * L.pstride,
                                             ((size_t)L.po_J0 + (size_t)pr[w].n * L.pld) * sizeof(double), hipMemcpyHostToDevice, h->stream));
        }
    }
    // (the output slabs are cleared by the prologue kernel of every run)
    stamp(4);
    HIPCHK(h, hipStreamSynchronize(h->stream));
    stamp(5);
    if (debug_upload)
        fprintf(stderr, "[upload] %d windows: resolve %.0f us, layout + pack %.0f, model %.0f, alloc + carry %.0f, enqueue %.0f, wait %.0f | bytes: ints %.2f MB, doubles %.2f MB, priors %s\n",
                nwin, t_ph[0], t_ph[1], t_ph[2], t_ph[3], t_ph[4], t_ph[5], n_ia * 4e-6, n_di * 8e-6, any_host_prior ? "host" : (any_resident ? "resident" : "none"));
    // what the slots hold now
    if ((int)B.slot.size() < nwin) B.slot.resize(nwin);
    for (int w = 0; w < nwin; ++w) {
        if (pr[w].resident) continue;
        BaBatch::PriorSlot& sl = B.slot[w];
        sl.n = pr[w].n; sl.nb = pr[w].nb;
        sl.kind.assign(pr[w].kind, pr[w].kind + (pr[w].n ? pr[w].nb : 0));
        sl.idx.assign(pr[w].idx, pr[w].idx + (pr[w].n ? pr[w].nb : 0));
    }
    B.any_margin = false;
    for (int w = 0; w < nwin; ++w) B.any_margin = B.any_margin || B.margin[w] != VG_MARGIN_NONE;
    B.solved_recorded = false;
    B.uploaded = true;
    return VG_OK;
}

// a marginalization run leaves new priors in mout / miout: remember how to find them (resolve_resident)
static void note_marg_run(BaBatch& B) {
    B.mout_pending = true;
    B.run_margin = B.margin;
    B.run_nwin = B.nwin; B.run_K = B.L.K;
    B.run_mo_J0 = B.L.mo_J0; B.run_mo_r0 = B.L.mo_r0; B.run_mo_x0 = B.L.mo_x0; B.run_mo_stride = B.L.mo_stride;
    B.run_mi_stride = B.L.mi_stride; B.run_mcap = B.L.mcap;
}

// all launches of one batch solve on h->stream (single-workgroup pipeline or the large-window path with its all-reduce hook)
#define BA_BIG_SLACK 8     // extra rounds of the large-window path: one round per attempted factorisation (mu escalations)
static int launch_solve(vg_handle* h, hipEvent_t* ev = nullptr, int* kinds = nullptr, int* n_launches = nullptr) {
    BaBatch& B = h->ba;
    hipError_t e;
    if (B.L.big) {
        int hook_rc = 0;
        e = ba_launch_solve_big(B.L, B.dL, B.P, B.rounds + BA_BIG_SLACK, BA_BIG_SLACK, h->stream, (BaAllReduce)B.allreduce, B.allreduce_user, &hook_rc,
                                ev, kinds, n_launches);
        if (e != hipSuccess && hook_rc) { h->err = "all-reduce hook returned " + std::to_string(hook_rc); return VG_ERR_HIP; }
    } else {
        // (forking the IMU / prior kernel onto h->aux was measured: the event record / wait pairs cost more than the ~30 us of
        //  overlap they buy — 2.72 vs 2.69 ms per 256-window solve — so the launches stay on one stream)
        e = ba_launch_solve(B.L, B.dL, B.P, B.rounds, h->stream, ev, kinds, n_launches, nullptr);
    }
    if (e != hipSuccess) { h->err = std::string("launch of ") + ba_failed_launch() + ": " + hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}

// ---- the solve pipeline as a hipGraph ---------------------------------------------------------------------------------
// One batch solve is 4 * rounds + 4 launches whose arguments (device layout block, buffer pointers, cost_only) do not change
// from run to run; in VG_LAUNCH_GRAPH mode they are captured from the handle's stream once and replayed with a single
// hipGraphLaunch.  The key holds everything a launch's grid / block / LDS size and arguments are computed from; what the
// kernels READ (layout block, tables, states) is data, not part of the graph.  The large-window path stays on direct
// launches (its all-reduce hook is arbitrary host code).  If the runtime cannot capture (hipStreamBeginCapture fails), the
// handle stays on direct launches: the same kernels either way, never another compute path.
static int resolve_launch_mode(BaBatch& B) {
    if (B.launch_mode < 0) {
        const char* e = B.no_env ? nullptr : getenv("VG_BA_LAUNCH_MODE");
        B.launch_mode = VG_LAUNCH_DEFAULT;
        if (e && !strcmp(e, "graph")) B.launch_mode = VG_LAUNCH_GRAPH;
        if (e && !strcmp(e, "direct")) B.launch_mode = VG_LAUNCH_DIRECT;
    }
    return B.launch_mode;
}
static void graph_key(const BaBatch& B, BaBatch::GraphKey& k) {
    const BaLayout& L = B.L;
    k.dL = B.dL; k.P = B.P;
    // (la_on / lds_linacc: vg_ba_set_fused_min_windows can flip the kernel sequence while every buffer and size stays the same)
    const int d[12] = {L.nwin, L.nig, L.nprw, L.nbf, L.nba, L.lds_pro, L.lds_lin, L.lds_solve, B.rounds, L.la_on, L.lds_linacc, L.big};
    memcpy(k.dims, d, sizeof(d));
}
static void graph_drop(BaBatch& B) {
    if (B.gexec) (void)hipGraphExecDestroy(B.gexec);
    B.gexec = nullptr;
}
extern "C" void ba_graph_release(vg_handle* h) { graph_drop(h->ba); }
static int launch_solve_graph(vg_handle* h) {
    BaBatch& B = h->ba;
    BaBatch::GraphKey k;
    graph_key(B, k);
    if (!B.gexec || memcmp(&k, &B.gkey, sizeof(k)) != 0) {
        graph_drop(B);
        HIPCHK(h, ba_prepare_launch());
        if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            B.graph_unavailable = true;
            return launch_solve(h);
        }
        const int rc = launch_solve(h);
        hipGraph_t g = nullptr;
        const hipError_t ec = hipStreamEndCapture(h->stream, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        HIPCHK(h, ec);
        const hipError_t ei = hipGraphInstantiate(&B.gexec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (ei != hipSuccess) { B.gexec = nullptr; HIPCHK(h, ei); }
        B.gkey = k;
        ++B.n_graph_captures;
    }
    HIPCHK(h, hipGraphLaunch(B.gexec, h->stream));
    ++B.n_graph_launches;
    return VG_OK;
}
static int launch_solve_in_mode(vg_handle* h) {
    BaBatch& B = h->ba;
    const bool graph = resolve_launch_mode(B) == VG_LAUNCH_GRAPH && !B.L.big && !B.graph_unavailable;
    return graph ? launch_solve_graph(h) : launch_solve(h);
}
extern "C" int vg_ba_set_launch_mode(vg_handle* h, int mode) {
    if (!h || (mode != VG_LAUNCH_DIRECT && mode != VG_LAUNCH_GRAPH)) return VG_ERR_BAD_ARG;
    h->ba.launch_mode = mode;
    if (mode == VG_LAUNCH_DIRECT) graph_drop(h->ba);
    return VG_OK;
}
extern "C" int vg_ba_set_fused_min_windows(vg_handle* h, int min_windows) {
    if (!h || min_windows < 0) return VG_ERR_BAD_ARG;
    h->ba.fused_min = min_windows;
    h->ba.uploaded = false;
    return VG_OK;
}
extern "C" int vg_ba_batch_is_fused(vg_handle* h) {
    if (!h || !h->ba.uploaded) return VG_ERR_BAD_ARG;
    return h->ba.L.la_on ? 1 : 0;
}
extern "C" int vg_ba_launch_stats(vg_handle* h, int* mode, long long* graph_launches, long long* graph_captures) {
    if (!h) return VG_ERR_BAD_ARG;
    if (mode) *mode = h->ba.graph_unavailable ? VG_LAUNCH_DIRECT : resolve_launch_mode(h->ba);
    if (graph_launches) *graph_launches = h->ba.n_graph_launches;
    if (graph_captures) *graph_captures = h->ba.n_graph_captures;
    return VG_OK;
}

extern "C" int vg_ba_batch_run_async(vg_handle* h) {
    VG_RANGE("vg_ba_batch_run_async");
    if (!h || !h->ba.uploaded) return VG_ERR_BAD_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    BaBatch& B = h->ba;
    const int rc = launch_solve_in_mode(h);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));           // states final: vg_ba_batch_download_state waits for this only
    B.solved_recorded = true;
    if (B.any_margin) { HIPCHK(h, ba_launch_marg(B.L, B.dL, B.P, h->stream)); note_marg_run(B); }
    return VG_OK;
}

// Large-window path (BASELINE configs[4]): force it for windows that would fit the single-workgroup pipeline (tests, sharded
// small windows) and install the hook that sums the reduce buffers over the ranks.  Both take effect at the next upload / run.
extern "C" int vg_ba_set_large_window(vg_handle* h, int force) {
    if (!h) return VG_ERR_BAD_ARG;
    h->ba.force_large = force != 0;
    h->ba.uploaded = false;
    return VG_OK;
}
// Form of the new prior factor (see sqrt_factor in ba_marg.hip).  Takes effect at the next upload.
extern "C" int vg_ba_set_marg_mode(vg_handle* h, int mode) {
    if (!h || (mode != VG_MARG_SQRT && mode != VG_MARG_EIGEN)) return VG_ERR_BAD_ARG;
    h->ba.marg_mode = mode;
    h->ba.uploaded = false;
    return VG_OK;
}
// Form of the IMU factors' sqrt_info (imu_sqrt_info / imu_sqrt_info_ref in ba_pipeline.hip).  Takes effect at the next upload.
extern "C" int vg_ba_set_imu_info_mode(vg_handle* h, int mode) {
    if (!h || (mode != VG_IMU_INFO_FACTOR && mode != VG_IMU_INFO_REFERENCE)) return VG_ERR_BAD_ARG;
    h->ba.imu_info_mode = mode;
    h->ba.uploaded = false;
    return VG_OK;
}
extern "C" int vg_ba_set_allreduce(vg_handle* h, vg_allreduce_fn fn, void* user) {
    if (!h) return VG_ERR_BAD_ARG;
    h->ba.allreduce = (void*)fn;
    h->ba.allreduce_user = user;
    return VG_OK;
}
extern "C" int vg_ba_reduce_layout(vg_handle* h, size_t* count1, size_t* count2) {
    if (!h || !h->ba.uploaded || !h->ba.L.big) return VG_ERR_BAD_ARG;
    if (count1) *count1 = (size_t)h->ba.nwin * h->ba.L.rb1_len;
    if (count2) *count2 = (size_t)h->ba.nwin * RB2_LEN;
    return VG_OK;
}

extern "C" int vg_ba_batch_run_timed(vg_handle* h, float* solve_ms, float* marg_ms) {
    if (!h || !h->ba.uploaded) return VG_ERR_BAD_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    BaBatch& B = h->ba;
    hipEvent_t e0 = h->ev0, e1 = h->ev1, e2 = h->ev2;
    HIPCHK(h, hipEventRecord(e0, h->stream));
    {
        const int rc = launch_solve_in_mode(h);
        if (rc) return rc;
    }
    HIPCHK(h, hipEventRecord(e1, h->stream));
    if (B.any_margin) { HIPCHK(h, ba_launch_marg(B.L, B.dL, B.P, h->stream)); note_marg_run(B); }
    HIPCHK(h, hipEventRecord(e2, h->stream));
    HIPCHK(h, hipEventSynchronize(e2));
    float a = 0, b = 0;
    HIPCHK(h, hipEventElapsedTime(&a, e0, e1));
    HIPCHK(h, hipEventElapsedTime(&b, e1, e2));
    if (solve_ms) *solve_ms = a;
    if (marg_ms) *marg_ms = b;
    return VG_OK;
}

// One batch run with a HIP event after every launch: ms[k] / n[k] = summed duration / number of launches of kernel
// class k (VG_BA_KERNEL_*).  The gaps between launches are part of the class that follows them.
extern "C" int vg_ba_batch_run_profiled(vg_handle* h, float* ms, int* n) {
    if (!h || !h->ba.uploaded || !ms || !n) return VG_ERR_BAD_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    BaBatch& B = h->ba;
    const int nev = B.L.big ? 9 * (B.rounds + BA_BIG_SLACK) + 8 : 4 * B.rounds + 5 + 1;
    std::vector<hipEvent_t> ev(nev, nullptr);
    std::vector<int> kinds(nev, 0);
    for (int i = 0; i < nev; ++i) HIPCHK(h, hipEventCreate(&ev[i]));
    int nl = 0;
    {
        const int rc = launch_solve(h, ev.data(), kinds.data(), &nl);
        if (rc) return rc;
    }
    if (B.any_margin) {
        HIPCHK(h, ba_launch_marg(B.L, B.dL, B.P, h->stream));
        note_marg_run(B);
        HIPCHK(h, hipEventRecord(ev[nl + 1], h->stream));
        kinds[nl] = VG_BA_KERNEL_MARG;
        ++nl;
    }
    HIPCHK(h, hipEventSynchronize(ev[nl]));
    for (int k = 0; k < VG_BA_KERNEL_COUNT; ++k) { ms[k] = 0.f; n[k] = 0; }
    for (int i = 0; i < nl; ++i) {
        float t = 0.f;
        HIPCHK(h, hipEventElapsedTime(&t, ev[i], ev[i + 1]));
        ms[kinds[i] & 0xff] += t;                       // (0x100: a further launch of the same class and round -- the reduced solve of the
        if (!(kinds[i] & 0x100)) n[kinds[i] & 0xff] += 1;  //  large-window path is three launches: time added, launch not counted)
    }
    for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
    return VG_OK;
}

extern "C" int vg_ba_batch_flops(vg_handle* h, double* solve_flops, double* marg_flops) {
    if (!h || !h->ba.uploaded) return VG_ERR_BAD_ARG;
    if (solve_flops) *solve_flops = h->ba.flops - h->ba.flops_marg;
    if (marg_flops) *marg_flops = h->ba.flops_marg;
    return VG_OK;
}

extern "C" int vg_ba_batch_flops_by_kernel(vg_handle* h, double* flops) {
    if (!h || !h->ba.uploaded || !flops) return VG_ERR_BAD_ARG;
    for (int k = 0; k < VG_BA_KERNEL_COUNT; ++k) flops[k] = h->ba.flops_k[k];
    return VG_OK;
}

extern "C" int vg_ba_batch_info(vg_handle* h, double* flops, double* bytes_in, double* bytes_out, int* lds_bytes) {
    if (!h || !h->ba.uploaded) return VG_ERR_BAD_ARG;
    if (flops) *flops = h->ba.flops;
    if (bytes_in) *bytes_in = h->ba.bytes_in;
    if (bytes_out) *bytes_out = h->ba.bytes_out;
    if (lds_bytes) *lds_bytes = h->ba.L.lds_solve;
    return VG_OK;
}

// ---- results.  The states (and the solver summaries) are final once ba_final_kernel has run; the marginalization kernel
// that follows on the stream only READS them.  A caller that needs the prior later than the states (Estimator::optimization():
// the prior is consumed by the NEXT frame's optimization, estimator.cpp:703-709) collects them separately:
//   vg_ba_batch_download_state  waits for the event recorded behind ba_final_kernel and copies on the second stream, so it
//                               returns while the marginalization kernel is still running;
//   vg_ba_batch_download_prior  waits for the whole stream.
// vg_ba_batch_download = both, in order.
static int unpack_states(vg_handle* h, int nwin, vg_ba_state* const* st, vg_ba_summary* sum) {
    BaBatch& B = h->ba;
    const BaLayout& L = B.L;
    std::vector<int> worst_t(64, VG_OK);
    tl_pack_cap = h->ba.pack_threads;
    for_windows(nwin, [&](int t, int w) {
        int& worst = worst_t[t];
        const double* o = B.h_out.data() + (size_t)w * L.ostride;
        const int* io = B.h_iout.data() + (size_t)w * L.oi_stride;
        if (st && st[w]) {
            vg_ba_state* s = st[w];
            if (s->pose) memcpy(s->pose, o + L.oo_pose, sizeof(double) * 7 * L.K);
            if (s->speedbias) memcpy(s->speedbias, o + L.oo_sb, sizeof(double) * 9 * L.K);
            if (s->ex_pose) memcpy(s->ex_pose, o + L.oo_ex, sizeof(double) * 7);
            if (s->td) *s->td = o[L.oo_td];
            if (s->inv_depth) memcpy(s->inv_depth, o + L.oo_lam, sizeof(double) * B.nL[w]);
            if (s->relo_pose && L.Kp > L.K) memcpy(s->relo_pose, o + L.oo_pose + 7 * L.K, sizeof(double) * 7);
        }
        if (io[0] != VG_OK) worst = io[0];
        if (sum) {
            vg_ba_summary& s = sum[w];
            memset(&s, 0, sizeof(s));
            s.status = io[0]; s.termination = io[1]; s.num_iterations = io[2]; s.num_accepted = io[3];
            s.initial_cost = o[L.oo_sum + 0]; s.final_cost = o[L.oo_sum + 1]; s.final_radius = o[L.oo_sum + 2];
            for (int k = 0; k < 9; ++k) s.gauge_rot[k] = o[L.oo_sum + 3 + k];
            for (int k = 0; k < 3; ++k) s.gauge_p0[k] = o[L.oo_sum + 12 + k];
            for (int k = 0; k < VG_MAX_ITERS; ++k) {
                s.it_cost[k] = o[L.oo_trace + 0 * VG_MAX_ITERS + k];
                s.it_cost_cand[k] = o[L.oo_trace + 1 * VG_MAX_ITERS + k];
                s.it_model[k] = o[L.oo_trace + 2 * VG_MAX_ITERS + k];
                s.it_radius[k] = o[L.oo_trace + 3 * VG_MAX_ITERS + k];
                s.it_step_norm[k] = o[L.oo_trace + 4 * VG_MAX_ITERS + k];
                s.it_flags[k] = io[4 + k];
            }
            for (int k = 0; k < 16; ++k) s.prof[k] = o[L.oo_trace + 5 * VG_MAX_ITERS + k];
        }
    });
    int worst = VG_OK;
    for (int v : worst_t) if (v != VG_OK) worst = v;
    return worst;
}

static int unpack_priors(vg_handle* h, int nwin, vg_ba_prior* const* pri) {
    BaBatch& B = h->ba;
    const BaLayout& L = B.L;
    static const bool debug_marg = getenv("VG_DEBUG_MARG") != nullptr;      // phase stamps of -DBA_PROFILE builds
    std::vector<int> too_small(64, 0);
    tl_pack_cap = h->ba.pack_threads;
    for_windows(nwin, [&](int t, int w) {
        if (pri && pri[w]) {
            vg_ba_prior* q = pri[w];
            q->n = q->m = q->nblocks = q->valid = 0;
            if (B.margin[w] != VG_MARGIN_NONE && B.any_margin) {
                const double* mo = B.h_mout.data() + (size_t)w * L.mo_stride;
                const int* mi = B.h_miout.data() + (size_t)w * L.mi_stride;
                q->valid = mi[0];
                if (debug_marg) {
                    const int* pf = mi + 8 + 2 * (L.K + 4);
                    fprintf(stderr, "[marg] eig1: sweeps=%d attempts=%d | eig2: sweeps=%d attempts=%d | kernel kcyc=%d | stamps: setup %d prior %d imu %d proj %d eig1 %d schur %d eig2 %d out %d\n",
                            mi[4] & 255, mi[4] >> 8, mi[5] & 255, mi[5] >> 8, mi[7], pf[0], pf[1] - pf[0], pf[2] - pf[1], pf[3] - pf[2], pf[4] - pf[3], pf[5] - pf[4], pf[6] - pf[5], pf[7] - pf[6]);
                    fprintf(stderr, "[marg] projection part (sqrt mode): offsets %d evaluate %d sort %d camera (wave 0) %d landmark rows %d\n",
                            pf[8] - pf[2], pf[9] - pf[8], pf[10] - pf[9], pf[11] - pf[10], pf[3] - pf[11]);
                    fprintf(stderr, "[marg] elimination (direct path): landmarks leave (MFMA, global operands) %d | Cholesky + inverse %d | X, Y, P'^-1 %d | certificate %d | kept tiles %d\n",
                            pf[12] - pf[3], pf[13] - pf[12], pf[14] - pf[13], pf[15] - pf[14], pf[4] - pf[15]);
                }
                if (mi[0]) {
                    const int n = mi[1], nb = mi[3];
                    if (n > q->cap || nb > q->cap_blocks) { too_small[t] = 1; return; }
                    // (x0 is documented as 9 * cap_blocks doubles: global sizes are 7 / 9 / 7 / 1, so nb <= cap_blocks bounds it)
                    q->n = n; q->m = mi[2]; q->nblocks = nb;
                    const int mcap = L.mcap;
                    int x0n = 0;
                    for (int b = 0; b < nb; ++b) {
                        q->block_kind[b] = mi[8 + b];
                        q->block_index[b] = mi[8 + (L.K + 4) + b];
                        x0n += blk_gsize(q->block_kind[b]);
                    }
                    for (int r = 0; r < n; ++r) memcpy(q->J0 + (size_t)r * n, mo + L.mo_J0 + (size_t)r * mcap, sizeof(double) * n);
                    memcpy(q->r0, mo + L.mo_r0, sizeof(double) * n);
                    memcpy(q->x0, mo + L.mo_x0, sizeof(double) * x0n);
                }
            }
        }
    });
    for (int v : too_small) if (v) { h->err = "vg_ba_prior capacity too small"; return VG_ERR_BAD_ARG; }
    return VG_OK;
}

static int copy_states(vg_handle* h, int nwin, hipStream_t stream) {
    BaBatch& B = h->ba;
    const BaLayout& L = B.L;
    HIPCHK(h, B.h_out.resize((size_t)nwin * L.ostride));
    HIPCHK(h, B.h_iout.resize((size_t)nwin * L.oi_stride));
    HIPCHK(h, hipMemcpyAsync(B.h_out.data(), B.P.out, B.h_out.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(h, hipMemcpyAsync(B.h_iout.data(), B.P.iout, B.h_iout.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
    return VG_OK;
}
static int copy_priors(vg_handle* h, int nwin, hipStream_t stream) {
    BaBatch& B = h->ba;
    const BaLayout& L = B.L;
    HIPCHK(h, B.h_mout.resize((size_t)nwin * L.mo_stride));
    HIPCHK(h, B.h_miout.resize((size_t)nwin * L.mi_stride));
    HIPCHK(h, hipMemcpyAsync(B.h_mout.data(), B.P.mout, B.h_mout.size() * sizeof(double), hipMemcpyDeviceToHost, stream));
    HIPCHK(h, hipMemcpyAsync(B.h_miout.data(), B.P.miout, B.h_miout.size() * sizeof(int), hipMemcpyDeviceToHost, stream));
    return VG_OK;
}

extern "C" int vg_ba_batch_download_state(vg_handle* h, int nwin, vg_ba_state* const* st, vg_ba_summary* sum) {
    VG_RANGE("vg_ba_batch_download_state");
    if (!h || !h->ba.uploaded || nwin != h->ba.nwin) return VG_ERR_BAD_ARG;
    if (!h->ba.solved_recorded) { h->err = "vg_ba_batch_download_state before vg_ba_batch_run_async"; return VG_ERR_BAD_ARG; }
    HIPCHK(h, hipStreamWaitEvent(h->aux, h->ev_fork, 0));        // ev_fork: recorded behind ba_final_kernel by run_async
    int rc = copy_states(h, nwin, h->aux);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->aux));
    return unpack_states(h, nwin, st, sum);
}

extern "C" int vg_ba_batch_download_prior(vg_handle* h, int nwin, vg_ba_prior* const* pri) {
    VG_RANGE("vg_ba_batch_download_prior");
    if (!h || !h->ba.uploaded || nwin != h->ba.nwin || !pri) return VG_ERR_BAD_ARG;
    BaBatch& B = h->ba;
    if (B.any_margin) {
        const int rc = copy_priors(h, nwin, h->stream);
        if (rc) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return unpack_priors(h, nwin, pri);
}

extern "C" int vg_ba_batch_download(vg_handle* h, int nwin, vg_ba_state* const* st, vg_ba_summary* sum,
                                    vg_ba_prior* const* pri) {
    VG_RANGE("vg_ba_batch_download");
    if (!h || !h->ba.uploaded || nwin != h->ba.nwin) return VG_ERR_BAD_ARG;
    BaBatch& B = h->ba;
    int rc = copy_states(h, nwin, h->stream);
    if (rc) return rc;
    if (B.any_margin && pri) { rc = copy_priors(h, nwin, h->stream); if (rc) return rc; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int worst = unpack_states(h, nwin, st, sum);
    if (pri) { rc = unpack_priors(h, nwin, pri); if (rc) return rc; }
    return worst;
}

extern "C" int vg_ba_optimize(vg_handle* h, const vg_ba_problem* in, int margin_flag, vg_ba_state* out_state,
                              vg_ba_summary* out_summary, vg_ba_prior* out_prior) {
    VG_RANGE("vg_ba_optimize");
    if (!h || !in) return VG_ERR_BAD_ARG;
    const vg_ba_problem* arr[1] = {in};
    int rc = vg_ba_batch_upload(h, 1, arr, &margin_flag);
    if (rc) return rc;
    rc = vg_ba_batch_run_async(h);
    if (rc) return rc;
    vg_ba_state* sarr[1] = {out_state};
    vg_ba_prior* parr[1] = {out_prior};
    return vg_ba_batch_download(h, 1, sarr, out_summary, parr);
}

// The same in two calls: the first returns as soon as the states are on the host (the marginalization kernel keeps
// running on the device), the second collects the prior -- any time before the handle's next upload.
extern "C" int vg_ba_optimize_begin(vg_handle* h, const vg_ba_problem* in, int margin_flag, vg_ba_state* out_state,
                                    vg_ba_summary* out_summary) {
    VG_RANGE("vg_ba_optimize_begin");
    if (!h || !in) return VG_ERR_BAD_ARG;
    const vg_ba_problem* arr[1] = {in};
    int rc = vg_ba_batch_upload(h, 1, arr, &margin_flag);
    if (rc) return rc;
    rc = vg_ba_batch_run_async(h);
    if (rc) return rc;
    vg_ba_state* sarr[1] = {out_state};
    return vg_ba_batch_download_state(h, 1, sarr, out_summary);
}
extern "C" int vg_ba_optimize_prior(vg_handle* h, vg_ba_prior* out_prior) {
    VG_RANGE("vg_ba_optimize_prior");
    if (!h || !out_prior) return VG_ERR_BAD_ARG;
    vg_ba_prior* parr[1] = {out_prior};
    return vg_ba_batch_download_prior(h, 1, parr);
}

extern "C" int vg_ba_eval_factors(vg_handle* h, const vg_ba_problem* in, double* proj_r, double* proj_J,
                                  double* imu_r, double* imu_J, double* prior_r) {
    if (!h || !in) return VG_ERR_BAD_ARG;
    const vg_ba_problem* arr[1] = {in};
    int mf = VG_MARGIN_NONE;
    int rc = vg_ba_batch_upload(h, 1, arr, &mf);
    if (rc) return rc;
    BaBatch& B = h->ba;
    const BaLayout& L = B.L;
    const int F = B.h_ia[L.io_hdr + H_F], nimu = L.K - 1, n = B.h_ia[L.io_hdr + H_NPRIOR];
    double* d = nullptr;
    const size_t tot = (size_t)F * 2 + (size_t)F * 40 + nimu * 15 + nimu * 450 + std::max(n, 1);
    HIPCHK(h, hipMalloc((void**)&d, tot * sizeof(double)));
    double* d_pr = d; double* d_pJ = d_pr + (size_t)F * 2; double* d_ir = d_pJ + (size_t)F * 40;
    double* d_iJ = d_ir + nimu * 15; double* d_qr = d_iJ + nimu * 450;
    hipError_t e = ba_launch_eval_factors(L, B.dL, B.P, d_pr, d_pJ, d_ir, d_iJ, d_qr, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess && proj_r) e = hipMemcpy(proj_r, d_pr, (size_t)F * 2 * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && proj_J) e = hipMemcpy(proj_J, d_pJ, (size_t)F * 40 * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && imu_r) e = hipMemcpy(imu_r, d_ir, (size_t)nimu * 15 * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && imu_J) e = hipMemcpy(imu_J, d_iJ, (size_t)nimu * 450 * 8, hipMemcpyDeviceToHost);
    if (e == hipSuccess && prior_r && n) e = hipMemcpy(prior_r, d_qr, (size_t)n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) { h->err = std::string("eval_factors: ") + hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}

// ---- capacities every layout is built for at least (build_layout) ---------------------------------------------------------
extern "C" int vg_ba_reserve(vg_handle* h, int max_landmarks, int max_factors, int max_obs, int max_prior_n) {
    if (!h || max_landmarks < 0 || max_factors < 0 || max_obs < 0 || max_prior_n < 0) return VG_ERR_BAD_ARG;
    h->ba.res_L = max_landmarks; h->ba.res_F = max_factors; h->ba.res_O = max_obs; h->ba.res_N = max_prior_n;
    h->ba.uploaded = false;
    return VG_OK;
}

// ---- windows that stay on the device from frame to frame (kernels: csrc/ba_seq.hip) ---------------------------------------
extern "C" hipError_t ba_seq_launch_front(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, const SeqDev& S, int cur, hipStream_t stream);
extern "C" hipError_t ba_seq_launch_slide(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, const SeqDev& S, int cur, hipStream_t stream);
extern "C" int ba_seq_limits(int* ft_max, int* nin_max, int* hdr_ints, int* in_rows_off);

static void seq_free(BaSeq& Q) {
    SeqDev& D = Q.D;
    for (int k = 0; k < 2; ++k) { (void)hipFree(D.ft_i[k]); (void)hipFree(D.ft_d[k]); D.ft_i[k] = nullptr; D.ft_d[k] = nullptr; }
    (void)hipFree(D.sp); (void)hipFree(D.in_i); (void)hipFree(D.in_d); (void)hipFree(D.info);
    D.sp = nullptr; D.in_i = nullptr; D.in_d = nullptr; D.info = nullptr;
    Q.h_in_i.release(); Q.h_in_d.release(); Q.h_info.release();
    Q.active = false;
}
extern "C" void ba_seq_release(vg_handle* h) { seq_free(h->ba.seq); }

extern "C" int vg_ba_seq_end(vg_handle* h) {
    if (!h) return VG_ERR_BAD_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    seq_free(h->ba.seq);
    return VG_OK;
}

// the track table of one window as the kernels of csrc/ba_seq.hip hold it, from the caller's arrays
static void seq_fill_tracks(const vg_ba_tracks* t, int FT, int K, int hdr_ints, int* i, double* d) {
    i[0] = t->n_features;
    size_t row = 0;
    for (int f = 0; f < t->n_features; ++f) {
        i[hdr_ints + f] = t->feature_id[f];
        i[hdr_ints + FT + f] = t->start_frame[f];
        i[hdr_ints + 2 * FT + f] = t->n_obs[f];
        i[hdr_ints + 3 * FT + f] = t->solve_flag ? t->solve_flag[f] : 0;
        i[hdr_ints + 4 * FT + f] = -1;
        d[f] = t->depth[f];
        for (int j = 0; j < t->n_obs[f]; ++j, ++row) {
            const double* r = t->obs + row * 8;                       // [x y z u v vx vy cur_td]
            double* o = d + FT + ((size_t)f * K + j) * 8;             // [x y u v vx vy cur_td z]
            o[0] = r[0]; o[1] = r[1]; o[2] = r[3]; o[3] = r[4]; o[4] = r[5]; o[5] = r[6]; o[6] = r[7]; o[7] = r[2];
        }
    }
}
static int seq_check_tracks(vg_handle* h, const vg_ba_tracks* t, int FT, int K) {
    if (t->n_features < 0 || t->n_features > FT || (t->n_features > 0 && (!t->feature_id || !t->start_frame || !t->n_obs || !t->depth || !t->obs))) {
        h->err = "bad track table"; return VG_ERR_BAD_ARG;
    }
    for (int f = 0; f < t->n_features; ++f)
        if (t->n_obs[f] < 1 || t->start_frame[f] < 0 || t->start_frame[f] + t->n_obs[f] > K - 1) {
            h->err = "a track reaches beyond frame K - 2 (the newest slot is filled by the next step)"; return VG_ERR_BAD_ARG;
        }
    return VG_OK;
}

extern "C" int vg_ba_seq_begin(vg_handle* h, int nwin, const vg_ba_seq_config* cfg, const vg_ba_problem* const* windows,
                               const vg_ba_tracks* const* tracks) {
    VG_RANGE("vg_ba_seq_begin");
    if (!h || nwin <= 0 || !cfg || !windows || !tracks) return VG_ERR_BAD_ARG;
    for (int w = 0; w < nwin; ++w) if (!windows[w] || !tracks[w]) return VG_ERR_BAD_ARG;
    BaBatch& B = h->ba;
    BaSeq& Q = B.seq;
    int ft_max = 0, nin_max = 0, hdr_ints = 0, rows_off = 0;
    const int l_max = ba_seq_limits(&ft_max, &nin_max, &hdr_ints, &rows_off);
    const int K = windows[0]->K;
    const int FT = cfg->max_features, NIN = cfg->max_new_obs;
    const int Lres = cfg->max_landmarks > 0 ? cfg->max_landmarks : FT;
    const int Fres = cfg->max_factors > 0 ? cfg->max_factors : 6 * Lres;
    if (FT < 1 || FT > ft_max || NIN < 1 || NIN > nin_max || Lres > l_max) { h->err = "vg_ba_seq_begin: capacities outside the kernels' tables"; return VG_ERR_UNSUPPORTED; }
    if (K < 4 || K > 12) { h->err = "vg_ba_seq_begin: a sequence needs 4 <= K <= 12 frames"; return VG_ERR_UNSUPPORTED; }
    if (B.allreduce || B.force_large) { h->err = "vg_ba_seq_begin: not offered on the large-window path"; return VG_ERR_UNSUPPORTED; }
    for (int w = 0; w < nwin; ++w) {
        const vg_ba_problem* p = windows[w];
        const vg_ba_tracks* t = tracks[w];
        if (p->relo_n != 0) { h->err = "vg_ba_seq_begin: relocalisation factors are not offered in a sequence"; return VG_ERR_UNSUPPORTED; }
        if (p->prior_n == VG_PRIOR_RESIDENT) { h->err = "vg_ba_seq_begin: the first prior comes from the host"; return VG_ERR_BAD_ARG; }
        if (p->max_iters != windows[0]->max_iters) { h->err = "vg_ba_seq_begin: windows must share max_iters"; return VG_ERR_BAD_ARG; }
        const int rct = seq_check_tracks(h, t, FT, K);
        if (rct) return rct;
    }
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    seq_free(Q);
    // the batch layout, the states, the pre-integrations and the first prior go the ordinary way (the windows' own landmark
    // tables are packed too and overwritten by the first step)
    const int keep_res[4] = {B.res_L, B.res_F, B.res_O, B.res_N};     // (the sequence's capacities hold for this layout only)
    B.res_L = std::max(B.res_L, Lres); B.res_F = std::max(B.res_F, Fres); B.res_O = std::max(B.res_O, Fres + Lres);
    B.res_N = std::max(B.res_N, 6 * K + 9 * 2 + 6 + 1);
    std::vector<int> flags(nwin, VG_MARGIN_OLD);
    int rc = vg_ba_batch_upload(h, nwin, windows, flags.data());
    B.res_L = keep_res[0]; B.res_F = keep_res[1]; B.res_O = keep_res[2]; B.res_N = keep_res[3];
    if (rc) return rc;
    if (B.L.big) { h->err = "vg_ba_seq_begin: window too wide for the single-workgroup pipeline"; B.uploaded = false; return VG_ERR_UNSUPPORTED; }
    std::fill(B.nL.begin(), B.nL.end(), std::min(B.L.Lcap, up(Lres, 16)));   // (state downloads: vg_ba_state::inv_depth, if given, takes
                                                                              //  max_landmarks rounded up to 16 values, as the header says)
    SeqDev& D = Q.D;
    D.K = K; D.FT = FT; D.NIN = NIN;
    D.fi_stride = up(hdr_ints + 5 * FT, 8);
    D.fd_stride = up(FT + FT * K * 8, 8);
    D.ii_stride = up(8 + NIN, 8);
    D.id_stride = up(rows_off + NIN * 8, 8);
    D.sp_stride = up(2 + 2 * (K + 4), 8);
    D.max_iters = windows[0]->max_iters; D.marg_mode = B.marg_mode;
    D.init_depth = cfg->init_depth; D.min_parallax = cfg->min_parallax;
    for (int k = 0; k < 2; ++k) {
        HIPCHK(h, hipMalloc((void**)&D.ft_i[k], (size_t)nwin * D.fi_stride * sizeof(int)));
        HIPCHK(h, hipMalloc((void**)&D.ft_d[k], (size_t)nwin * D.fd_stride * sizeof(double)));
        HIPCHK(h, hipMemsetAsync(D.ft_i[k], 0, (size_t)nwin * D.fi_stride * sizeof(int), h->stream));
        HIPCHK(h, hipMemsetAsync(D.ft_d[k], 0, (size_t)nwin * D.fd_stride * sizeof(double), h->stream));
    }
    HIPCHK(h, hipMalloc((void**)&D.sp, (size_t)nwin * D.sp_stride * sizeof(int)));
    HIPCHK(h, hipMalloc((void**)&D.in_i, (size_t)nwin * D.ii_stride * sizeof(int)));
    HIPCHK(h, hipMalloc((void**)&D.in_d, (size_t)nwin * D.id_stride * sizeof(double)));
    HIPCHK(h, hipMalloc((void**)&D.info, (size_t)nwin * VG_SEQ_INFO_INTS * sizeof(int)));
    HIPCHK(h, hipMemsetAsync(D.info, 0, (size_t)nwin * VG_SEQ_INFO_INTS * sizeof(int), h->stream));
    // track tables and prior block tables
    std::vector<int> ti((size_t)nwin * D.fi_stride, 0), sp((size_t)nwin * D.sp_stride, 0);
    std::vector<double> td((size_t)nwin * D.fd_stride, 0.0);
    for (int w = 0; w < nwin; ++w) {
        seq_fill_tracks(tracks[w], FT, K, hdr_ints, ti.data() + (size_t)w * D.fi_stride, td.data() + (size_t)w * D.fd_stride);
        const BaBatch::PriorSlot& s = B.slot[w];
        int* q = sp.data() + (size_t)w * D.sp_stride;
        q[0] = s.n; q[1] = s.n ? s.nb : 0;
        for (int b = 0; b < q[1]; ++b) { q[2 + b] = s.kind[b]; q[2 + (K + 4) + b] = s.idx[b]; }
    }
    HIPCHK(h, hipMemcpy(D.ft_i[0], ti.data(), ti.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(D.ft_d[0], td.data(), td.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(D.sp, sp.data(), sp.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    Q.nwin = nwin; Q.cur = 0; Q.active = true;
    B.solved_recorded = false;
    return VG_OK;
}

extern "C" int vg_ba_seq_step_async(vg_handle* h, int nwin, const vg_ba_frame* const* frames) {
    VG_RANGE("vg_ba_seq_step_async");
    if (!h || !frames) return VG_ERR_BAD_ARG;
    BaBatch& B = h->ba;
    BaSeq& Q = B.seq;
    if (!Q.active || !B.uploaded || nwin != Q.nwin || nwin != B.nwin) { h->err = "vg_ba_seq_step_async: no sequence with that many windows"; return VG_ERR_BAD_ARG; }
    const SeqDev& D = Q.D;
    int rows_off = 0;
    (void)ba_seq_limits(nullptr, nullptr, nullptr, &rows_off);
    for (int w = 0; w < nwin; ++w) {
        const vg_ba_frame* f = frames[w];
        if (!f || !f->imu_new || f->n_obs < 0 || (f->n_obs > 0 && (!f->feature_id || !f->obs))) return VG_ERR_BAD_ARG;
        if (f->n_obs > D.NIN) { h->err = "vg_ba_seq_step_async: more observations than vg_ba_seq_config::max_new_obs"; return VG_ERR_UNSUPPORTED; }
        // the `image` map of the reference is keyed by feature id: one observation per id, ascending (the device matches every
        // observation to its track independently: a repeated id would be two lanes appending to one track)
        for (int k = 1; k < f->n_obs; ++k)
            if (f->feature_id[k] <= f->feature_id[k - 1]) { h->err = "vg_ba_seq_step_async: feature ids of a frame must be strictly ascending"; return VG_ERR_BAD_ARG; }
    }
    HIPCHK(h, hipSetDevice(h->device));
    // the staging buffers are re-used: the previous step's copies must have left them
    HIPCHK(h, hipEventSynchronize(h->ev_join));
    HIPCHK(h, Q.h_in_i.resize((size_t)nwin * D.ii_stride));
    HIPCHK(h, Q.h_in_d.resize((size_t)nwin * D.id_stride));
    auto put_imu = [](double* d, const vg_imu_preint& m) {
        d[0] = m.sum_dt;
        memcpy(d + 1, m.delta_p, 24); memcpy(d + 4, m.delta_q, 32); memcpy(d + 8, m.delta_v, 24);
        memcpy(d + 11, m.linearized_ba, 24); memcpy(d + 14, m.linearized_bg, 24);
        memcpy(d + 17, m.jacobian, 225 * 8); memcpy(d + 242, m.covariance, 225 * 8);
    };
    tl_pack_cap = h->ba.pack_threads;
    for_windows(nwin, [&](int, int w) {
        const vg_ba_frame* f = frames[w];
        int* i = Q.h_in_i.data() + (size_t)w * D.ii_stride;
        double* d = Q.h_in_d.data() + (size_t)w * D.id_stride;
        i[0] = f->n_obs; i[1] = f->imu_merged ? 1 : 0; i[2] = f->imu_new->valid; i[3] = f->imu_merged ? f->imu_merged->valid : 0;
        i[4] = i[5] = i[6] = i[7] = 0;
        memcpy(i + 8, f->feature_id, sizeof(int) * f->n_obs);
        memcpy(d, f->pose, 7 * 8); memcpy(d + 7, f->speedbias, 9 * 8);
        put_imu(d + 16, *f->imu_new);
        if (f->imu_merged) put_imu(d + 16 + BA_IMU_STRIDE, *f->imu_merged);
        for (int k = 0; k < f->n_obs; ++k) { memcpy(d + rows_off + (size_t)k * 8, f->obs + (size_t)k * 7, 7 * 8); d[rows_off + (size_t)k * 8 + 7] = 0.0; }
    });
    HIPCHK(h, hipMemcpyAsync(D.in_i, Q.h_in_i.data(), Q.h_in_i.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(D.in_d, Q.h_in_d.data(), Q.h_in_d.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipEventRecord(h->ev_join, h->stream));
    hipError_t e = ba_seq_launch_front(B.L, B.dL, B.P, D, Q.cur, h->stream);
    if (e != hipSuccess) { h->err = std::string("launch of the sequence front kernels: ") + hipGetErrorString(e); return VG_ERR_HIP; }
    int rc = launch_solve_in_mode(h);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->ev_fork, h->stream));           // states final (vg_ba_batch_download_state)
    B.solved_recorded = true;
    B.any_margin = true;
    HIPCHK(h, ba_launch_marg(B.L, B.dL, B.P, h->stream));
    e = ba_launch_carry_prior(B.nwin, B.P.mout, B.P.miout, B.L.mo_J0, B.L.mo_r0, B.L.mo_x0, B.L.mo_stride, B.L.mi_stride, B.L.mcap,
                              9 * (B.L.K + 4), B.P.pri, B.L.po_x0, B.L.po_r0, B.L.po_J0, B.L.pld, B.L.pstride, h->stream);
    if (e == hipSuccess) e = ba_seq_launch_slide(B.L, B.dL, B.P, D, Q.cur, h->stream);
    if (e != hipSuccess) { h->err = std::string("launch of the sequence slide kernels: ") + hipGetErrorString(e); return VG_ERR_HIP; }
    Q.cur ^= 1;
    B.mout_pending = false;                                      // (carried already; the slots' host mirror is not maintained in a sequence)
    return VG_OK;
}

extern "C" int vg_ba_seq_info(vg_handle* h, int nwin, int* info) {
    if (!h || !info) return VG_ERR_BAD_ARG;
    BaSeq& Q = h->ba.seq;
    if (!Q.active || nwin != Q.nwin) return VG_ERR_BAD_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(info, Q.D.info, (size_t)nwin * VG_SEQ_INFO_INTS * sizeof(int), hipMemcpyDeviceToHost));
    return VG_OK;
}

extern "C" int vg_ba_seq_get_tracks(vg_handle* h, int window, int cap, int* n_features, int* feature_id, int* start_frame, int* n_obs,
                                    int* solve_flag, double* depth, double* obs) {
    if (!h || !n_features || cap < 0) return VG_ERR_BAD_ARG;
    BaSeq& Q = h->ba.seq;
    if (!Q.active || window < 0 || window >= Q.nwin) return VG_ERR_BAD_ARG;
    const SeqDev& D = Q.D;
    int hdr_ints = 0;
    (void)ba_seq_limits(nullptr, nullptr, &hdr_ints, nullptr);
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<int> ti(D.fi_stride);
    HIPCHK(h, hipMemcpy(ti.data(), D.ft_i[Q.cur] + (size_t)window * D.fi_stride, ti.size() * sizeof(int), hipMemcpyDeviceToHost));
    const int n = ti[0];
    *n_features = n;
    if (n > cap) { h->err = "vg_ba_seq_get_tracks: capacity too small"; return VG_ERR_BAD_ARG; }
    for (int f = 0; f < n; ++f) {
        if (feature_id) feature_id[f] = ti[hdr_ints + f];
        if (start_frame) start_frame[f] = ti[hdr_ints + D.FT + f];
        if (n_obs) n_obs[f] = ti[hdr_ints + 2 * D.FT + f];
        if (solve_flag) solve_flag[f] = ti[hdr_ints + 3 * D.FT + f];
    }
    const double* dsrc = D.ft_d[Q.cur] + (size_t)window * D.fd_stride;
    if (depth && n) HIPCHK(h, hipMemcpy(depth, dsrc, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    if (obs && n) HIPCHK(h, hipMemcpy(obs, dsrc + D.FT, (size_t)n * D.K * 8 * sizeof(double), hipMemcpyDeviceToHost));
    return VG_OK;
}

// ---- hand-back / re-seed of one window of a running sequence ----------------------------------------------------------------
extern "C" int vg_ba_seq_export(vg_handle* h, int window, double* pose, double* speedbias, double* ex_pose, double* td,
                                vg_imu_preint* imu, vg_ba_prior* prior) {
    VG_RANGE("vg_ba_seq_export");
    if (!h) return VG_ERR_BAD_ARG;
    BaBatch& B = h->ba;
    BaSeq& Q = B.seq;
    if (!Q.active || window < 0 || window >= Q.nwin) return VG_ERR_BAD_ARG;
    const BaLayout& L = B.L;
    const int K = L.K;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    std::vector<double> di(L.dstride);
    std::vector<int> ia(L.istride), sp(Q.D.sp_stride);
    HIPCHK(h, hipMemcpy(di.data(), B.P.din + (size_t)window * L.dstride, di.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(ia.data(), B.P.iarr + (size_t)window * L.istride, ia.size() * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(sp.data(), Q.D.sp + (size_t)window * Q.D.sp_stride, sp.size() * sizeof(int), hipMemcpyDeviceToHost));
    if (pose) memcpy(pose, di.data() + L.do_pose, sizeof(double) * 7 * K);
    if (speedbias) memcpy(speedbias, di.data() + L.do_sb, sizeof(double) * 9 * K);
    if (ex_pose) memcpy(ex_pose, di.data() + L.do_ex, sizeof(double) * 7);
    if (td) *td = di[L.do_td];
    if (imu) {
        for (int k = 0; k < K - 1; ++k) {
            const double* d = di.data() + L.do_imu + (size_t)k * BA_IMU_STRIDE;
            vg_imu_preint& m = imu[k];
            memset(&m, 0, sizeof(m));
            m.sum_dt = d[0];
            memcpy(m.delta_p, d + 1, 24); memcpy(m.delta_q, d + 4, 32); memcpy(m.delta_v, d + 8, 24);
            memcpy(m.linearized_ba, d + 11, 24); memcpy(m.linearized_bg, d + 14, 24);
            memcpy(m.jacobian, d + 17, 225 * 8); memcpy(m.covariance, d + 242, 225 * 8);
            m.valid = (k < K - 2) ? ia[L.io_imu_valid + k] : 0;      // (the newest interval is a placeholder between two frames)
        }
    }
    if (prior) {
        prior->n = prior->m = prior->nblocks = prior->valid = 0;
        const int n = sp[0], nb = n ? sp[1] : 0;
        if (n > 0) {
            if (n > prior->cap || nb > prior->cap_blocks) { h->err = "vg_ba_prior capacity too small"; return VG_ERR_BAD_ARG; }
            std::vector<double> pr(L.pstride);
            HIPCHK(h, hipMemcpy(pr.data(), B.P.pri + (size_t)window * L.pstride, pr.size() * sizeof(double), hipMemcpyDeviceToHost));
            int x0n = 0;
            for (int b = 0; b < nb; ++b) {
                prior->block_kind[b] = sp[2 + b];
                prior->block_index[b] = sp[2 + (K + 4) + b];
                x0n += blk_gsize(sp[2 + b]);
            }
            for (int r = 0; r < n; ++r) memcpy(prior->J0 + (size_t)r * n, pr.data() + L.po_J0 + (size_t)r * L.pld, sizeof(double) * n);
            memcpy(prior->r0, pr.data() + L.po_r0, sizeof(double) * n);
            memcpy(prior->x0, pr.data() + L.po_x0, sizeof(double) * x0n);
            prior->n = n; prior->nblocks = nb; prior->valid = 1;
        }
    }
    return VG_OK;
}

extern "C" int vg_ba_seq_import(vg_handle* h, int window, const vg_ba_problem* p, const vg_ba_tracks* t) {
    VG_RANGE("vg_ba_seq_import");
    if (!h || !p || !t) return VG_ERR_BAD_ARG;
    BaBatch& B = h->ba;
    BaSeq& Q = B.seq;
    if (!Q.active || window < 0 || window >= Q.nwin) return VG_ERR_BAD_ARG;
    const BaLayout& L = B.L;
    const SeqDev& D = Q.D;
    const int K = L.K, FT = D.FT;
    int hdr_ints = 0;
    (void)ba_seq_limits(nullptr, nullptr, &hdr_ints, nullptr);
    if (p->K != K || (p->estimate_extrinsic != 0) != (L.e != 0) || (p->estimate_td != 0) != (L.t != 0) || p->relo_n != 0 || p->prior_n == VG_PRIOR_RESIDENT ||
        p->max_iters != D.max_iters) { h->err = "vg_ba_seq_import: the window does not fit the running sequence (K, options)"; return VG_ERR_BAD_ARG; }
    int rc = check_problem(h, p);
    if (rc) return rc;
    if (p->prior_n > L.Ncap || p->prior_nblocks > L.NBcap || p->prior_nblocks > K + 4) { h->err = "vg_ba_seq_import: prior beyond the sequence's capacities"; return VG_ERR_UNSUPPORTED; }
    rc = seq_check_tracks(h, t, FT, K);
    if (rc) return rc;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // states, pre-integrations, parameters: the slot's input slab, packed like an upload (the landmark tables are rebuilt by the
    // next step anyway); the prior into the slot of the prior buffer
    std::vector<int> ia(L.istride, 0);
    std::vector<double> di(L.dstride, 0.0), pr(L.pstride, 0.0);
    PriorRef ref;
    ref.n = p->prior_n; ref.nb = p->prior_n ? p->prior_nblocks : 0; ref.kind = p->prior_block_kind; ref.idx = p->prior_block_index;
    vg_ba_problem q = *p;
    q.L = 0; q.n_obs = 0;
    rc = pack_window(h, L, &q, ref, VG_MARGIN_OLD, ia.data(), di.data(), pr.data());
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(B.P.iarr + (size_t)window * L.istride, ia.data(), ia.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(B.P.din + (size_t)window * L.dstride, di.data(), di.size() * sizeof(double), hipMemcpyHostToDevice));
    if (ref.n > 0) HIPCHK(h, hipMemcpy(B.P.pri + (size_t)window * L.pstride, pr.data(), pr.size() * sizeof(double), hipMemcpyHostToDevice));
    std::vector<int> sp(D.sp_stride, 0);
    sp[0] = ref.n; sp[1] = ref.nb;
    for (int b = 0; b < ref.nb; ++b) { sp[2 + b] = ref.kind[b]; sp[2 + (K + 4) + b] = ref.idx[b]; }
    HIPCHK(h, hipMemcpy(D.sp + (size_t)window * D.sp_stride, sp.data(), sp.size() * sizeof(int), hipMemcpyHostToDevice));
    // the track table the next step reads
    std::vector<int> ti(D.fi_stride, 0);
    std::vector<double> td(D.fd_stride, 0.0);
    seq_fill_tracks(t, FT, K, hdr_ints, ti.data(), td.data());
    HIPCHK(h, hipMemcpy(D.ft_i[Q.cur] + (size_t)window * D.fi_stride, ti.data(), ti.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(D.ft_d[Q.cur] + (size_t)window * D.fd_stride, td.data(), td.size() * sizeof(double), hipMemcpyHostToDevice));
    return VG_OK;
}
