// ba_layout.h — device memory layout of a batch of sliding windows (shared by the host packer ba_host.hip and the
// kernels of ba_pipeline.hip / ba_marg.hip).  Every per-window array lives at  base + window * stride.
//
// Tangent-space column order of the reduced system:
//   camera part  [ pose_0 .. pose_{Kp-1} (6 each) | ex (6, if estimated) | td (1, if estimated) ]   Rc columns
//   speed-bias   [ sb_0 .. sb_{K-1} (9 each) ]                                                      9K columns
// The landmark columns (1 each) are eliminated first (Schur complement on MFMA), the speed-bias blocks next (they form
// a block-tridiagonal chain: IMU factor k couples sb_k, sb_k+1, pose_k, pose_k+1; the prior adds sb_0), and only the
// Rc x Rc camera part is factorised densely.
#pragma once

#ifndef BA_NT
#define BA_NT 512                 // threads per workgroup of the per-window kernels (8 wavefronts)
#endif
#define BA_NW (BA_NT / 64)
// The per-round solve kernel of the single-workgroup path (ba_solve_kernel) runs 4 wavefronts per window and keeps its LDS carve
// under half a CU's 160 KB for the reference shape (K = 11: 72 KB), so that TWO windows are resident per CU (round 5): every phase
// of the solve is a latency-bound chain carried by one or two wavefronts, and the only lever on such a kernel is more independent
// work per CU.  8 waves x 256 VGPRs and 134 KB of LDS had made that impossible.
// A batch of a few windows (the drop-in's single window) has the CU to itself: it takes the same kernel source built for 8
// wavefronts (ba_solve_w8.hip defines SV_NT = 512 before this header and compiles ba_pipeline.hip a second time).
#ifndef SV_NT
#define SV_NT 256
#endif
#define SV_NW (SV_NT / 64)
#define SV_WG_PER_CU (SV_NT <= 256 ? 2 : 1)   // workgroups the register budget of ba_solve_kernel is cut for
#define BA_LIN_NT 256             // projection factors per workgroup of the linearisation kernel
#define BA_ACC_NT 256             // threads per workgroup of the accumulation kernel
#define BA_MAX_K 13               // frames incl. relocalisation pose, windows solved by one workgroup out of LDS
#define BA_MAX_K_LARGE 40         // the same for the large-window path (reduced system in HBM, see BaLayout::big)
#define BA_MARG_LDS_INTS 400      // int tables of ba_marg_kernel in LDS (MGI_* in ba_marg.hip)
#ifndef BA_IMU_BATCH
#define BA_IMU_BATCH 16           // IMU factors linearised per pass of the linearisation workgroup (LDS panels)
#endif
#ifndef BA_IMU_GROUP
#define BA_IMU_GROUP 2            // IMU factors per workgroup of the linearisation kernel
#endif
#define BA_IMU_STRIDE 472         // doubles per vg_imu_preint record on device
#define BA_OBS_STRIDE 8
#define BA_SUM_DOUBLES 16     // init cost, cost, radius, gauge rot_diff (9), post-solve position of frame 0 (3)
#define BA_HDR_INTS 16

enum { H_L = 0, H_F, H_NPRIOR, H_NBLK, H_MAXIT, H_NCHUNK, H_MARGIN, H_STATUS, H_MARGMODE };

// per-window solver state that lives in HBM between the launches of one solve (doubles; integers stored exactly)
enum {
    C_IT = 0, C_NACC, C_NINV, C_TERM, C_STATUS, C_RADIUS, C_MU, C_MUSOLVED, C_REUSE, C_COST, C_XNORM, C_ALPHA, C_GTN2,
    C_GNN2, C_GTGN, C_DNORM, C_MODEL, C_CUR, C_PENDING, C_DONE, C_STEPNORM, C_XNORMC, C_INITCOST, C_SCALED, C_QCAM,
    // large-window path only: phase of the round (0 nothing to do, 1 Gauss-Newton step solved, 2 step reuse), the camera /
    // speed-bias shares of |gn|^2, gt.gn, |x - x_cand|^2, |x_cand|^2 (the landmark shares travel through the reduce buffers)
    C_PHASE, C_GNN2C, C_GTGNC, C_STEP2C, C_XN2C,
    C_T0,                          // device wall clock at the start of the solve (vg_ba_problem::max_solver_time_s)
    C_TIMEDOUT,                    // large-window path: the time test of the round's first half, for its second half
    C_NCTL = 32
};

// reduce buffer 1 of the large-window path, per window:  [Sp  Rc(Rc+1)/2 | gp  Rc | T  (Rc+1)(Rc+2)/2 | scalars RB1_NSCAL]
//   Sp, gp = J^T J / J^T r of this rank's projection factors (camera part);  T = sum_l omega_l [W_l; b_l][W_l; b_l]^T of this
//   rank's landmarks (packed lower, augmented row Rc = rhs);  scalars below.  Summed over the ranks by the caller's
//   all-reduce (RCCL over xGMI) between the Schur kernel and the solve kernel; a single rank skips the collective.
enum { RB1_COST = 0, RB1_GTL2, RB1_LAM2, RB1_STEP2, RB1_NBIG, RB1_NSCAL = 8 };
// reduce buffer 2: landmark shares of |gn|^2, gt.gn, the Cauchy-point term, the count of non-finite step entries -- one
// group of four per workgroup of ba_big_landmark_kernel (a fixed number, so that the buffer has the same length on every
// rank), summed in workgroup order by the step kernel
enum { RB2_GNN2 = 0, RB2_GTGN, RB2_QL, RB2_NONFIN };
#define BA_BIG_LM_BLOCKS 32
#define BA_BIG_ZERO_BLOCKS 8      // workgroups of the Schur kernel that clear the chain blocks XC / D / E for the next assembly
#define RB2_LEN (4 * BA_BIG_LM_BLOCKS)

// R-vectors of the solve kernels, columns [camera Rc | speed-bias 9K]
enum { V_G = 0, V_SC, V_DG, V_GT, V_GN, V_U, V_Y, V_T, V_DI, V_NVEC };
// One entry of the gather plan: where the terms of ONE stored entry of the unscaled reduced system come from.
//   dst    : LDS slot (doubles from the start of the carve) | kind << 28   (kind 0: Hessian entry, 1: gradient entry)
//   base   : offset into the linearisation buffer of the projection factors' term (Sp / gp), -1 = none
//   imu    : (even + 1) | (odd + 1) << 16 with even / odd = f * 512 + local index of the term of the even / odd IMU factor, 0 = none
//   cols   : Hessian: reduced columns ca << 16 | cb (prior term J0^T J0 [pinv ca][pinv cb]); gradient: column ca (J0^T r [pinv ca])
struct alignas(16) AsmPlanEntry { int dst, base, imu, cols; };

struct BaLayout {
    int nwin, K, Kp, e, t;
    int Rc, RcPad, R, Rpad;        // RcPad = up(Rc + 1, 16): the rhs travels as the augmented row / column Rc
    int Lcap, Fcap, Ocap, Ncap, NBcap;
    int REC;                       // doubles per projection-factor record
    int nbf, nbl, nba, ntask;      // workgroups per window: projection tiles, cost partials (nbf + nig + 1), accumulation; owner tasks
    int nig, igs, nprw;            // IMU linearisation: groups per window, factors per group (one workgroup each), workgroups for the prior (0: group 0 does it)
    int nst;                       // doubles of one state copy [pose Kp*7 | sb K*9 | ex 7 | td 1]
    int imu_info;                  // form of the IMU factors' sqrt_info (vg_ba_set_imu_info_mode): 0 = U^-1 of covariance = U U^T, 1 = inverse() then LLT as imu_factor.h:64 spells it
    // ---- int arrays (offsets in ints, per window)
    int io_hdr, io_lm_start, io_lm_fbeg, io_fac_i, io_fac_j, io_fac_lm, io_fac_oi, io_fac_oj, io_fac_slot,
        io_pair_ptr, io_task_list, io_imu_valid, io_pb_kind, io_pb_idx, io_pb_col, io_pb_off, io_pb_x0off, istride;
    // ---- double inputs (offsets in doubles, per window)
    int do_pose, do_sb, do_ex, do_td, do_lam, do_obs, do_imu, do_par, dstride;
    // ---- prior factor (doubles per window, buffer of its own: it can stay on the device from one frame's marginalization to
    //      the next frame's solve, VG_PRIOR_RESIDENT): x0 [NB x 9] | r0 [pld] | J0 [pld x pld], row stride pld
    int po_x0, po_r0, po_J0, pld, pstride;
    // ---- scratch (doubles, per window)
    int so_ctl, so_part, so_x, so_lam;          // control block, cost partials, 2 state copies (nst each), 2 x Lcap
    int so_imuU, so_Hp, so_rec;                 // sqrt_info factors, J0^T J0, projection records [Fcap][REC]
    int so_J0t;                                 // J0 transposed (row stride Ncap), written by the prologue
    int so_sc, so_sl, so_dg, so_gt, so_gn;      // Jacobi scaling (R / Lcap), saved Dg, gt, gn over [R | Lcap] for step reuse
    int so_yl, so_lsc;                          // landmark step / sl/sqrt(h~)
    int so_xp;                                  // single-workgroup path: the eliminated speed-bias rows X_k = L_k^-1 [C_k | g_k], [9K][ldc], parked
                                                // here by the chain elimination for the back substitution (they no longer stay in LDS)
    int so_buf, buf_stride;                     // two linearisation buffers; offsets below are relative to a buffer
    int bo_Sp, bo_gp, bo_h, bo_b, bo_Wt, bo_imuJ, bo_pr, bo_gpr;
    int sstride;
    // ---- outputs (doubles / ints per window)
    int oo_pose, oo_sb, oo_ex, oo_td, oo_lam, oo_sum, oo_trace, ostride;
    int oi_stride;                 // int outputs: [status, termination, num_iterations, num_accepted, flags[VG_MAX_ITERS]]
    // ---- marginalization outputs (doubles / ints per window)
    int mo_J0, mo_r0, mo_x0, mo_stride;       // doubles
    int mi_stride;                            // ints: [valid, n, m, nblocks, kind[NBcap], idx[NBcap]]
    int ms_stride;                            // marginalization scratch doubles per window
    int mcap, mg_ld, mg_posmax, mg_cs, mg_lds_bytes; // kept-dimension capacity, LDS eig leading dim, max (m+n)
    // ---- LDS carve of the solve kernel (offsets in doubles)
    int l_S, l_XC, l_D, l_E, l_dinv, l_vec, l_red, l_wd, l_z, l_pmap, ldc, lds_solve;
    // single-workgroup path: l_ring = two slots of 18 coupling rows [2][18][ldc] (the block pair being eliminated and the pair before
    // it; l_wd aliases it: the staged landmark tile is only needed after the chain), l_pinv = reduced column -> prior index (R ints)
    int l_ring, l_pinv;
    // gather plan of assemble_small (single-workgroup path): pl_n entries of four ints, pl_off bytes behind the start of the device
    // copy of this struct (built by the host with the layout: it depends on nothing else)
    int pl_off, pl_n;
    int lds_lin, lds_pro;                     // dynamic LDS bytes of the linearisation / prologue kernels
    // ---- fused projection kernel (ba_linacc_proj_kernel): eligible windows (la_on), landmarks per chunk by first factor index
    //      (la_chq), staged-record capacity (la_chf = la_chq + 16), LDS offsets (doubles) of the pair blocks and of the key table
    int la_on, la_chq, la_chf, la_P, la_key, la_x, lds_linacc;      // (la_x: the state of the linearisation point, staged once per launch)
    // ---- host only: the batch is solved by ba_solve_w8_kernel (8 wavefronts per window) instead of ba_solve_kernel (4): few windows
    int sv_w8;
    // ---- ba_solve_w8_kernel: the landmark tile of its side-by-side Schur phase (chain_schur_split) behind the carve: offset in doubles;
    //      lds_solve_w8 = dynamic LDS bytes of that launch (host only)
    int l_wd8, lds_solve_w8;
    // ---- host only: the prologue's independent pieces run as workgroups side by side (the latency layout: few windows on an empty chip)
    int pro_split;
    // ---- large-window path (big != 0): the camera part does not fit the LDS carve above.  S stays in LDS (packed, with
    //      the rhs row), everything else of the carve lives in HBM scratch at so_bigm (the l_* offsets are then relative to
    //      it); the landmark Schur complement is formed by a multi-workgroup kernel into reduce buffer 1.
    int big;
    int so_bigm;                              // HBM home of XC / D / E / dinv / vec / wd / z / pmap
    int l_di, l_cz;                           // LDS offsets of the 1/L_jj vector and of the chain elimination's scratch (big path)
    int l_Sg;                                 // big path: the reduced system between the two halves of ba_solve_big_kernel (HBM, offset into so_bigm)
    int so_dgl, so_gtl;                       // landmark Dg, gt per linearisation buffer (2 x Lcap each)
    int rb1_len, rb1_T, rb1_scal;             // reduce buffer 1: doubles per window, offsets of T and of the scalars
    int nts;                                  // lower 16x16 tiles of T = workgroups of the Schur kernel (+ 1 landmark workgroup)
};

struct BaPtrs {
    int* iarr;
    double* din;
    double* scr;
    double* out;
    int* iout;
    double* mout;
    int* miout;
    double* mscr;
    double* rb1;                              // large-window path: reduce buffers [nwin][rb1_len], [nwin][RB2_LEN]
    double* rb2;
    double* pri;                              // prior factors [nwin][pstride]
};

// parameter slots at do_par
enum { P_FOCAL = 0, P_TR, P_ROW, P_GNORM, P_MAXTIME, P_NPAR = 8 };
