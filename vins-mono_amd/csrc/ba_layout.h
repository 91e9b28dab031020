// ba_layout.h — device memory layout of a batch of sliding windows (shared by the host packer
// ba_host.hip and the kernels in ba_kernels.hip).  One workgroup owns one window; every per-window
// array lives at  base + window * stride  so a workgroup streams its own contiguous slab of HBM.
//
// Reduced-system column order (tangent space):
//   [ pose_0 .. pose_{Kp-1} (6 each) | ex (6, if estimated) | td (1, if estimated) | speedbias_0 .. (9 each) ]
//   `Rc` = width of the "camera part" touched by projection factors, R = Rc + 9K.
#pragma once

#define BA_NT 512                 // threads per workgroup (8 wavefronts)
#define BA_NW (BA_NT / 64)
#define BA_MAX_K 13               // frames incl. relocalisation pose
#define BA_IMU_STRIDE 472         // doubles per vg_imu_preint record on device
#define BA_OBS_STRIDE 8
#define BA_SUM_DOUBLES 8
#define BA_HDR_INTS 16

enum { H_L = 0, H_F, H_NPRIOR, H_NBLK, H_MAXIT, H_NCHUNK, H_MARGIN, H_STATUS };

struct BaLayout {
    int nwin, K, Kp, e, t;
    int Rc, RcPad, R;
    int Lcap, Fcap, Ocap, Ncap, NBcap, Ccap;
    int REC, chunk_cap;
    // ---- int arrays (offsets in ints, per window)
    int io_hdr, io_lm_start, io_lm_fbeg, io_fac_i, io_fac_j, io_fac_lm, io_fac_oi, io_fac_oj, io_fac_slot,
        io_chunk_fbeg, io_chunk_lbeg, io_pair_ptr, io_imu_valid, io_pb_kind, io_pb_idx, io_pb_col, io_pb_off,
        io_pb_x0off, istride;
    // ---- double inputs (offsets in doubles, per window)
    int do_pose, do_sb, do_ex, do_td, do_lam, do_obs, do_imu, do_pJ0, do_pJ0t, do_pr0, do_px0, do_par, dstride;
    // ---- scratch (doubles, per window)
    int so_imuU, so_imuJ, so_imuR, so_Hp, so_pr, so_prc, so_pu, so_Wt, so_h, so_b, so_sl, so_dgl, so_gtl, so_gnl, so_ul,
        so_lam, so_lamc, so_yl, sstride;
    // ---- outputs (doubles / ints per window)
    int oo_pose, oo_sb, oo_ex, oo_td, oo_lam, oo_sum, oo_trace, ostride;
    int oi_stride;                 // int outputs: [status, termination, num_iterations, num_accepted, flags[VG_MAX_ITERS]]
    // ---- marginalization outputs (doubles / ints per window)
    int mo_J0, mo_r0, mo_x0, mo_stride;       // doubles
    int mi_stride;                            // ints: [valid, n, m, nblocks, kind[NBcap], idx[NBcap]]
    int ms_stride;                            // marginalization scratch doubles per window
    int mcap, mg_ld, mg_posmax, mg_cs, mg_lds_bytes; // kept-dimension capacity, LDS eig leading dim, max (m+n)
    // ---- LDS carve (offsets in doubles)
    int l_S, l_stage, l_vec, l_red, l_wd, l_x, l_xc, l_misc, l_pmap, lds_bytes;
    int nvec;                                 // number of R-vectors at l_vec, each Rpad long
    int Rpad;
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
};

// parameter slots at do_par
enum { P_FOCAL = 0, P_TR, P_ROW, P_GNORM, P_NPAR = 8 };
