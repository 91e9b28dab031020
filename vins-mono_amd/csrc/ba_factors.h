// ba_factors.h — per-factor residual / Jacobian device functions (FP64).
//   proj_eval     : ProjectionFactor::Evaluate    (factor/projection_factor.cpp:21-121)
//                   ProjectionTdFactor::Evaluate  (factor/projection_td_factor.cpp:34-141)
//   ImuCtx / imu_*: IMUFactor::Evaluate           (factor/imu_factor.h:19-179)
//                   IntegrationBase::evaluate     (factor/integration_base.h:160-186)
// Jacobians are produced directly in tangent space (the reference fills a 7-wide pose block whose
// last column is zero and Ceres drops it through PoseLocalParameterization::ComputeJacobian).
#pragma once
#include "ba_math.h"

// device record of vg_imu_preint (BA_IMU_STRIDE doubles)
enum { IM_SUMDT = 0, IM_DP = 1, IM_DQ = 4, IM_DV = 8, IM_BA = 11, IM_BG = 14, IM_JAC = 17, IM_COV = 242 };

// ------------------------------------------------------------------------------------------------
// Projection factor.  obs rows: [x y u v vx vy cur_td pad].  J* are [2][6] row-major, Jl / Jtd [2].
template <bool TD, bool JAC, bool EX>
DEV void proj_eval(const double* pose_i, const double* pose_j, const double* ex, double lam,
                   const double* oi, const double* oj, double td, double focal, double tr, double row,
                   double* r, double* Ji, double* Jj, double* Jex, double* Jl, double* Jtd) {
    const double s = focal / 1.5;
    double pts_i[3] = {oi[0], oi[1], 1.0};
    double ptj_x = oj[0], ptj_y = oj[1];
    double vel_i[3] = {0, 0, 0}, vel_jx = 0, vel_jy = 0;
    if (TD) {
        vel_i[0] = oi[4]; vel_i[1] = oi[5];
        vel_jx = oj[4]; vel_jy = oj[5];
        const double ai = td - oi[6] + tr / row * (oi[3] - row / 2);
        const double aj = td - oj[6] + tr / row * (oj[3] - row / 2);
        pts_i[0] -= ai * vel_i[0]; pts_i[1] -= ai * vel_i[1];
        ptj_x -= aj * vel_jx; ptj_y -= aj * vel_jy;
    }
    double Ri[9], Rj[9], Rc[9];
    q_to_R(pose_i + 3, Ri);
    q_to_R(pose_j + 3, Rj);
    q_to_R(ex + 3, Rc);
    const double ilam = 1.0 / lam;
    const double pci[3] = {pts_i[0] * ilam, pts_i[1] * ilam, pts_i[2] * ilam};
    double pbi[3], pw[3], pbj[3], pcj[3], t3[3];
    m3_vec(Rc, pci, pbi);
    pbi[0] += ex[0]; pbi[1] += ex[1]; pbi[2] += ex[2];
    m3_vec(Ri, pbi, pw);
    pw[0] += pose_i[0]; pw[1] += pose_i[1]; pw[2] += pose_i[2];
    t3[0] = pw[0] - pose_j[0]; t3[1] = pw[1] - pose_j[1]; t3[2] = pw[2] - pose_j[2];
    m3t_vec(Rj, t3, pbj);
    t3[0] = pbj[0] - ex[0]; t3[1] = pbj[1] - ex[1]; t3[2] = pbj[2] - ex[2];
    m3t_vec(Rc, t3, pcj);
    const double idep = 1.0 / pcj[2];
    r[0] = s * (pcj[0] * idep - ptj_x);
    r[1] = s * (pcj[1] * idep - ptj_y);
    if (!JAC) return;
    // reduce = sqrt_info * [[1/z,0,-x/z^2],[0,1/z,-y/z^2]]
    const double r00 = s * idep, r02 = -s * pcj[0] * (idep * idep), r12 = -s * pcj[1] * (idep * idep);
    double A[9];                         // ric^T Rj^T
    {
        double RjRc[9];
        m3_mul(Rj, Rc, RjRc);            // (Rj ric)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) A[i * 3 + j] = RjRc[j * 3 + i];
    }
    double M[9], Sk[9], T[9];
    // pose i: [A , -A Ri [pbi]x]
    skew3(pbi, Sk);
    m3_mul(A, Ri, T);                    // A Ri
    m3_mul(T, Sk, M);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Ji[c] = r00 * A[c] + r02 * A[6 + c];
        Ji[6 + c] = r00 * A[3 + c] + r12 * A[6 + c];
        Ji[3 + c] = -(r00 * M[c] + r02 * M[6 + c]);
        Ji[9 + c] = -(r00 * M[3 + c] + r12 * M[6 + c]);
    }
    // pose j: [-A , ric^T [pbj]x]
    skew3(pbj, Sk);
    m3t_mul(Rc, Sk, M);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        Jj[c] = -Ji[c];
        Jj[6 + c] = -Ji[6 + c];
        Jj[3 + c] = r00 * M[c] + r02 * M[6 + c];
        Jj[9 + c] = r00 * M[3 + c] + r12 * M[6 + c];
    }
    double tmp_r[9];
    m3_mul(T, Rc, tmp_r);                // ric^T Rj^T Ri ric
    {
        double v[3];
        m3_vec(tmp_r, pts_i, v);
        const double k = -(ilam * ilam);
        Jl[0] = (r00 * v[0] + r02 * v[2]) * k;
        Jl[1] = (r00 * v[1] + r12 * v[2]) * k;
        if (TD) {
            m3_vec(tmp_r, vel_i, v);
            const double kk = -ilam;
            Jtd[0] = (r00 * v[0] + r02 * v[2]) * kk + s * vel_jx;
            Jtd[1] = (r00 * v[1] + r12 * v[2]) * kk + s * vel_jy;
        }
    }
    if (EX) {
        // left: ric^T (Rj^T Ri - I) = T - ric^T
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) M[i * 3 + j] = T[i * 3 + j] - Rc[j * 3 + i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Jex[c] = r00 * M[c] + r02 * M[6 + c];
            Jex[6 + c] = r00 * M[3 + c] + r12 * M[6 + c];
        }
        // right: -tmp_r [pci]x + [tmp_r pci]x + [ric^T (Rj^T (Ri tic + Pi - Pj) - tic)]x
        double v1[3], v2[3], v3[3];
        skew3(pci, Sk);
        m3_mul(tmp_r, Sk, M);
        m3_vec(tmp_r, pci, v1);
        m3_vec(Ri, ex, v2);
        v2[0] += pose_i[0] - pose_j[0]; v2[1] += pose_i[1] - pose_j[1]; v2[2] += pose_i[2] - pose_j[2];
        m3t_vec(Rj, v2, v3);
        v3[0] -= ex[0]; v3[1] -= ex[1]; v3[2] -= ex[2];
        m3t_vec(Rc, v3, v2);
        v1[0] += v2[0]; v1[1] += v2[1]; v1[2] += v2[2];
        skew3(v1, Sk);
#pragma unroll
        for (int k = 0; k < 9; ++k) M[k] = Sk[k] - M[k];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            Jex[3 + c] = r00 * M[c] + r02 * M[6 + c];
            Jex[9 + c] = r00 * M[3 + c] + r12 * M[6 + c];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// IMU factor context: everything needed for the raw (un-weighted) residual and any raw Jacobian column.
struct ImuCtx {
    double Rinv[9];      // R(Qi^-1)
    double vP[3], vV[3]; // Qi^-1 (..) terms
    double r[15];        // raw residual
    double M1[9];        // [Qleft(Qj^-1 Qi) Qright(corrected_delta_q)]_3x3
    double M2d[9];       // [Qleft(Qj^-1 Qi delta_q)]_3x3 * dq_dbg
    double M3[9];        // [Qleft(corrected_delta_q^-1 Qi^-1 Qj)]_3x3
    double dt;
};

// PP: pointer type of the pre-integration record (generic, or typed as global memory by the caller)
template <bool JAC, typename PP>
DEV void imu_ctx(PP pre, const double* pose_i, const double* sb_i, const double* pose_j,
                 const double* sb_j, double g_norm, ImuCtx& c) {
    const double dt = pre[IM_SUMDT];
    c.dt = dt;
    const PP Jm = pre + IM_JAC;
    double qi_inv[4], t3[3];
    q_inv(pose_i + 3, qi_inv);
    q_to_R(qi_inv, c.Rinv);
    t3[0] = pose_j[0] - pose_i[0] - sb_i[0] * dt;
    t3[1] = pose_j[1] - pose_i[1] - sb_i[1] * dt;
    t3[2] = 0.5 * g_norm * dt * dt + pose_j[2] - pose_i[2] - sb_i[2] * dt;
    m3_vec(c.Rinv, t3, c.vP);
    t3[0] = sb_j[0] - sb_i[0];
    t3[1] = sb_j[1] - sb_i[1];
    t3[2] = g_norm * dt + sb_j[2] - sb_i[2];
    m3_vec(c.Rinv, t3, c.vV);
    double dba[3], dbg[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        dba[k] = sb_i[3 + k] - pre[IM_BA + k];
        dbg[k] = sb_i[6 + k] - pre[IM_BG + k];
    }
    double th[3], cdp[3], cdv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        th[k] = Jm[(3 + k) * 15 + 12] * dbg[0] + Jm[(3 + k) * 15 + 13] * dbg[1] + Jm[(3 + k) * 15 + 14] * dbg[2];
        cdp[k] = pre[IM_DP + k] + Jm[k * 15 + 9] * dba[0] + Jm[k * 15 + 10] * dba[1] + Jm[k * 15 + 11] * dba[2]
                 + Jm[k * 15 + 12] * dbg[0] + Jm[k * 15 + 13] * dbg[1] + Jm[k * 15 + 14] * dbg[2];
        cdv[k] = pre[IM_DV + k] + Jm[(6 + k) * 15 + 9] * dba[0] + Jm[(6 + k) * 15 + 10] * dba[1] + Jm[(6 + k) * 15 + 11] * dba[2]
                 + Jm[(6 + k) * 15 + 12] * dbg[0] + Jm[(6 + k) * 15 + 13] * dbg[1] + Jm[(6 + k) * 15 + 14] * dbg[2];
    }
    const double dq[4] = {th[0] / 2.0, th[1] / 2.0, th[2] / 2.0, 1.0};
    const double pdq[4] = {pre[IM_DQ], pre[IM_DQ + 1], pre[IM_DQ + 2], pre[IM_DQ + 3]};
    double cdq[4], cdq_inv[4], qij[4], qe[4];
    q_mul(pdq, dq, cdq);
    q_inv(cdq, cdq_inv);
    q_mul(qi_inv, pose_j + 3, qij);
    q_mul(cdq_inv, qij, qe);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        c.r[k] = c.vP[k] - cdp[k];
        c.r[3 + k] = 2.0 * qe[k];
        c.r[6 + k] = c.vV[k] - cdv[k];
        c.r[9 + k] = sb_j[3 + k] - sb_i[3 + k];
        c.r[12 + k] = sb_j[6 + k] - sb_i[6 + k];
    }
    if (!JAC) return;
    double qj_inv[4], qji[4], qjid[4], L3[9], D[9];
    q_inv(pose_j + 3, qj_inv);
    q_mul(qj_inv, pose_i + 3, qji);
    qleft_qright3(qji, cdq, c.M1);
    q_mul(qji, pdq, qjid);
    qleft3(qjid, L3);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) D[i * 3 + j] = Jm[(3 + i) * 15 + 12 + j];
    m3_mul(L3, D, c.M2d);
    qleft3(qe, c.M3);
}

// raw Jacobian column `col` (0..29: pose_i 6 | sb_i 9 | pose_j 6 | sb_j 9) into out[15].
// Written with selects only (no run-time register-array index), so that ImuCtx and `out` stay in VGPRs.
DEV double sel3(double a0, double a1, double a2, int k) { return k == 0 ? a0 : (k == 1 ? a1 : a2); }
template <typename PP>
DEV void imu_raw_col(const ImuCtx& c, PP pre, int col, double* out) {
    const PP Jm = pre + IM_JAC;
    const int cb = col / 3, k = col - 3 * cb;
#pragma unroll
    for (int q = 0; q < 15; ++q) out[q] = 0.0;
    // column k of the 3x3 blocks that can appear
    const double ri0 = sel3(c.Rinv[0], c.Rinv[1], c.Rinv[2], k), ri1 = sel3(c.Rinv[3], c.Rinv[4], c.Rinv[5], k), ri2 = sel3(c.Rinv[6], c.Rinv[7], c.Rinv[8], k);
    if (cb == 0) { out[0] = -ri0; out[1] = -ri1; out[2] = -ri2; }
    else if (cb == 1) {
        // skew(v) column k: k=0 -> (0, v2, -v1); k=1 -> (-v2, 0, v0); k=2 -> (v1, -v0, 0)
        out[0] = sel3(0.0, -c.vP[2], c.vP[1], k); out[1] = sel3(c.vP[2], 0.0, -c.vP[0], k); out[2] = sel3(-c.vP[1], c.vP[0], 0.0, k);
        out[3] = -sel3(c.M1[0], c.M1[1], c.M1[2], k); out[4] = -sel3(c.M1[3], c.M1[4], c.M1[5], k); out[5] = -sel3(c.M1[6], c.M1[7], c.M1[8], k);
        out[6] = sel3(0.0, -c.vV[2], c.vV[1], k); out[7] = sel3(c.vV[2], 0.0, -c.vV[0], k); out[8] = sel3(-c.vV[1], c.vV[0], 0.0, k);
    } else if (cb == 2) {
        out[0] = -ri0 * c.dt; out[1] = -ri1 * c.dt; out[2] = -ri2 * c.dt;
        out[6] = -ri0; out[7] = -ri1; out[8] = -ri2;
    } else if (cb == 3) {
        out[0] = -Jm[0 * 15 + 9 + k]; out[1] = -Jm[1 * 15 + 9 + k]; out[2] = -Jm[2 * 15 + 9 + k];
        out[6] = -Jm[6 * 15 + 9 + k]; out[7] = -Jm[7 * 15 + 9 + k]; out[8] = -Jm[8 * 15 + 9 + k];
        out[9] = k == 0 ? -1.0 : 0.0; out[10] = k == 1 ? -1.0 : 0.0; out[11] = k == 2 ? -1.0 : 0.0;
    } else if (cb == 4) {
        out[0] = -Jm[0 * 15 + 12 + k]; out[1] = -Jm[1 * 15 + 12 + k]; out[2] = -Jm[2 * 15 + 12 + k];
        out[3] = -sel3(c.M2d[0], c.M2d[1], c.M2d[2], k); out[4] = -sel3(c.M2d[3], c.M2d[4], c.M2d[5], k); out[5] = -sel3(c.M2d[6], c.M2d[7], c.M2d[8], k);
        out[6] = -Jm[6 * 15 + 12 + k]; out[7] = -Jm[7 * 15 + 12 + k]; out[8] = -Jm[8 * 15 + 12 + k];
        out[12] = k == 0 ? -1.0 : 0.0; out[13] = k == 1 ? -1.0 : 0.0; out[14] = k == 2 ? -1.0 : 0.0;
    } else if (cb == 5) { out[0] = ri0; out[1] = ri1; out[2] = ri2; }
    else if (cb == 6) {
        out[3] = sel3(c.M3[0], c.M3[1], c.M3[2], k); out[4] = sel3(c.M3[3], c.M3[4], c.M3[5], k); out[5] = sel3(c.M3[6], c.M3[7], c.M3[8], k);
    } else if (cb == 7) { out[6] = ri0; out[7] = ri1; out[8] = ri2; }
    else if (cb == 8) { out[9] = k == 0 ? 1.0 : 0.0; out[10] = k == 1 ? 1.0 : 0.0; out[11] = k == 2 ? 1.0 : 0.0; }
    else { out[12] = k == 0 ? 1.0 : 0.0; out[13] = k == 1 ? 1.0 : 0.0; out[14] = k == 2 ? 1.0 : 0.0; }
}
