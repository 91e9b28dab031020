// tri_dlt.h — the per-landmark DLT triangulation of FeatureManager::triangulate (feature_manager.cpp:216-254), shared by
// triangulate_kernel (vg_triangulate) and ba_seq_tri_kernel (windows that stay on the device).  One thread per landmark: the
// 2n x 4 DLT matrix (rows f0*P2 - f2*P0, f1*P2 - f2*P1 of the camera matrices relative to the first observing frame, :222-241)
// lives in a thread-private LDS column ([element][thread], conflict-free); its right singular vector of the smallest singular
// value comes from a one-sided (Hestenes) Jacobi SVD on the four columns -- the quantity the reference takes from
// Eigen::JacobiSVD(...).matrixV().rightCols<1>() (:244); depth = v[2] / v[3], replaced by INIT_DEPTH when < 0.1 (:245-254).
#pragma once
#include "ba_math.h"

#define TRI_MAXOBS 12
#define TRI_ROWS (2 * TRI_MAXOBS)

// A: [TRI_ROWS * 4][64] doubles of LDS, t: this thread's column.  Ps [K x 3], Rs [K x 9 row-major], Ric / Tic: extrinsics.
// The landmark is observed in frames i0 .. i0 + n - 1; point(j, p) fills p[3] with feature_per_frame[j].point.
template <typename PointFn>
DEV double tri_dlt_depth(double (*A)[64], int t, const double* Ps, const double* Rs, const double* Ric, const double* Tic, int i0, int n,
                         PointFn point, double init_depth) {
    const int m = 2 * n;
    double R0[9], t0[3], tmp[3];
    m3_mul(Rs + 9 * i0, Ric, R0);                          // R0 = Rs[imu_i] * ric[0]
    m3_vec(Rs + 9 * i0, Tic, tmp);
    for (int k = 0; k < 3; ++k) t0[k] = Ps[3 * i0 + k] + tmp[k];
    for (int j = 0; j < n; ++j) {
        const int f = i0 + j;
        double R1[9], t1[3], d[3], tt[3], R[9];
        m3_mul(Rs + 9 * f, Ric, R1);
        m3_vec(Rs + 9 * f, Tic, tmp);
        for (int k = 0; k < 3; ++k) { t1[k] = Ps[3 * f + k] + tmp[k]; d[k] = t1[k] - t0[k]; }
        m3t_vec(R0, d, tt);                                // t = R0^T (t1 - t0)
        m3t_mul(R0, R1, R);                                // R = R0^T R1
        // P = [R^T | -R^T t]
        double P[12];
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) P[r * 4 + c] = R[c * 3 + r];
            P[r * 4 + 3] = -(R[0 * 3 + r] * tt[0] + R[1 * 3 + r] * tt[1] + R[2 * 3 + r] * tt[2]);
        }
        double pt[3];
        point(j, pt);
        const double nrm = sqrt(pt[0] * pt[0] + pt[1] * pt[1] + pt[2] * pt[2]);
        const double f0 = pt[0] / nrm, f1 = pt[1] / nrm, f2 = pt[2] / nrm;          // point.normalized()
        for (int c = 0; c < 4; ++c) {
            A[(2 * j) * 4 + c][t] = f0 * P[8 + c] - f2 * P[0 + c];
            A[(2 * j + 1) * 4 + c][t] = f1 * P[8 + c] - f2 * P[4 + c];
        }
    }
    // ---- one-sided Jacobi SVD on the 4 columns; V accumulated in registers
    double V[16];
    for (int k = 0; k < 16; ++k) V[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double alpha = 0.0, beta = 0.0, gamma = 0.0;
                for (int r = 0; r < m; ++r) { const double a = A[r * 4 + p][t], b = A[r * 4 + q][t]; alpha += a * a; beta += b * b; gamma += a * b; }
                if (fabs(gamma) > 1e-16 * sqrt(alpha * beta) && gamma != 0.0) {
                    rotated = true;
                    const double zeta = (beta - alpha) / (2.0 * gamma);
                    const double tn = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double c = 1.0 / sqrt(1.0 + tn * tn), s = c * tn;
                    for (int r = 0; r < m; ++r) {
                        const double a = A[r * 4 + p][t], b = A[r * 4 + q][t];
                        A[r * 4 + p][t] = c * a - s * b; A[r * 4 + q][t] = s * a + c * b;
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const double a = V[r * 4 + p], b = V[r * 4 + q];
                        V[r * 4 + p] = c * a - s * b; V[r * 4 + q] = s * a + c * b;
                    }
                }
            }
        if (!rotated) break;
    }
    double best = 0.0;
    int bi = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        double s2 = 0.0;
        for (int r = 0; r < m; ++r) { const double a = A[r * 4 + c][t]; s2 += a * a; }
        if (c == 0 || s2 < best) { best = s2; bi = c; }
    }
    double v2 = 0.0, v3 = 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) if (c == bi) { v2 = V[2 * 4 + c]; v3 = V[3 * 4 + c]; }
    double dep = v2 / v3;
    if (dep < 0.1) dep = init_depth;                       // (a NaN compares false and is kept, as in the reference)
    return dep;
}
