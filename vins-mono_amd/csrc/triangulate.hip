// triangulate.hip — FeatureManager::triangulate (vins_estimator/src/feature_manager.cpp:202-257) on gfx950:
// the initial inverse-depth source of every landmark that Estimator::optimization() then refines
// (solveOdometry, estimator.cpp:596-598; SURVEY.md 8(f) row 4, first half).
//
// One thread per landmark: the 2n x 4 DLT matrix (rows f0*P2 - f2*P0, f1*P2 - f2*P1 of the camera matrices relative
// to the first observing frame, :222-241) lives in a thread-private LDS column ([element][thread], conflict-free);
// its right singular vector of the smallest singular value comes from a one-sided (Hestenes) Jacobi SVD on the four
// columns — the same quantity the reference takes from Eigen::JacobiSVD(...).matrixV().rightCols<1>() (:244);
// depth = v[2] / v[3], replaced by INIT_DEPTH when < 0.1 (:245-254).
#include "vg_range.h"
#include <hip/hip_runtime.h>
#include <vector>
#include "ba_math.h"
#include "vg_handle.h"
#include "../../include/vinsgpu.h"

#define TRI_MAXOBS 12
#define TRI_ROWS (2 * TRI_MAXOBS)

extern "C" __global__ __launch_bounds__(64) void triangulate_kernel(int K, const double* __restrict__ Ps, const double* __restrict__ Rs,
                                                                    const double* __restrict__ tic, const double* __restrict__ ric, int L,
                                                                    const int* __restrict__ start, const int* __restrict__ nobs,
                                                                    const int* __restrict__ obs_off, const double* __restrict__ points,
                                                                    double init_depth, double* __restrict__ depth) {
    __shared__ double A[TRI_ROWS * 4][64];
    const int t = threadIdx.x, l = blockIdx.x * 64 + t;
    if (l >= L) return;
    const int i0 = start[l], n = nobs[l];
    const int m = 2 * n;
    double Ric[9], Tic[3];
    for (int k = 0; k < 9; ++k) Ric[k] = ric[k];
    for (int k = 0; k < 3; ++k) Tic[k] = tic[k];
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
        const double* pt = points + 3 * (size_t)(obs_off[l] + j);
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
    depth[l] = dep;
}

extern "C" int vg_triangulate(vg_handle* h, int K, const double* Ps, const double* Rs, const double* tic, const double* ric, int L,
                              const int* start, const int* nobs, const int* obs_off, const double* points, double init_depth,
                              double* depth) {
    VG_RANGE("vg_triangulate");
    if (!h || K < 1 || !Ps || !Rs || !tic || !ric || L < 0 || (L > 0 && (!start || !nobs || !obs_off || !points || !depth))) return VG_ERR_BAD_ARG;
    if (L == 0) return VG_OK;
    size_t npt = 0;
    for (int l = 0; l < L; ++l) {
        if (nobs[l] < 1 || start[l] < 0 || start[l] + nobs[l] > K || obs_off[l] < 0) { h->err = "vg_triangulate: inconsistent track table"; return VG_ERR_BAD_ARG; }
        if (nobs[l] > TRI_MAXOBS) { h->err = "vg_triangulate: more than 12 observations per landmark"; return VG_ERR_UNSUPPORTED; }
        npt = std::max(npt, (size_t)obs_off[l] + nobs[l]);
    }
    hipError_t e = hipSetDevice(h->device);
    double *d_ps = nullptr, *d_rs = nullptr, *d_ext = nullptr, *d_pts = nullptr, *d_dep = nullptr;
    int* d_tab = nullptr;
    auto fail = [&](hipError_t err) {
        h->err = std::string("vg_triangulate: ") + hipGetErrorString(err);
        (void)hipFree(d_ps); (void)hipFree(d_rs); (void)hipFree(d_ext); (void)hipFree(d_pts); (void)hipFree(d_dep); (void)hipFree(d_tab);
        return VG_ERR_HIP;
    };
    if (e != hipSuccess) return fail(e);
    std::vector<double> ext(12);
    for (int k = 0; k < 3; ++k) ext[k] = tic[k];
    for (int k = 0; k < 9; ++k) ext[3 + k] = ric[k];
    std::vector<int> tab((size_t)3 * L);
    for (int l = 0; l < L; ++l) { tab[l] = start[l]; tab[L + l] = nobs[l]; tab[2 * L + l] = obs_off[l]; }
    if ((e = hipMalloc((void**)&d_ps, sizeof(double) * 3 * K)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&d_rs, sizeof(double) * 9 * K)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&d_ext, sizeof(double) * 12)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&d_pts, sizeof(double) * 3 * npt)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&d_dep, sizeof(double) * L)) != hipSuccess) return fail(e);
    if ((e = hipMalloc((void**)&d_tab, sizeof(int) * 3 * L)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_ps, Ps, sizeof(double) * 3 * K, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_rs, Rs, sizeof(double) * 9 * K, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_ext, ext.data(), sizeof(double) * 12, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_pts, points, sizeof(double) * 3 * npt, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_tab, tab.data(), sizeof(int) * 3 * L, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
    hipLaunchKernelGGL(triangulate_kernel, dim3((L + 63) / 64), dim3(64), 0, h->stream, K, d_ps, d_rs, d_ext, d_ext + 3, L, d_tab, d_tab + L,
                       d_tab + 2 * L, d_pts, init_depth, d_dep);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(depth, d_dep, sizeof(double) * L, hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return fail(e);
    (void)hipFree(d_ps); (void)hipFree(d_rs); (void)hipFree(d_ext); (void)hipFree(d_pts); (void)hipFree(d_dep); (void)hipFree(d_tab);
    return VG_OK;
}
