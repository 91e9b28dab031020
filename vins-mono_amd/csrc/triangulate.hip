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
#include "tri_dlt.h"
#include "vg_handle.h"
#include "../../include/vinsgpu.h"

extern "C" __global__ __launch_bounds__(64) void triangulate_kernel(int K, const double* __restrict__ Ps, const double* __restrict__ Rs,
                                                                    const double* __restrict__ tic, const double* __restrict__ ric, int L,
                                                                    const int* __restrict__ start, const int* __restrict__ nobs,
                                                                    const int* __restrict__ obs_off, const double* __restrict__ points,
                                                                    double init_depth, double* __restrict__ depth) {
    __shared__ double A[TRI_ROWS * 4][64];
    const int t = threadIdx.x, l = blockIdx.x * 64 + t;
    if (l >= L) return;
    double Ric[9], Tic[3];
    for (int k = 0; k < 9; ++k) Ric[k] = ric[k];
    for (int k = 0; k < 3; ++k) Tic[k] = tic[k];
    const double* pts = points + 3 * (size_t)obs_off[l];
    depth[l] = tri_dlt_depth(A, t, Ps, Rs, Ric, Tic, start[l], nobs[l],
                             [pts](int j, double* p) { p[0] = pts[3 * j]; p[1] = pts[3 * j + 1]; p[2] = pts[3 * j + 2]; }, init_depth);
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
