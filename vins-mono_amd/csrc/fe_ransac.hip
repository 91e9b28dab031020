// fe_ransac.hip — FeatureTracker::rejectWithF (feature_tracker/src/feature_tracker.cpp:169-202) on gfx950:
// cv::findFundamentalMat(un_cur_pts, un_forw_pts, cv::FM_RANSAC, F_THRESHOLD, 0.99, status)  (SURVEY.md 8(f) row 3).
//
// What is kept from OpenCV's RANSACPointSetRegistrator + FMEstimatorCallback ([3P], fundam.cpp / ptsetreg.cpp):
//   * the per-point error  max(d1^2 / |l1|^2, d2^2 / |l2|^2)  of the two point-to-epipolar-line distances, evaluated in
//     double, cast to float and compared with threshold^2;  the model with the most inliers wins (first one on ties);
//     the returned mask is the inlier set of that model (no final refit);
//   * Hartley normalisation of the sampled points, the linear solve for f as the null vector of the 9-column design
//     matrix, the rank-2 projection of F.
// What is NOT reproducible and is replaced, documented in oracle/ASSUMPTIONS.md (F9): OpenCV draws its samples from
// cv::RNG(-1) with an adaptive iteration count and solves 7-point cubics; here FE_RANSAC_HYP = 256 hypotheses are drawn
// by a counter-based generator (so the result is a pure function of the input), each from 8 points with the normalised
// 8-point algorithm.  All hypotheses run in parallel, one thread each: the 8 x 9 design matrix and the accumulated right
// singular vectors live in thread-private LDS columns ([element][thread], conflict-free), the null vector comes from a
// one-sided Jacobi SVD (as in triangulate.hip).
#include "vg_range.h"
#include <hip/hip_runtime.h>
#include <string>
#include <vector>
#include "ba_math.h"
#include "vg_handle.h"
#include "../../include/vinsgpu.h"

#define FE_RANSAC_HYP 256
#define FE_RANSAC_MAXPTS 1024

// counter-based generator: hash of (hypothesis, draw) — SplitMix64 finaliser
DEV unsigned long long fr_mix(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// 8 distinct indices in [0, n), n >= 8: draw d = 0, 1, ... of hypothesis k, skipping repeats
DEV void fr_sample(int k, int n, int* idx) {
    int have = 0;
    for (unsigned d = 0; have < 8; ++d) {
        const int c = (int)(fr_mix(((unsigned long long)(unsigned)k << 32) | d) % (unsigned long long)n);
        bool dup = false;
        for (int q = 0; q < have; ++q) dup = dup || idx[q] == c;
        if (!dup) idx[have++] = c;
    }
}

// error of correspondence (x1,y1) -> (x2,y2) under F (row-major 3x3), FMEstimatorCallback::computeError
DEV float fr_error(const double* f, double x1, double y1, double x2, double y2) {
    double a = f[0] * x1 + f[1] * y1 + f[2], b = f[3] * x1 + f[4] * y1 + f[5], c = f[6] * x1 + f[7] * y1 + f[8];
    const double s2 = 1.0 / (a * a + b * b), d2 = x2 * a + y2 * b + c;
    a = f[0] * x2 + f[3] * y2 + f[6]; b = f[1] * x2 + f[4] * y2 + f[7]; c = f[2] * x2 + f[5] * y2 + f[8];
    const double s1 = 1.0 / (a * a + b * b), d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 > e2 ? e1 : e2);
}

// hypotheses: F[k][9] and its inlier count
extern "C" __global__ __launch_bounds__(64) void fe_ransac_hyp_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int n,
                                                                      float thresh2, double* __restrict__ Fout, int* __restrict__ count) {
    __shared__ double A[72][64];          // design matrix, element (row r, col c) at A[r * 9 + c][thread]
    __shared__ double V[81][64];          // accumulated right singular vectors
    const int t = threadIdx.x, k = blockIdx.x * 64 + t;
    int idx[8];
    fr_sample(k, n, idx);
    // Hartley normalisation of the 8 sampled points of each image
    double c1x = 0, c1y = 0, c2x = 0, c2y = 0;
    for (int i = 0; i < 8; ++i) { c1x += p1[2 * idx[i]]; c1y += p1[2 * idx[i] + 1]; c2x += p2[2 * idx[i]]; c2y += p2[2 * idx[i] + 1]; }
    c1x /= 8; c1y /= 8; c2x /= 8; c2y /= 8;
    double s1 = 0, s2 = 0;
    for (int i = 0; i < 8; ++i) {
        const double ax = p1[2 * idx[i]] - c1x, ay = p1[2 * idx[i] + 1] - c1y, bx = p2[2 * idx[i]] - c2x, by = p2[2 * idx[i] + 1] - c2y;
        s1 += sqrt(ax * ax + ay * ay); s2 += sqrt(bx * bx + by * by);
    }
    bool degenerate = !(s1 > 1e-12) || !(s2 > 1e-12);
    s1 = degenerate ? 1.0 : 8.0 * 1.4142135623730951 / s1;
    s2 = degenerate ? 1.0 : 8.0 * 1.4142135623730951 / s2;
    for (int i = 0; i < 8; ++i) {
        const double x1 = (p1[2 * idx[i]] - c1x) * s1, y1 = (p1[2 * idx[i] + 1] - c1y) * s1;
        const double x2 = (p2[2 * idx[i]] - c2x) * s2, y2 = (p2[2 * idx[i] + 1] - c2y) * s2;
        const double row[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
        for (int c = 0; c < 9; ++c) A[i * 9 + c][t] = row[c];
    }
    for (int e = 0; e < 81; ++e) V[e][t] = (e % 10 == 0) ? 1.0 : 0.0;
    // one-sided Jacobi on the 9 columns: A V = U Sigma; the column that ends with the smallest norm spans the null space
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                double al = 0.0, be = 0.0, ga = 0.0;
                for (int r = 0; r < 8; ++r) { const double a = A[r * 9 + p][t], b = A[r * 9 + q][t]; al += a * a; be += b * b; ga += a * b; }
                if (fabs(ga) > 1e-15 * sqrt(al * be) && ga != 0.0) {
                    rotated = true;
                    const double zeta = (be - al) / (2.0 * ga);
                    const double tn = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double cs = 1.0 / sqrt(1.0 + tn * tn), sn = cs * tn;
                    for (int r = 0; r < 8; ++r) {
                        const double a = A[r * 9 + p][t], b = A[r * 9 + q][t];
                        A[r * 9 + p][t] = cs * a - sn * b; A[r * 9 + q][t] = sn * a + cs * b;
                    }
                    for (int r = 0; r < 9; ++r) {
                        const double a = V[r * 9 + p][t], b = V[r * 9 + q][t];
                        V[r * 9 + p][t] = cs * a - sn * b; V[r * 9 + q][t] = sn * a + cs * b;
                    }
                }
            }
        if (!rotated) break;
    }
    int bi = 0;
    double best = 0.0;
    for (int c = 0; c < 9; ++c) {
        double nn = 0.0;
        for (int r = 0; r < 8; ++r) { const double a = A[r * 9 + c][t]; nn += a * a; }
        if (c == 0 || nn < best) { best = nn; bi = c; }
    }
    double Fn[9];
    for (int e = 0; e < 9; ++e) Fn[e] = V[e * 9 + bi][t];
    // rank 2: F <- F (I - v3 v3^T), v3 = right singular vector of the smallest singular value = eigenvector of F^T F
    {
        double M[9], W[9];
        m3t_mul(Fn, Fn, M);                                   // F^T F
        for (int e = 0; e < 9; ++e) W[e] = (e % 4 == 0) ? 1.0 : 0.0;
        for (int sweep = 0; sweep < 30; ++sweep) {            // cyclic two-sided Jacobi on the symmetric 3x3
            double off = fabs(M[1]) + fabs(M[2]) + fabs(M[5]);
            if (!(off > 1e-300)) break;
            bool rotated = false;
#pragma unroll
            for (int pq = 0; pq < 3; ++pq) {
                const int p = pq == 2 ? 1 : 0, q = pq == 0 ? 1 : 2;
                const double apq = M[p * 3 + q];
                if (fabs(apq) <= 1e-17 * sqrt(fabs(M[p * 4] * M[q * 4])) || apq == 0.0) continue;
                rotated = true;
                const double th = (M[q * 4] - M[p * 4]) / (2.0 * apq);
                const double tn = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(1.0 + th * th));
                const double cs = 1.0 / sqrt(1.0 + tn * tn), sn = cs * tn;
                for (int r = 0; r < 3; ++r) {                 // M <- M J
                    const double a = M[r * 3 + p], b = M[r * 3 + q];
                    M[r * 3 + p] = cs * a - sn * b; M[r * 3 + q] = sn * a + cs * b;
                }
                for (int r = 0; r < 3; ++r) {                 // M <- J^T M
                    const double a = M[p * 3 + r], b = M[q * 3 + r];
                    M[p * 3 + r] = cs * a - sn * b; M[q * 3 + r] = sn * a + cs * b;
                }
                for (int r = 0; r < 3; ++r) {
                    const double a = W[r * 3 + p], b = W[r * 3 + q];
                    W[r * 3 + p] = cs * a - sn * b; W[r * 3 + q] = sn * a + cs * b;
                }
            }
            if (!rotated) break;
        }
        int mi = 0;
        if (M[4] < M[mi * 4]) mi = 1;
        if (M[8] < M[mi * 4]) mi = 2;
        const double v[3] = {W[mi], W[3 + mi], W[6 + mi]};
        for (int r = 0; r < 3; ++r) {
            const double fv = Fn[r * 3] * v[0] + Fn[r * 3 + 1] * v[1] + Fn[r * 3 + 2] * v[2];
            for (int c = 0; c < 3; ++c) Fn[r * 3 + c] -= fv * v[c];
        }
    }
    // de-normalise: F = T2^T Fn T1,  T = [s 0 -s cx; 0 s -s cy; 0 0 1]
    double F[9];
    {
        const double T1[9] = {s1, 0, -s1 * c1x, 0, s1, -s1 * c1y, 0, 0, 1};
        const double T2[9] = {s2, 0, -s2 * c2x, 0, s2, -s2 * c2y, 0, 0, 1};
        double tmp[9];
        m3_mul(Fn, T1, tmp);
        m3t_mul(T2, tmp, F);
    }
    int cnt = 0;
    bool finite = true;
    for (int e = 0; e < 9; ++e) finite = finite && (F[e] == F[e]) && fabs(F[e]) < 1e300;
    if (finite && !degenerate)
        for (int i = 0; i < n; ++i) cnt += fr_error(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= thresh2 ? 1 : 0;
    for (int e = 0; e < 9; ++e) Fout[(size_t)k * 9 + e] = F[e];
    count[k] = (finite && !degenerate) ? cnt : -1;
}

// best hypothesis (most inliers, lowest index on ties) and its inlier mask
extern "C" __global__ __launch_bounds__(FE_RANSAC_HYP) void fe_ransac_pick_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int n,
                                                                                 float thresh2, const double* __restrict__ Fall,
                                                                                 const int* __restrict__ count, unsigned char* __restrict__ status,
                                                                                 int* __restrict__ out) {
    __shared__ int key[FE_RANSAC_HYP];
    __shared__ double F[9];
    const int t = threadIdx.x;
    key[t] = count[t] * FE_RANSAC_HYP + (FE_RANSAC_HYP - 1 - t);          // max key = most inliers, then the lowest index
    __syncthreads();
    for (int s = FE_RANSAC_HYP / 2; s > 0; s >>= 1) {
        if (t < s) key[t] = key[t] > key[t + s] ? key[t] : key[t + s];
        __syncthreads();
    }
    const int bestk = FE_RANSAC_HYP - 1 - (key[0] % FE_RANSAC_HYP + FE_RANSAC_HYP) % FE_RANSAC_HYP;
    const bool any = key[0] >= 0;
    if (t < 9) F[t] = Fall[(size_t)bestk * 9 + t];
    __syncthreads();
    for (int i = t; i < n; i += FE_RANSAC_HYP)
        status[i] = (any && fr_error(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= thresh2) ? 1 : 0;
    if (t == 0) { out[0] = any ? bestk : -1; out[1] = any ? count[bestk] : 0; }
}

extern "C" int vg_fe_reject_with_f(vg_handle* h, const float* cur_un_xy, const float* forw_un_xy, int n, double threshold, uint8_t* status,
                                   int* n_inliers, double* F_out) {
    VG_RANGE("vg_fe_reject_with_f");
    if (!h || n < 0 || (n && (!cur_un_xy || !forw_un_xy || !status)) || !(threshold > 0)) return VG_ERR_BAD_ARG;
    if (n < 8) { h->err = "vg_fe_reject_with_f: fewer than 8 correspondences"; return VG_ERR_BAD_ARG; }
    if (n > FE_RANSAC_MAXPTS) { h->err = "vg_fe_reject_with_f: more than 1024 correspondences"; return VG_ERR_UNSUPPORTED; }
    hipError_t e = hipSetDevice(h->device);
    auto fail = [&](hipError_t err) {
        h->err = std::string("vg_fe_reject_with_f: ") + hipGetErrorString(err);
        return VG_ERR_HIP;
    };
    if (e != hipSuccess) return fail(e);
    // One allocation for the life of the handle (n <= FE_RANSAC_MAXPTS): this call sits on the per-frame path, and hipFree
    // synchronises the whole device — it would stall the BA handle's asynchronous marginalization / state download.
    const size_t off_F = sizeof(float) * 4 * FE_RANSAC_MAXPTS, off_i = off_F + sizeof(double) * 9 * FE_RANSAC_HYP;
    const size_t off_s = off_i + sizeof(int) * (FE_RANSAC_HYP + 2 + 2), total = off_s + FE_RANSAC_MAXPTS;
    if (!h->ransac_buf && (e = hipMalloc(&h->ransac_buf, total)) != hipSuccess) { h->ransac_buf = nullptr; return fail(e); }
    char* base = (char*)h->ransac_buf;
    float* d_p = (float*)base;
    double* d_F = (double*)(base + off_F);
    int* d_i = (int*)(base + off_i);
    unsigned char* d_s = (unsigned char*)(base + off_s);
    if ((e = hipMemcpyAsync(d_p, cur_un_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_p + 2 * n, forw_un_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
    const float thresh2 = (float)(threshold * threshold);
    hipLaunchKernelGGL(fe_ransac_hyp_kernel, dim3(FE_RANSAC_HYP / 64), dim3(64), 0, h->stream, d_p, d_p + 2 * n, n, thresh2, d_F, d_i);
    hipLaunchKernelGGL(fe_ransac_pick_kernel, dim3(1), dim3(FE_RANSAC_HYP), 0, h->stream, d_p, d_p + 2 * n, n, thresh2, d_F, d_i, d_s, d_i + FE_RANSAC_HYP);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    int res[2] = {-1, 0};
    if ((e = hipMemcpyAsync(status, d_s, n, hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(res, d_i + FE_RANSAC_HYP, sizeof(res), hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return fail(e);
    if (F_out) {
        if (res[0] >= 0) { if ((e = hipMemcpy(F_out, d_F + (size_t)res[0] * 9, sizeof(double) * 9, hipMemcpyDeviceToHost)) != hipSuccess) return fail(e); }
        else for (int k = 0; k < 9; ++k) F_out[k] = 0.0;
    }
    if (res[0] < 0) {
        // no usable hypothesis (every sample degenerate or non-finite): the estimate failed, nothing is rejected — the tracks
        // survive the frame instead of being wiped (documented choice, oracle/ASSUMPTIONS.md F9)
        for (int i = 0; i < n; ++i) status[i] = 1;
        res[1] = n;
    }
    if (n_inliers) *n_inliers = res[1];
    return VG_OK;
}
