// fe_ransac.hip — FeatureTracker::rejectWithF (feature_tracker/src/feature_tracker.cpp:169-202) on gfx950:
// cv::findFundamentalMat(un_cur_pts, un_forw_pts, cv::FM_RANSAC, F_THRESHOLD, 0.99, status)  (SURVEY.md 8(f) row 3).
//
// Follows OpenCV 3.3's RANSACPointSetRegistrator / LMeDSPointSetRegistrator + FMEstimatorCallback ([3P], fundam.cpp /
// ptsetreg.cpp, as recalled: oracle/ASSUMPTIONS.md F9 lists what is matched and what cannot be):
//   * the SAMPLE SCHEDULE is what OpenCV's sequential loop would draw: cv::RNG((uint64)-1), seven distinct rng.uniform(0, n)
//     indices per iteration, redrawn while the last point of a sample is collinear with two earlier ones.  It depends on
//     the points only, not on any model, so the host generates all of it up front (maxIters = 1000 iterations) ...
//   * ... and the device evaluates every iteration in parallel, one thread each: run7Point (null space of the 7 x 9 design
//     matrix of the raw points by one-sided Jacobi, the cubic det(x G + H) = 0 by cv::solveCubic, up to three models scaled
//     to F33 = 1), per model the inlier count (n >= 15: error max(d1^2/|l1|^2, d2^2/|l2|^2) as float <= threshold^2) or the
//     median error (8 <= n < 15: OpenCV falls back to LMedS there);
//   * the host then replays the sequential bookkeeping over the per-iteration results: a model replaces the best one if it
//     has more inliers than max(best, 6), after every improvement niters = RANSACUpdateNumIters(0.99, outlier ratio, 7,
//     niters), and iterations at or beyond niters do not count — the same model OpenCV's loop ends with;
//   * the mask is the inlier set of that model (LMedS: error <= (2.5 * 1.4826 (1 + 5 / (n - 7)) sqrt(median))^2).
// The models of one sample are ranked best first with a canonical tie order (OpenCV's visiting order follows the basis its
// SVD returns for the two-dimensional null space).  The null vectors live in thread-private LDS columns
// ([element][thread], conflict-free).
#include "vg_range.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <string>
#include <chrono>
#include <vector>
#include "ba_math.h"
#include "fe_layout.h"
#include "vg_handle.h"
#include "../../include/vinsgpu.h"

#define FE_LMEDS_MAXPTS 14

// error of correspondence (x1,y1) -> (x2,y2) under F (row-major 3x3), FMEstimatorCallback::computeError
DEV float fr_error(const double* f, double x1, double y1, double x2, double y2) {
    double a = f[0] * x1 + f[1] * y1 + f[2], b = f[3] * x1 + f[4] * y1 + f[5], c = f[6] * x1 + f[7] * y1 + f[8];
    const double s2 = 1.0 / (a * a + b * b), d2 = x2 * a + y2 * b + c;
    a = f[0] * x2 + f[3] * y2 + f[6]; b = f[1] * x2 + f[4] * y2 + f[7]; c = f[2] * x2 + f[5] * y2 + f[8];
    const double s1 = 1.0 / (a * a + b * b), d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 > e2 ? e1 : e2);
}

// cv::solveCubic for c0 x^3 + c1 x^2 + c2 x + c3 (the branch structure of OpenCV 3.3)
DEV int fr_solve_cubic(const double* cf, double* r) {
    double a0 = cf[0], a1 = cf[1], a2 = cf[2], a3 = cf[3];
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) return a3 == 0 ? -1 : 0;
            r[0] = -a3 / a2;
            return 1;
        }
        double d = a2 * a2 - 4 * a1 * a3;
        if (d < 0) return 0;
        d = sqrt(d);
        const double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
        if (fabs(q1) > fabs(q2)) { r[0] = q1 / a1; r[1] = a3 / q1; } else { r[0] = q2 / a1; r[1] = a3 / q2; }
        return d > 0 ? 2 : 1;
    }
    a0 = 1.0 / a0; a1 *= a0; a2 *= a0; a3 *= a0;
    const double Q = (a1 * a1 - 3 * a2) * (1.0 / 9), R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1.0 / 54);
    const double Qcubed = Q * Q * Q;
    double d = Qcubed - R * R;
    if (d >= 0) {
        const double theta = acos(R / sqrt(Qcubed)), sqrtQ = sqrt(Q);
        const double t0 = -2 * sqrtQ, t1 = theta * (1.0 / 3), t2 = a1 * (1.0 / 3);
        r[0] = t0 * cos(t1) - t2;
        r[1] = t0 * cos(t1 + (2.0 * 3.1415926535897932384626433832795 / 3)) - t2;
        r[2] = t0 * cos(t1 + (4.0 * 3.1415926535897932384626433832795 / 3)) - t2;
        return 3;
    }
    d = sqrt(-d);
    double e = pow(d + fabs(R), 0.333333333333);
    if (R > 0) e = -e;
    r[0] = (e + Q / e) - a1 * (1.0 / 3);
    return 1;
}
// canonical order of two models of one sample (scale- and sign-free): entries of F / (its entry of largest magnitude)
DEV bool fr_model_before(const double* Fa, const double* Fb) {
    double ma = 0, mb = 0;
    for (int e = 0; e < 9; ++e) { if (fabs(Fa[e]) > fabs(ma)) ma = Fa[e]; if (fabs(Fb[e]) > fabs(mb)) mb = Fb[e]; }
    for (int e = 0; e < 9; ++e) {
        const double va = ma != 0 ? Fa[e] / ma : Fa[e], vb = mb != 0 ? Fb[e] / mb : Fb[e];
        if (fabs(va - vb) > 1e-6) return va < vb;
    }
    return false;
}

// First half of an iteration of the schedule, one thread per sample: run7Point -> the up-to-three models of the sample
// (models[k][3][9]; fe_ransac_count_kernel ranks them: the best model of the sample (F[k][9]) and that model's inlier count
// (lmeds == 0) or median error (lmeds != 0, n <= FE_LMEDS_MAXPTS); count -1 = the sample gave no model).
// haveCollinearPoints() of the registrator for one point set: is the LAST point of the sample collinear with two earlier ones (the
// same double expressions as last_point_collinear() below; this file is compiled with -ffp-contract=off)
DEV bool fr_last_collinear(const float* __restrict__ p, const int* idx) {
    bool col = false;
    const double xi = p[2 * idx[6]], yi = p[2 * idx[6] + 1];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const double dx1 = (double)p[2 * idx[j]] - xi, dy1 = (double)p[2 * idx[j] + 1] - yi;
#pragma unroll
        for (int k = 0; k < j; ++k) {
            const double dx2 = (double)p[2 * idx[k]] - xi, dy2 = (double)p[2 * idx[k] + 1] - yi;
            col = col || fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2));
        }
    }
    return col;
}

// NINE LANES PER SAMPLE (round 5; one thread per sample before): lane c of a group owns column c of the 7 x 9 design matrix and of
// the accumulated right singular vectors, and a sweep of the one-sided Jacobi visits the 36 column pairs as 9 rounds of 4 disjoint
// pairs (round r pairs the columns with i + j = r mod 9; one column rests) -- the two lanes of a pair exchange their 16 numbers through
// ds_bpermute and both evaluate the rotation from the same operands in the same order, so they agree to the bit.  The launch has a
// wavefront per SIMD at most: what it costs is the LENGTH of the dependent instruction sequence, and that is a quarter of the
// one-thread form's (165 -> see DESIGN.md section 2 for 1000 samples).  Any basis of the two-dimensional null space gives the same
// pencil of models; the inlier masks are held to the oracle's two-sided Jacobi as before.
//
// `ctl` (vg_fe_read_image: the call runs without the host in between): the number of correspondences is ctl[RI_N1], the schedule
// is row n - 15 of the resident table of point-independent schedules, and a sample OpenCV would have REDRAWN (its last point
// collinear with two earlier ones, which depends on the points) raises ctl[RI_FALLBACK] -- the host then repeats the estimate
// with the exact schedule.  ctl == nullptr: the arguments are what they say (vg_fe_reject_with_f).
#define FR_GROUP 9                       // lanes per sample
#define FR_PER_WAVE (64 / FR_GROUP)      // 7 samples per wavefront; lane 63 idles
extern "C" __global__ __launch_bounds__(64) void fe_ransac7_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int n,
                                                                   const int* __restrict__ sched, int nsched, double* __restrict__ models,
                                                                   int* __restrict__ ctl) {
    if (ctl) {
        n = ctl[RI_N1];
        if (n < 15 || n > FE_RANSAC_MAXPTS || ctl[RI_PUBLISH] == 0) return;
        sched += (size_t)(n - 15) * 7 * FE_RANSAC_MAXIT;
    }
    const int lane = threadIdx.x, g = lane / FR_GROUP, c = lane - FR_GROUP * g;
    const int k = blockIdx.x * FR_PER_WAVE + g;
    const bool live = g < FR_PER_WAVE && k < nsched;
    const int gb = g < FR_PER_WAVE ? FR_GROUP * g : 64 - FR_GROUP;      // first lane of the group (lane 63: any valid lanes; its results are dropped)
    int idx[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) idx[i] = live ? sched[(size_t)k * 7 + i] : i;
    if (ctl && live && c == 0 && (fr_last_collinear(p1, idx) || fr_last_collinear(p2, idx))) atomicOr(&ctl[RI_FALLBACK], RI_FB_COLLINEAR);
    // column c of the design matrix (row i = [x1 x0, x1 y0, x1, y1 x0, y1 y0, y1, x0, y0, 1]) and of the identity
    double a[7], v[9];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const double x0 = p1[2 * idx[i]], y0 = p1[2 * idx[i] + 1], x1 = p2[2 * idx[i]], y1 = p2[2 * idx[i] + 1];
        const double u = c < 3 ? x1 : (c < 6 ? y1 : 1.0);
        const int cm = c - 3 * (c / 3);
        const double w = cm == 0 ? x0 : (cm == 1 ? y0 : 1.0);
        a[i] = u * w;
    }
#pragma unroll
    for (int e = 0; e < 9; ++e) v[e] = (e == c) ? 1.0 : 0.0;
    // one-sided Jacobi on the 9 columns: A V = U Sigma; the two columns that end with the smallest norms span the null space.
    // A has rank 7: two columns shrink to rounding noise, and a pair with such a column never passes the orthogonality test (noise
    // against noise) -- it is still rotated when its turn comes, but only rotations between two columns that carry signal (norm^2
    // above 1e-26 |A|_F^2) keep the sweeps going: ~7 sweeps instead of all 40.
    double scale2 = 0.0;
    {
        double nn = 0.0;
#pragma unroll
        for (int r = 0; r < 7; ++r) nn += a[r] * a[r];
#pragma unroll
        for (int q = 0; q < FR_GROUP; ++q) scale2 += __shfl(nn, gb + q);
    }
    const double signal = 1e-26 * scale2;
    const unsigned long long gmask = 0x1ffull << gb;
    bool active = live;                                    // (uniform over the group; lane 63 and the samples past the schedule rest)
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool rotated = false;
#pragma unroll 1
        for (int rd = 0; rd < FR_GROUP; ++rd) {
            int p = rd - c;
            p = p < 0 ? p + FR_GROUP : p;
            const bool lo = c < p;
            double b[7], w[9];
#pragma unroll
            for (int r = 0; r < 7; ++r) b[r] = __shfl(a[r], gb + p);
#pragma unroll
            for (int e = 0; e < 9; ++e) w[e] = __shfl(v[e], gb + p);
            // (al, be, ga) of the pair as the column with the smaller index sees them: both lanes form the same sums
            double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const double x = lo ? a[r] : b[r], y = lo ? b[r] : a[r];
                al += x * x; be += y * y; ga += x * y;
            }
            if (active && p != c && fabs(ga) > 1e-15 * sqrt(al * be) && ga != 0.0) {
                rotated = rotated || (al > signal && be > signal);
                const double zeta = (be - al) / (2.0 * ga);
                const double tn = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + tn * tn), sn = cs * tn;
                // column lo: cs x - sn y; column hi: sn x + cs y  (x = the lo column, y = the hi column)
#pragma unroll
                for (int r = 0; r < 7; ++r) a[r] = lo ? cs * a[r] - sn * b[r] : sn * b[r] + cs * a[r];
#pragma unroll
                for (int e = 0; e < 9; ++e) v[e] = lo ? cs * v[e] - sn * w[e] : sn * w[e] + cs * v[e];
            }
        }
        active = active && (__ballot(rotated) & gmask) != 0ull;
        if (!__any(active)) break;
    }
    int i2 = 0, i1 = -1;                                  // i2: smallest column norm, i1: second smallest
    {
        double nn = 0.0;
#pragma unroll
        for (int r = 0; r < 7; ++r) nn += a[r] * a[r];
        double nrm[9];
#pragma unroll
        for (int q = 0; q < FR_GROUP; ++q) nrm[q] = __shfl(nn, gb + q);
        double n2 = 0.0, n1 = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q)
            if (q == 0 || nrm[q] < n2) { n2 = nrm[q]; i2 = q; }
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            if (q == i2) continue;
            if (i1 < 0 || nrm[q] < n1) { n1 = nrm[q]; i1 = q; }
        }
    }
    // the two null vectors = columns i2 and i1 of V, fetched from the lanes that own them; from here on every lane of the group computes
    // the same numbers and lane 0 of the group stores them
    double f1[9], f2[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) {
        const double v2 = __shfl(v[e], gb + i2), v1 = __shfl(v[e], gb + i1);
        f2[e] = v2; f1[e] = v1 - v2;
    }
    double cf[4];
    {
        double t0 = f2[4] * f2[8] - f2[5] * f2[7], t1 = f2[3] * f2[8] - f2[5] * f2[6], t2 = f2[3] * f2[7] - f2[4] * f2[6];
        cf[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
        cf[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) + f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) -
                f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) + f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
                f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
        t0 = f1[4] * f1[8] - f1[5] * f1[7]; t1 = f1[3] * f1[8] - f1[5] * f1[6]; t2 = f1[3] * f1[7] - f1[4] * f1[6];
        cf[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
        cf[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) + f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) -
                f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) + f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
                f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    }
    double roots[3] = {0, 0, 0};
    const int nr = fr_solve_cubic(cf, roots);
    // the up-to-three models of the sample, in the order of the roots; a model that is not finite is marked by F[0] = NaN
    for (int m = 0; m < 3; ++m) {
        double F[9];
        bool ok = (nr >= 1 && nr <= 3) && m < nr;
        if (ok) {
            double lambda = roots[m], mu = 1.0;
            const double sc = f1[8] * roots[m] + f2[8];
            if (fabs(sc) > 2.220446049250313e-16) { mu = 1.0 / sc; lambda *= mu; F[8] = 1.0; } else F[8] = 0.0;
            for (int e = 0; e < 8; ++e) F[e] = f1[e] * lambda + f2[e] * mu;
            for (int e = 0; e < 9; ++e) ok = ok && (F[e] == F[e]) && fabs(F[e]) < 1e300;
        }
        if (live && c == 0)
            for (int e = 0; e < 9; ++e) models[((size_t)k * 3 + m) * 9 + e] = ok ? F[e] : __builtin_nan("");
    }
}

// Second half of an iteration, one WAVEFRONT per sample (round 4: the thread that solved the sample used to walk all n points for each
// of its models, 3 x 150 error evaluations in sequence -- about half of the 0.4 ms of the call): the lanes share the points, a model's
// inlier count is a sum of ballots, the selection among the models (most inliers, canonical order on ties: exactly the sequence of
// comparisons of the one-thread form) is evaluated uniformly, and the inlier set of the chosen model leaves as ceil(n / 64) ballot words
// so that the host needs no second kernel for the mask.  LMedS (n < 15): lane 0 ranks the errors as before.
extern "C" __global__ __launch_bounds__(64) void fe_ransac_count_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int n, float thresh2,
                                                                        int lmeds, const double* __restrict__ models, int nsched,
                                                                        double* __restrict__ Fout, int* __restrict__ count, double* __restrict__ median,
                                                                        unsigned long long* __restrict__ inl_words, const int* __restrict__ ctl) {
    const int k = blockIdx.x, lane = threadIdx.x;
    if (ctl) {
        n = ctl[RI_N1];
        if (n < 15 || n > FE_RANSAC_MAXPTS || ctl[RI_PUBLISH] == 0) return;
    }
    if (k >= nsched) return;
    const int nw = (n + 63) >> 6;
    double bestF[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int bgood = -1;
    double bmed = 0.0;
    bool have = false;
    for (int m = 0; m < 3; ++m) {
        double F[9];
        for (int e = 0; e < 9; ++e) F[e] = models[((size_t)k * 3 + m) * 9 + e];
        if (!(F[0] == F[0])) continue;                      // (uniform: no such model)
        if (!lmeds) {
            int good = 0;
            for (int w = 0; w < nw; ++w) {
                const int i = 64 * w + lane;
                const bool in = i < n && fr_error(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= thresh2;
                good += __popcll(__ballot(in));
            }
            if (!have || good > bgood || (good == bgood && fr_model_before(F, bestF))) {
                bgood = good; have = true;
                for (int e = 0; e < 9; ++e) bestF[e] = F[e];
            }
        } else {
            float er[FE_LMEDS_MAXPTS];
            for (int i = 0; i < FE_LMEDS_MAXPTS; ++i) er[i] = i < n ? fr_error(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) : 3.0e38f;
            bool nan = false;
            for (int i = 0; i < FE_LMEDS_MAXPTS; ++i) nan = nan || (i < n && !(er[i] == er[i]));
            // selection by rank (no dynamically indexed sort of a register array): median = element(s) of rank n/2 (and n/2 - 1)
            float lo = 0.f, hi = 0.f;
            for (int i = 0; i < FE_LMEDS_MAXPTS; ++i) {
                if (i >= n) continue;
                int rk = 0;
                for (int j = 0; j < FE_LMEDS_MAXPTS; ++j) rk += (j < n && (er[j] < er[i] || (er[j] == er[i] && j < i))) ? 1 : 0;
                if (rk == n / 2) hi = er[i];
                if (rk == n / 2 - 1) lo = er[i];
            }
            double med = (n & 1) ? (double)hi : (double)(lo + hi) * 0.5;
            if (nan) continue;
            // (n <= 13: the median of a model that fits its 7 sample points exactly lies inside the fitted set and is rounding noise;
            //  snapped to zero so that the FIRST such sample wins instead of noise: oracle/ASSUMPTIONS.md F9)
            if (med < 1e-12) med = 0.0;
            if (!have || med < bmed || (med == bmed && fr_model_before(F, bestF))) {
                bmed = med; have = true; bgood = 0;
                for (int e = 0; e < 9; ++e) bestF[e] = F[e];
            }
        }
    }
    if (have && !lmeds)
        for (int w = 0; w < nw; ++w) {
            const int i = 64 * w + lane;
            const bool in = i < n && fr_error(bestF, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= thresh2;
            const unsigned long long bal = __ballot(in);
            if (lane == 0) inl_words[(size_t)k * nw + w] = bal;
        }
    if (lane == 0) {
        for (int e = 0; e < 9; ++e) Fout[(size_t)k * 9 + e] = bestF[e];
        count[k] = have ? bgood : -1;
        median[k] = bmed;
    }
}

// inlier mask of model F[sel] against thresh2
extern "C" __global__ __launch_bounds__(256) void fe_ransac_mask_kernel(const float* __restrict__ p1, const float* __restrict__ p2, int n, float thresh2,
                                                                       const double* __restrict__ Fall, int sel, unsigned char* __restrict__ status) {
    __shared__ double F[9];
    if (threadIdx.x < 9) F[threadIdx.x] = Fall[(size_t)sel * 9 + threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) status[i] = fr_error(F, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= thresh2 ? 1 : 0;
}

// ---- host side: OpenCV's sample schedule and its sequential bookkeeping --------------------------------------------------
namespace {
struct CvRng {                                        // cv::RNG
    unsigned long long state;
    explicit CvRng(unsigned long long s) : state(s ? s : 0xffffffffull) {}
    unsigned next() { state = (unsigned long long)(unsigned)state * 4164903690ull + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};
// haveCollinearPoints(): is the LAST point of the sample collinear with two earlier ones
bool last_point_collinear(const float* p, const int* idx, int count) {
    const int i = count - 1;
    for (int j = 0; j < i; ++j) {
        const double dx1 = (double)p[2 * idx[j]] - p[2 * idx[i]], dy1 = (double)p[2 * idx[j] + 1] - p[2 * idx[i] + 1];
        for (int k = 0; k < j; ++k) {
            const double dx2 = (double)p[2 * idx[k]] - p[2 * idx[i]], dy2 = (double)p[2 * idx[k] + 1] - p[2 * idx[i] + 1];
            if (std::fabs(dx2 * dy1 - dy2 * dx1) <= 1.1920928955078125e-07 * (std::fabs(dx1) + std::fabs(dy1) + std::fabs(dx2) + std::fabs(dy2))) return true;
        }
    }
    return false;
}
// PointSetRegistrator getSubset() (checkPartialSubsets == false)
bool get_subset(CvRng& rng, const float* p1, const float* p2, int n, int* idx, int max_attempts) {
    int iters = 0, i = 0;
    for (; iters < max_attempts; ++iters) {
        for (i = 0; i < 7 && iters < max_attempts;) {
            for (;;) {
                const int c = idx[i] = rng.uniform(0, n);
                int j = 0;
                for (; j < i; ++j) if (c == idx[j]) break;
                if (j == i) break;
            }
            ++i;
        }
        if (i == 7 && p1 && (last_point_collinear(p1, idx, 7) || last_point_collinear(p2, idx, 7))) continue;     // (p1 == nullptr: the part of the schedule that depends on n only)
        break;
    }
    return i == 7 && iters < max_attempts;
}
int update_num_iters(double p, double ep, int model_points, int max_iters) {         // RANSACUpdateNumIters
    p = std::min(std::max(p, 0.0), 1.0);
    ep = std::min(std::max(ep, 0.0), 1.0);
    double num = std::max(1.0 - p, 2.2250738585072014e-308);
    double denom = 1.0 - std::pow(1.0 - ep, model_points);
    if (denom < 2.2250738585072014e-308) return 0;
    num = std::log(num); denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::nearbyint(num / denom);
}
}  // namespace

// Device scratch of the estimate: one allocation for the life of the handle (n <= FE_RANSAC_MAXPTS).  The call sits on the per-frame
// path, and hipFree synchronises the whole device — it would stall the BA handle's asynchronous marginalization / state download.
hipError_t fe_ransac_buffers(vg_handle* h, FeRansacBufs* b) {
    const size_t off_F = sizeof(float) * 4 * FE_RANSAC_MAXPTS, off_m = off_F + sizeof(double) * 9 * FE_RANSAC_MAXIT;
    const size_t off_i = off_m + sizeof(double) * FE_RANSAC_MAXIT, off_sc = off_i + sizeof(int) * FE_RANSAC_MAXIT;
    const size_t off_s = off_sc + sizeof(int) * 7 * FE_RANSAC_MAXIT, off_md = (off_s + FE_RANSAC_MAXPTS + 255) / 256 * 256;
    const size_t off_w = off_md + sizeof(double) * 27 * FE_RANSAC_MAXIT, total = off_w + sizeof(unsigned long long) * (FE_RANSAC_MAXPTS / 64) * FE_RANSAC_MAXIT;
    if (!h->ransac_buf) {
        const hipError_t e = hipMalloc(&h->ransac_buf, total);
        if (e != hipSuccess) { h->ransac_buf = nullptr; return e; }
    }
    char* base = (char*)h->ransac_buf;
    b->p = (float*)base; b->F = (double*)(base + off_F); b->med = (double*)(base + off_m); b->cnt = (int*)(base + off_i);
    b->sched = (int*)(base + off_sc); b->s = (unsigned char*)(base + off_s); b->models = (double*)(base + off_md);
    b->words = (unsigned long long*)(base + off_w);
    return hipSuccess;
}

// What vg_fe_read_image keeps resident so that rejectWithF needs no host in the middle of a frame: for every n in [15, nmax] the sample
// schedule as far as it depends on n alone (the draws of cv::RNG((uint64)-1) and the redraws of an index already in the sample; the
// collinearity redraws depend on the points: fe_ransac7_kernel detects a sample that needed one) and, by inlier count c, the iteration
// bound RANSACUpdateNumIters(0.99, (n - c) / n, 7, 1000) -- computed HERE, with the host's libm, so that the device's bookkeeping takes
// the very decisions of the host's.  sched: [(nmax - 14)][FE_RANSAC_MAXIT][7], niters: [(nmax + 1)][stride].
void fe_ransac_tables(int nmax, std::vector<int>& sched, std::vector<int>& niters, int stride) {
    sched.assign((size_t)std::max(nmax - 14, 0) * FE_RANSAC_MAXIT * 7, 0);
    niters.assign((size_t)(nmax + 1) * stride, FE_RANSAC_MAXIT);
    for (int n = 15; n <= nmax; ++n) {
        CvRng rng((unsigned long long)-1);
        int* row = sched.data() + (size_t)(n - 15) * FE_RANSAC_MAXIT * 7;
        for (int it = 0; it < FE_RANSAC_MAXIT; ++it) get_subset(rng, nullptr, nullptr, n, row + (size_t)it * 7, 10000);
        for (int c = 0; c <= n; ++c) niters[(size_t)n * stride + c] = update_num_iters(0.99, (double)(n - c) / n, 7, FE_RANSAC_MAXIT);
    }
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
    // the schedule: what OpenCV's loop would draw in its first maxIters iterations (it depends on the points only)
    static const bool debug_phases = getenv("VG_DEBUG_RANSAC") != nullptr;      // phase times of this call on stderr
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    const bool lmeds = n < 15;                         // findFundamentalMat: RANSAC needs 15 points, LMedS otherwise
    const int maxit = lmeds ? std::max(update_num_iters(0.99, 0.45, 7, 1000), 3) : FE_RANSAC_MAXIT;
    std::vector<int> sched((size_t)maxit * 7);
    int nsched = 0;
    {
        CvRng rng((unsigned long long)-1);
        for (; nsched < maxit; ++nsched)
            if (!get_subset(rng, cur_un_xy, forw_un_xy, n, sched.data() + (size_t)nsched * 7, lmeds ? 1000 : 10000)) break;
    }
    const double ms_sched = since(t_begin);
    FeRansacBufs rb;
    if ((e = fe_ransac_buffers(h, &rb)) != hipSuccess) return fail(e);
    float* d_p = rb.p;
    double *d_F = rb.F, *d_med = rb.med, *d_models = rb.models;
    int *d_cnt = rb.cnt, *d_sched = rb.sched;
    unsigned char* d_s = rb.s;
    unsigned long long* d_words = rb.words;
    int best = -1, n_in = n;
    double t2_mask = threshold * threshold;
    if (nsched > 0) {
        if ((e = hipMemcpyAsync(d_p, cur_un_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
        if ((e = hipMemcpyAsync(d_p + 2 * n, forw_un_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
        if ((e = hipMemcpyAsync(d_sched, sched.data(), sizeof(int) * 7 * nsched, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
        const float thresh2 = (float)(threshold * threshold);
        hipLaunchKernelGGL(fe_ransac7_kernel, dim3((nsched + 6) / 7), dim3(64), 0, h->stream, d_p, d_p + 2 * n, n, d_sched, nsched, d_models, (int*)nullptr);      // 7 samples per wavefront
        hipLaunchKernelGGL(fe_ransac_count_kernel, dim3(nsched), dim3(64), 0, h->stream, d_p, d_p + 2 * n, n, thresh2, lmeds ? 1 : 0, d_models, nsched,
                           d_F, d_cnt, d_med, d_words, (const int*)nullptr);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e);
        const int nw = (n + 63) / 64;
        std::vector<int> cnt(nsched);
        std::vector<double> med(nsched), Fall(lmeds ? 0 : (size_t)nsched * 9);
        std::vector<unsigned long long> words(lmeds ? 0 : (size_t)nsched * nw);
        if ((e = hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int) * nsched, hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return fail(e);
        if ((e = hipMemcpyAsync(med.data(), d_med, sizeof(double) * nsched, hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return fail(e);
        // RANSAC: the inlier sets and the models of all iterations come along (a few tens of KB): once the bookkeeping below has picked
        // the iteration, its mask is already here -- no second kernel, no second synchronisation
        if (!lmeds) {
            if ((e = hipMemcpyAsync(words.data(), d_words, sizeof(unsigned long long) * words.size(), hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return fail(e);
            if ((e = hipMemcpyAsync(Fall.data(), d_F, sizeof(double) * Fall.size(), hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return fail(e);
        }
        if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return fail(e);
        if (debug_phases) fprintf(stderr, "[ransac] n %d, schedule of %d samples %.3f ms (host), upload + kernel + download %.3f ms\n", n, nsched, ms_sched, since(t_begin) - ms_sched);
        // the sequential bookkeeping of the registrator over the per-iteration results
        if (!lmeds) {
            int niters = FE_RANSAC_MAXIT, max_good = 0;
            for (int it = 0; it < nsched && it < niters; ++it) {
                if (cnt[it] < 0) continue;
                if (cnt[it] > std::max(max_good, 6)) {
                    best = it; max_good = cnt[it];
                    niters = update_num_iters(0.99, (double)(n - cnt[it]) / n, 7, niters);
                }
            }
        } else {
            double min_median = 1.7976931348623157e308;
            for (int it = 0; it < nsched; ++it) {
                if (cnt[it] < 0) continue;
                if (med[it] < min_median) { min_median = med[it]; best = it; }
            }
            if (best >= 0) {
                const double sigma = std::max(2.5 * 1.4826 * (1 + 5.0 / (n - 7)) * std::sqrt(min_median), 0.001);
                t2_mask = sigma * sigma;
            }
        }
        if (best >= 0 && !lmeds) {
            n_in = 0;
            for (int i = 0; i < n; ++i) { status[i] = (unsigned char)((words[(size_t)best * nw + (i >> 6)] >> (i & 63)) & 1ull); n_in += status[i]; }
            if (F_out) for (int q = 0; q < 9; ++q) F_out[q] = Fall[(size_t)best * 9 + q];
        } else if (best >= 0) {
            hipLaunchKernelGGL(fe_ransac_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, d_p, d_p + 2 * n, n, (float)t2_mask, d_F, best, d_s);
            if ((e = hipGetLastError()) != hipSuccess) return fail(e);
            if ((e = hipMemcpyAsync(status, d_s, n, hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return fail(e);
            if (F_out && (e = hipMemcpyAsync(F_out, d_F + (size_t)best * 9, sizeof(double) * 9, hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return fail(e);
            if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return fail(e);
            n_in = 0;
            for (int i = 0; i < n; ++i) n_in += status[i];
            if (lmeds && n_in < 7) best = -1;          // LMeDS reports failure below modelPoints inliers
        }
    }
    if (best < 0) {
        // no model (no valid sample, no finite solution): the estimate failed, nothing is rejected — the tracks survive the frame
        // instead of being wiped (documented choice, oracle/ASSUMPTIONS.md F9)
        for (int i = 0; i < n; ++i) status[i] = 1;
        n_in = n;
        if (F_out) for (int k = 0; k < 9; ++k) F_out[k] = 0.0;
    }
    if (n_inliers) *n_inliers = n_in;
    return VG_OK;
}
