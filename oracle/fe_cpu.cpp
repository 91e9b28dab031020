// fe_cpu.cpp — dependency-free C++17 restatement of the feature_tracker front end's arithmetic
// (TEST INFRASTRUCTURE + bench.py's timed cpu_baseline; never linked into the product).
//
// PARITY UNPINNED.  FeatureTracker::readImage (feature_tracker/src/feature_tracker.cpp:81-167) does no
// arithmetic itself: it calls OpenCV (de-facto 3.3.1 via ROS Kinetic, docker/Dockerfile:1), which is not
// vendored and not installed here.  What is restated, from the published OpenCV algorithms (details and
// every choice among OpenCV's own build variants are recorded in oracle/ASSUMPTIONS.md):
//   cv::createCLAHE(3.0, Size(8,8))->apply ......... call site feature_tracker.cpp:87-93   [clahe.cpp]
//   cv::calcOpticalFlowPyrLK(.., Size(21,21), 3) ... call site feature_tracker.cpp:113     [lkpyramid.cpp,
//        pyramids.cpp: buildOpticalFlowPyramid / pyrDown, calcSharrDeriv, LKTrackerInvoker]
//   cv::goodFeaturesToTrack(.., 0.01, MIN_DIST, mask) call site feature_tracker.cpp:149    [featureselect.cpp,
//        corner.cpp: cornerMinEigenVal = Sobel + cov + boxFilter + calcMinEigenVal]
// Compile with -ffp-contract=off: the float expressions below are evaluated exactly as written.
#include <atomic>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace {
inline int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
inline int cv_round(float v) { return (int)std::lrintf(v); }                 // round half to even
inline int cv_floor(float v) { return (int)std::floor(v); }
inline int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }
inline uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// ---- pyrDown (pyramids.cpp): separable [1 4 6 4 1], BORDER_REFLECT_101, dst = ((w+1)/2, (h+1)/2), (sum+128)>>8
void pyr_down(const uint8_t* s, int w, int h, uint8_t* d) {
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    std::vector<int> rowbuf((size_t)h * dw);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < dw; ++x) {
            const uint8_t* r = s + (size_t)y * w;
            rowbuf[(size_t)y * dw + x] = r[reflect101(2 * x - 2, w)] + 4 * r[reflect101(2 * x - 1, w)] + 6 * r[reflect101(2 * x, w)]
                                         + 4 * r[reflect101(2 * x + 1, w)] + r[reflect101(2 * x + 2, w)];
        }
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            auto R = [&](int yy) { return rowbuf[(size_t)reflect101(yy, h) * dw + x]; };
            d[(size_t)y * dw + x] = (uint8_t)((R(2 * y - 2) + 4 * R(2 * y - 1) + 6 * R(2 * y) + 4 * R(2 * y + 1) + R(2 * y + 2) + 128) >> 8);
        }
}

struct Level { int w, h; std::vector<uint8_t> img; std::vector<int16_t> deriv; };   // deriv interleaved (Ix,Iy)

// ---- calcSharrDeriv (lkpyramid.cpp): un-normalised Scharr, reflect-101 at the image edge
void scharr(Level& L) {
    const int w = L.w, h = L.h;
    L.deriv.assign((size_t)w * h * 2, 0);
    auto I = [&](int y, int x) { return (int)L.img[(size_t)reflect101(y, h) * w + reflect101(x, w)]; };
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            auto t0 = [&](int xx) { return 3 * (I(y - 1, xx) + I(y + 1, xx)) + 10 * I(y, xx); };
            auto t1 = [&](int xx) { return I(y + 1, xx) - I(y - 1, xx); };
            // x+-1 reflect exactly like the row buffers of calcSharrDeriv: t[-1] = t[1], t[w] = t[w-2]
            const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            L.deriv[((size_t)y * w + x) * 2] = (int16_t)(t0(xp) - t0(xm));
            L.deriv[((size_t)y * w + x) * 2 + 1] = (int16_t)(3 * (t1(xm) + t1(xp)) + 10 * t1(x));
        }
}

void build_pyramid(const uint8_t* img, int w, int h, int max_level, std::vector<Level>& pyr, bool derivs) {
    pyr.resize(max_level + 1);
    pyr[0].w = w; pyr[0].h = h; pyr[0].img.assign(img, img + (size_t)w * h);
    for (int l = 1; l <= max_level; ++l) {
        pyr[l].w = (pyr[l - 1].w + 1) / 2; pyr[l].h = (pyr[l - 1].h + 1) / 2;
        pyr[l].img.resize((size_t)pyr[l].w * pyr[l].h);
        pyr_down(pyr[l - 1].img.data(), pyr[l - 1].w, pyr[l - 1].h, pyr[l].img.data());
    }
    if (derivs) for (auto& L : pyr) scharr(L);
}

// intensity with the REFLECT_101 pyramid border; derivative with the CONSTANT(0) border
inline int pix(const Level& L, int x, int y) { return L.img[(size_t)reflect101(y, L.h) * L.w + reflect101(x, L.w)]; }
inline int der(const Level& L, int x, int y, int c) { return (x < 0 || y < 0 || x >= L.w || y >= L.h) ? 0 : L.deriv[((size_t)y * L.w + x) * 2 + c]; }

const int WIN = 21, W_BITS = 14;
const float FLT_SCALE = 1.f / (1 << 20);

struct Weights { int w00, w01, w10, w11; };
inline Weights bilinear_weights(float a, float b) {
    Weights q;
    q.w00 = cv_round((1.f - a) * (1.f - b) * (1 << W_BITS));
    q.w01 = cv_round(a * (1.f - b) * (1 << W_BITS));
    q.w10 = cv_round((1.f - a) * b * (1 << W_BITS));
    q.w11 = (1 << W_BITS) - q.w00 - q.w01 - q.w10;
    return q;
}

static std::atomic<long long> g_lk_iterations{0};
extern "C" long long oracle_fe_lk_iterations(int reset) { const long long v = g_lk_iterations; if (reset) g_lk_iterations = 0; return v; }

// LKTrackerInvoker for one point on one level.  A / b sums are exact 64-bit integers converted once to float
// (oracle/ASSUMPTIONS.md F3: the scalar and SSE builds of OpenCV already differ from each other here).
void lk_point_level(const Level& I, const Level& J, int level, int max_level, float px, float py, float& nx, float& ny,
                    uint8_t& status, float& err, int max_count, double eps2, float min_eig_thr) {
    const float half = (WIN - 1) * 0.5f;
    float prevx = px * (float)(1. / (1 << level)), prevy = py * (float)(1. / (1 << level));
    float nextx, nexty;
    if (level == max_level) { nextx = prevx; nexty = prevy; }
    else { nextx = nx * 2.f; nexty = ny * 2.f; }
    nx = nextx; ny = nexty;
    prevx -= half; prevy -= half;
    const int ipx = cv_floor(prevx), ipy = cv_floor(prevy);
    if (ipx < -WIN || ipx >= I.w || ipy < -WIN || ipy >= I.h) {
        if (level == 0) { status = 0; err = 0; }
        return;
    }
    Weights q = bilinear_weights(prevx - ipx, prevy - ipy);
    int16_t Ibuf[WIN * WIN], dI[WIN * WIN * 2];
    int64_t iA11 = 0, iA12 = 0, iA22 = 0;
    for (int y = 0; y < WIN; ++y)
        for (int x = 0; x < WIN; ++x) {
            const int X = ipx + x, Y = ipy + y;
            const int ival = descale(pix(I, X, Y) * q.w00 + pix(I, X + 1, Y) * q.w01 + pix(I, X, Y + 1) * q.w10 + pix(I, X + 1, Y + 1) * q.w11, W_BITS - 5);
            const int ixval = descale(der(I, X, Y, 0) * q.w00 + der(I, X + 1, Y, 0) * q.w01 + der(I, X, Y + 1, 0) * q.w10 + der(I, X + 1, Y + 1, 0) * q.w11, W_BITS);
            const int iyval = descale(der(I, X, Y, 1) * q.w00 + der(I, X + 1, Y, 1) * q.w01 + der(I, X, Y + 1, 1) * q.w10 + der(I, X + 1, Y + 1, 1) * q.w11, W_BITS);
            Ibuf[y * WIN + x] = (int16_t)ival; dI[(y * WIN + x) * 2] = (int16_t)ixval; dI[(y * WIN + x) * 2 + 1] = (int16_t)iyval;
            iA11 += (int64_t)ixval * ixval; iA12 += (int64_t)ixval * iyval; iA22 += (int64_t)iyval * iyval;
        }
    const float A11 = (float)iA11 * FLT_SCALE, A12 = (float)iA12 * FLT_SCALE, A22 = (float)iA22 * FLT_SCALE;
    float D = A11 * A22 - A12 * A12;
    const float minEig = (A22 + A11 - std::sqrt((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * WIN * WIN);
    if (minEig < min_eig_thr || D < 1.1920929e-07f) {
        if (level == 0) status = 0;
        return;
    }
    D = 1.f / D;
    nextx -= half; nexty -= half;
    float pdx = 0, pdy = 0;
    for (int j = 0; j < max_count; ++j) {
        const int inx = cv_floor(nextx), iny = cv_floor(nexty);
        if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) {
            if (level == 0) status = 0;
            break;
        }
        ++g_lk_iterations;                     // (statistics for DESIGN.md: how many iterations a track costs; not part of the result)
        Weights r = bilinear_weights(nextx - inx, nexty - iny);
        int64_t ib1 = 0, ib2 = 0;
        for (int y = 0; y < WIN; ++y)
            for (int x = 0; x < WIN; ++x) {
                const int X = inx + x, Y = iny + y;
                const int diff = descale(pix(J, X, Y) * r.w00 + pix(J, X + 1, Y) * r.w01 + pix(J, X, Y + 1) * r.w10 + pix(J, X + 1, Y + 1) * r.w11, W_BITS - 5) - Ibuf[y * WIN + x];
                ib1 += (int64_t)diff * dI[(y * WIN + x) * 2]; ib2 += (int64_t)diff * dI[(y * WIN + x) * 2 + 1];
            }
        const float b1 = (float)ib1 * FLT_SCALE, b2 = (float)ib2 * FLT_SCALE;
        const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
        nextx += dx; nexty += dy;
        nx = nextx + half; ny = nexty + half;
        if ((double)dx * dx + (double)dy * dy <= eps2) break;
        if (j > 0 && std::abs(dx + pdx) < 0.01 && std::abs(dy + pdy) < 0.01) {
            nx -= dx * 0.5f; ny -= dy * 0.5f;
            break;
        }
        pdx = dx; pdy = dy;
    }
    if (status && level == 0) {
        const float ex = nx - half, ey = ny - half;
        const int inx = cv_floor(ex), iny = cv_floor(ey);
        if (inx < -WIN || inx >= J.w || iny < -WIN || iny >= J.h) { status = 0; return; }
        Weights r = bilinear_weights(ex - inx, ey - iny);
        int64_t e = 0;
        for (int y = 0; y < WIN; ++y)
            for (int x = 0; x < WIN; ++x) {
                const int X = inx + x, Y = iny + y;
                const int diff = descale(pix(J, X, Y) * r.w00 + pix(J, X + 1, Y) * r.w01 + pix(J, X, Y + 1) * r.w10 + pix(J, X + 1, Y + 1) * r.w11, W_BITS - 5) - Ibuf[y * WIN + x];
                e += std::abs(diff);
            }
        // lkpyramid.cpp: `err[ptidx] = errval * 1.f/(32*winSize.width*winSize.height)` parses as a DIVISION (found by the
        // second restatement oracle/fe_numpy.py, which follows SURVEY Appendix B.3; the first version multiplied by 1.f/14112)
        err = ((float)e * 1.f) / (float)(32 * WIN * WIN);
    }
}

// ---- cornerMinEigenVal(blockSize 3, ksize 3) (corner.cpp)
void min_eig_map(const uint8_t* img, int w, int h, float* eig) {
    const double scale_d = 1.0 / ((double)(1 << 2) * 3 * 255.0);
    const float k1 = (float)(1.0 * scale_d), k2 = (float)(2.0 * scale_d);      // Sobel smoothing taps * scale, as float kernels
    std::vector<float> dx((size_t)w * h), dy((size_t)w * h), rbuf((size_t)w * h);
    auto P = [&](int y, int x) { return (int)img[(size_t)reflect101(y, h) * w + reflect101(x, w)]; };
    // Dx: rows [-1 0 1] (exact), columns [s 2s s]:  f0*r1 + f1*(r0 + r2)
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) rbuf[(size_t)y * w + x] = (float)(P(y, x + 1) - P(y, x - 1));
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        const float r0 = rbuf[(size_t)reflect101(y - 1, h) * w + x], r1 = rbuf[(size_t)y * w + x], r2 = rbuf[(size_t)reflect101(y + 1, h) * w + x];
        dx[(size_t)y * w + x] = k2 * r1 + k1 * (r0 + r2);
    }
    // Dy: rows [s 2s s] on uchar: k0*S[x] + k1*(S[x-1] + S[x+1]) (integer pair sum), columns [-1 0 1]
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) rbuf[(size_t)y * w + x] = (float)P(y, x) * k2 + (float)(P(y, x - 1) + P(y, x + 1)) * k1;
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x)
        dy[(size_t)y * w + x] = rbuf[(size_t)reflect101(y + 1, h) * w + x] - rbuf[(size_t)reflect101(y - 1, h) * w + x];
    // cov = (dx*dx, dx*dy, dy*dy) as float; 3x3 un-normalised box in double; min eigenvalue in float.
    // The box is SEPARABLE, as cv::boxFilter runs it (RowSum<float, double> then ColumnSum<double, float>, ksize 3, normalize = false,
    // BORDER_REFLECT_101): a row sum ((p[x-1] + p[x]) + p[x+1]) in double per channel, then the sum of three row sums ((r[y-1] + r[y]) +
    // r[y+1]) -- round 5 (ASSUMPTIONS F10): the first version added the nine values in one sequential double sum,
    // which is neither OpenCV's order nor cheap (27 conversions + 27 FP64 adds per pixel instead of 3 + 12).  OpenCV's row / column
    // filters slide (s += p[x+2] - p[x-1]): the same sums in exact arithmetic, not reproduced here (a running sum over 752 columns
    // is a serial chain; the double rounding it would add is far below the float the result is cast to).
    std::vector<double> rs((size_t)w * h * 3);
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        double sxx = 0, sxy = 0, syy = 0;
        for (int u = -1; u <= 1; ++u) {
            const size_t o = (size_t)y * w + reflect101(x + u, w);
            const float gx = dx[o], gy = dy[o];
            sxx += (double)(gx * gx); sxy += (double)(gx * gy); syy += (double)(gy * gy);
        }
        double* r = &rs[((size_t)y * w + x) * 3];
        r[0] = sxx; r[1] = sxy; r[2] = syy;
    }
    for (int y = 0; y < h; ++y) for (int x = 0; x < w; ++x) {
        double sxx = 0, sxy = 0, syy = 0;
        for (int v = -1; v <= 1; ++v) {
            const double* r = &rs[((size_t)reflect101(y + v, h) * w + x) * 3];
            sxx += r[0]; sxy += r[1]; syy += r[2];
        }
        const float a = (float)sxx * 0.5f, b = (float)sxy, c = (float)syy * 0.5f;
        eig[(size_t)y * w + x] = (float)((a + c) - std::sqrt((a - c) * (a - c) + b * b));
    }
}
}  // namespace

extern "C" {

void oracle_fe_pyrdown(const uint8_t* src, int w, int h, uint8_t* dst) { pyr_down(src, w, h, dst); }

// derivative image of one level (int16 interleaved Ix,Iy)
void oracle_fe_scharr(const uint8_t* img, int w, int h, int16_t* out) {
    Level L; L.w = w; L.h = h; L.img.assign(img, img + (size_t)w * h);
    scharr(L);
    memcpy(out, L.deriv.data(), sizeof(int16_t) * L.deriv.size());
}

// cv::calcOpticalFlowPyrLK(prev, next, prevPts, nextPts, status, err, Size(21,21), max_level) with the default
// TermCriteria(COUNT+EPS, 30, 0.01), flags = 0, minEigThreshold = 1e-4.  Points are (x, y) float pairs.
void oracle_fe_lk(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_xy, int n, int max_level,
                  float* next_xy, uint8_t* status, float* err) {
    std::vector<Level> pI, pJ;
    // buildOpticalFlowPyramid stops before a level that would not be larger than the window in both dimensions
    // (lkpyramid.cpp: `if (sz.width <= winSize.width || sz.height <= winSize.height) return level - 1` [3P], F1)
    {
        int lw = w, lh = h, lv = 0;
        while (lv < max_level) {
            const int nw = (lw + 1) / 2, nh = (lh + 1) / 2;
            if (nw <= 21 || nh <= 21) break;
            lw = nw; lh = nh; ++lv;
        }
        max_level = lv;
    }
    build_pyramid(prev, w, h, max_level, pI, true);
    build_pyramid(next, w, h, max_level, pJ, false);
    for (int i = 0; i < n; ++i) { status[i] = 1; err[i] = 0; next_xy[2 * i] = 0; next_xy[2 * i + 1] = 0; }
    for (int level = max_level; level >= 0; --level)
        for (int i = 0; i < n; ++i)
            lk_point_level(pI[level], pJ[level], level, max_level, prev_xy[2 * i], prev_xy[2 * i + 1], next_xy[2 * i], next_xy[2 * i + 1],
                           status[i], err[i], 30, 0.01 * 0.01, 1e-4f);
}

// the same with the points of every level spread over `nthreads` host threads (OpenCV runs LKTrackerInvoker under
// parallel_for_ over the points; the pyramids are built once).  Results are identical to oracle_fe_lk.
void oracle_fe_lk_mt(const uint8_t* prev, const uint8_t* next, int w, int h, const float* prev_xy, int n, int max_level,
                     float* next_xy, uint8_t* status, float* err, int nthreads) {
    std::vector<Level> pI, pJ;
    {
        int lw = w, lh = h, lv = 0;
        while (lv < max_level) {
            const int nw = (lw + 1) / 2, nh = (lh + 1) / 2;
            if (nw <= 21 || nh <= 21) break;
            lw = nw; lh = nh; ++lv;
        }
        max_level = lv;
    }
    build_pyramid(prev, w, h, max_level, pI, true);
    build_pyramid(next, w, h, max_level, pJ, false);
    for (int i = 0; i < n; ++i) { status[i] = 1; err[i] = 0; next_xy[2 * i] = 0; next_xy[2 * i + 1] = 0; }
    nthreads = std::max(1, std::min(nthreads, n));
    for (int level = max_level; level >= 0; --level) {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t)
            pool.emplace_back([&, t]() {
                for (int i = t; i < n; i += nthreads)
                    lk_point_level(pI[level], pJ[level], level, max_level, prev_xy[2 * i], prev_xy[2 * i + 1], next_xy[2 * i],
                                   next_xy[2 * i + 1], status[i], err[i], 30, 0.01 * 0.01, 1e-4f);
            });
        for (auto& th : pool) th.join();
    }
}

void oracle_fe_mineig(const uint8_t* img, int w, int h, float* eig) { min_eig_map(img, w, h, eig); }

// cv::goodFeaturesToTrack(img, corners, max_corners, quality, min_dist, mask, 3, false).  Returns count.
int oracle_fe_gftt(const uint8_t* img, int w, int h, const uint8_t* mask, int max_corners, double quality, double min_dist,
                   float* out_xy) {
    std::vector<float> eig((size_t)w * h);
    min_eig_map(img, w, h, eig.data());
    double maxVal = 0;
    bool any = false;
    for (size_t k = 0; k < eig.size(); ++k) if (!mask || mask[k]) { if (!any || eig[k] > maxVal) maxVal = eig[k]; any = true; }
    const float thr = (float)(maxVal * quality);
    for (auto& v : eig) v = v > thr ? v : 0.f;
    std::vector<int> cand;
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            const float val = eig[(size_t)y * w + x];
            if (val == 0 || (mask && !mask[(size_t)y * w + x])) continue;
            float mx = val;
            for (int v = -1; v <= 1; ++v) for (int u = -1; u <= 1; ++u) mx = std::max(mx, eig[(size_t)(y + v) * w + x + u]);
            if (val == mx) cand.push_back(y * w + x);
        }
    std::sort(cand.begin(), cand.end(), [&](int a, int b) { return eig[a] > eig[b] ? true : (eig[a] < eig[b] ? false : a > b); });
    int n = 0;
    if (min_dist >= 1) {
        const int cell = (int)std::lrint(min_dist);
        const int gw = (w + cell - 1) / cell, gh = (h + cell - 1) / cell;
        std::vector<std::vector<int>> grid((size_t)gw * gh);
        const double md2 = min_dist * min_dist;
        for (int idx : cand) {
            const int y = idx / w, x = idx % w;
            const int xc = x / cell, yc = y / cell;
            const int x1 = std::max(0, xc - 1), y1 = std::max(0, yc - 1), x2 = std::min(gw - 1, xc + 1), y2 = std::min(gh - 1, yc + 1);
            bool good = true;
            for (int yy = y1; yy <= y2 && good; ++yy)
                for (int xx = x1; xx <= x2 && good; ++xx)
                    for (int o : grid[(size_t)yy * gw + xx]) {
                        const float dx = (float)(x - o % w), dy = (float)(y - o / w);
                        if (dx * dx + dy * dy < md2) { good = false; break; }
                    }
            if (good) {
                grid[(size_t)yc * gw + xc].push_back(idx);
                out_xy[2 * n] = (float)x; out_xy[2 * n + 1] = (float)y;
                if (++n == max_corners && max_corners > 0) break;
            }
        }
    } else {
        for (int idx : cand) {
            out_xy[2 * n] = (float)(idx % w); out_xy[2 * n + 1] = (float)(idx / w);
            if (++n == max_corners && max_corners > 0) break;
        }
    }
    return n;
}

// cv::createCLAHE(clip, Size(8,8))->apply on an image whose size divides by 8 (752x480: tiles of 94x60)
int oracle_fe_clahe(const uint8_t* src, int w, int h, double clip, uint8_t* dst) {
    const int TX = 8, TY = 8;
    if (w % TX || h % TY) return -1;
    const int tw = w / TX, th = h / TY, area = tw * th;
    const float lutScale = 255.f / area;
    int clipLimit = (int)(clip * area / 256);
    clipLimit = std::max(clipLimit, 1);
    std::vector<uint8_t> lut((size_t)TX * TY * 256);
    for (int ty = 0; ty < TY; ++ty)
        for (int tx = 0; tx < TX; ++tx) {
            int hist[256] = {0};
            for (int y = 0; y < th; ++y) for (int x = 0; x < tw; ++x) hist[src[(size_t)(ty * th + y) * w + tx * tw + x]]++;
            int clipped = 0;
            for (int i = 0; i < 256; ++i) if (hist[i] > clipLimit) { clipped += hist[i] - clipLimit; hist[i] = clipLimit; }
            const int batch = clipped / 256;
            int residual = clipped - batch * 256;
            for (int i = 0; i < 256; ++i) hist[i] += batch;
            if (residual != 0) {
                const int step = std::max(256 / residual, 1);
                for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
            }
            int sum = 0;
            for (int i = 0; i < 256; ++i) { sum += hist[i]; lut[((size_t)ty * TX + tx) * 256 + i] = sat_u8(cv_round(sum * lutScale)); }
        }
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    for (int y = 0; y < h; ++y) {
        const float tyf = y * inv_th - 0.5f;
        int ty1 = cv_floor(tyf), ty2 = ty1 + 1;
        const float ya = tyf - ty1, ya1 = 1.0f - ya;
        ty1 = std::max(ty1, 0); ty2 = std::min(ty2, TY - 1);
        for (int x = 0; x < w; ++x) {
            const float txf = x * inv_tw - 0.5f;
            int tx1 = cv_floor(txf), tx2 = tx1 + 1;
            const float xa = txf - tx1, xa1 = 1.0f - xa;
            tx1 = std::max(tx1, 0); tx2 = std::min(tx2, TX - 1);
            const int v = src[(size_t)y * w + x];
            const float res = (lut[((size_t)ty1 * TX + tx1) * 256 + v] * xa1 + lut[((size_t)ty1 * TX + tx2) * 256 + v] * xa) * ya1
                              + (lut[((size_t)ty2 * TX + tx1) * 256 + v] * xa1 + lut[((size_t)ty2 * TX + tx2) * 256 + v] * xa) * ya;
            dst[(size_t)y * w + x] = sat_u8(cv_round(res));
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// FeatureTracker::setMask (feature_tracker.cpp:36-69): start from the fisheye mask or all-255 (:38-41); visit the
// points by track_cnt descending (:48-51 — std::sort there, i.e. unstable; canonicalised to STABLE, ASSUMPTIONS F7);
// keep a point iff mask.at<uchar>(Point(pt)) == 255 (:59; Point2f -> Point rounds half to even) and blank
// cv::circle(mask, pt, radius, 0, -1) (:64) — restated as { (x, y) : dx^2 + dy^2 <= r^2 } clipped to the image [3P].
// Points whose rounded position is outside the image are skipped (the reference would read out of bounds).
// kept_index receives indices into the input in kept order; mask_out (w*h) the final mask.  Returns the kept count.
int oracle_fe_setmask(const float* pts_xy, const int* track_cnt, int n, const uint8_t* base_mask, int w, int h, int radius,
                      int* kept_index, uint8_t* mask_out) {
    std::vector<uint8_t> mask((size_t)w * h, 255);
    if (base_mask) std::memcpy(mask.data(), base_mask, mask.size());
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return track_cnt[a] > track_cnt[b]; });
    int nk = 0;
    for (int i : order) {
        const int px = (int)std::nearbyintf(pts_xy[2 * i]), py = (int)std::nearbyintf(pts_xy[2 * i + 1]);
        if (px < 0 || py < 0 || px >= w || py >= h) continue;
        if (mask[(size_t)py * w + px] != 255) continue;
        kept_index[nk++] = i;
        for (int y = std::max(0, py - radius); y <= std::min(h - 1, py + radius); ++y)
            for (int x = std::max(0, px - radius); x <= std::min(w - 1, px + radius); ++x)
                if ((x - px) * (x - px) + (y - py) * (y - py) <= radius * radius) mask[(size_t)y * w + x] = 0;
    }
    if (mask_out) std::memcpy(mask_out, mask.data(), mask.size());
    return nk;
}

// PinholeCamera::liftProjective (camera_model/src/camera_models/PinholeCamera.cc:450-510) with the recursive
// distortion model, n = 8 (:484-493), distortion() of :646-661; result (x/z, y/z) as float like cv::Point2f
// (feature_tracker.cpp:262-267).  intr = fx fy cx cy k1 k2 p1 p2.
void oracle_fe_lift(const float* pts_xy, int n, const double* intr, float* out_xy) {
    const double fx = intr[0], fy = intr[1], cx = intr[2], cy = intr[3], k1 = intr[4], k2 = intr[5], p1 = intr[6], p2 = intr[7];
    const double m_inv_K11 = 1.0 / fx, m_inv_K13 = -cx / fx, m_inv_K22 = 1.0 / fy, m_inv_K23 = -cy / fy;
    for (int i = 0; i < n; ++i) {
        const double mx_d = m_inv_K11 * (double)pts_xy[2 * i] + m_inv_K13, my_d = m_inv_K22 * (double)pts_xy[2 * i + 1] + m_inv_K23;
        double mx_u = mx_d, my_u = my_d;
        for (int it = 0; it < 8; ++it) {
            const double mx2 = mx_u * mx_u, my2 = my_u * my_u, mxy = mx_u * my_u, rho2 = mx2 + my2;
            const double rad = k1 * rho2 + k2 * rho2 * rho2;
            const double dx = mx_u * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2);
            const double dy = my_u * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2);
            mx_u = mx_d - dx; my_u = my_d - dy;
        }
        out_xy[2 * i] = (float)mx_u; out_xy[2 * i + 1] = (float)my_u;
    }
}

// ---- FeatureTracker::rejectWithF (feature_tracker.cpp:169-202): cv::findFundamentalMat(un_cur, un_forw, FM_RANSAC, thr, 0.99,
// status), restated after OpenCV 3.3's fundam.cpp / ptsetreg.cpp AS RECALLED (ASSUMPTIONS.md F9 lists what is matched and
// what cannot be):
//   * n >= 15: RANSACPointSetRegistrator(modelPoints 7, maxIters 1000): cv::RNG((uint64)-1) (multiply-with-carry, coefficient
//     4164903690), getSubset (7 distinct rng.uniform(0, n) draws, redrawn while the last point of either sample is collinear
//     with two earlier ones), run7Point (null space of the 7 x 9 design matrix of the RAW points, cubic det(x G + H) = 0 by
//     cv::solveCubic, up to three models, each scaled to F33 = 1), the error max(d1^2/|l1|^2, d2^2/|l2|^2) as float against
//     thr^2, a model replaces the best one if it has MORE inliers than max(best, 6), and after every improvement
//     niters = RANSACUpdateNumIters(0.99, outlier ratio, 7, niters);
//   * 8 <= n < 15: LMeDSPointSetRegistrator: niters = max(RANSACUpdateNumIters(0.99, 0.45, 7, 1000), 3) samples, the model with
//     the smallest median error, inliers = error <= (2.5 * 1.4826 * (1 + 5 / (n - 7)) * sqrt(median))^2, at least 0.001^2.
// The models of ONE sample are visited best first with a canonical tie order (the order OpenCV visits them in depends on the
// basis its SVD returns for the two-dimensional null space, which no restatement can reproduce).
// Returns the inlier count (n and an all-ones status if no model was found: documented choice).
static float fr_error(const double* f, double x1, double y1, double x2, double y2) {
    double a = f[0] * x1 + f[1] * y1 + f[2], b = f[3] * x1 + f[4] * y1 + f[5], c = f[6] * x1 + f[7] * y1 + f[8];
    const double s2 = 1.0 / (a * a + b * b), d2 = x2 * a + y2 * b + c;
    a = f[0] * x2 + f[3] * y2 + f[6]; b = f[1] * x2 + f[4] * y2 + f[7]; c = f[2] * x2 + f[5] * y2 + f[8];
    const double s1 = 1.0 / (a * a + b * b), d1 = x1 * a + y1 * b + c;
    const double e1 = d1 * d1 * s1, e2 = d2 * d2 * s2;
    return (float)(e1 > e2 ? e1 : e2);
}
struct CvRng {                                        // cv::RNG
    unsigned long long state;
    explicit CvRng(unsigned long long s) : state(s ? s : 0xffffffffull) {}
    unsigned next() { state = (unsigned long long)(unsigned)state * 4164903690ull + (unsigned)(state >> 32); return (unsigned)state; }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};
static bool fr_collinear_last(const float* p, const int* idx, int count) {      // haveCollinearPoints: last point against pairs
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
static bool fr_get_subset(CvRng& rng, const float* p1, const float* p2, int n, int* idx, int max_attempts) {
    int iters = 0, i = 0;
    for (; iters < max_attempts; ++iters) {
        for (i = 0; i < 7 && iters < max_attempts;) {
            int c;
            for (;;) {
                c = idx[i] = rng.uniform(0, n);
                int j = 0;
                for (; j < i; ++j) if (c == idx[j]) break;
                if (j == i) break;
            }
            ++i;
        }
        if (i == 7 && (fr_collinear_last(p1, idx, 7) || fr_collinear_last(p2, idx, 7))) continue;
        break;
    }
    return i == 7 && iters < max_attempts;
}
static int fr_solve_cubic(const double* cf, double* r) {          // cv::solveCubic(c0 x^3 + c1 x^2 + c2 x + c3)
    double a0 = cf[0], a1 = cf[1], a2 = cf[2], a3 = cf[3];
    if (a0 == 0) {
        if (a1 == 0) {
            if (a2 == 0) return a3 == 0 ? -1 : 0;
            r[0] = -a3 / a2;
            return 1;
        }
        double d = a2 * a2 - 4 * a1 * a3;
        if (d < 0) return 0;
        d = std::sqrt(d);
        const double q1 = (-a2 + d) * 0.5, q2 = (a2 + d) * -0.5;
        if (std::fabs(q1) > std::fabs(q2)) { r[0] = q1 / a1; r[1] = a3 / q1; } else { r[0] = q2 / a1; r[1] = a3 / q2; }
        return d > 0 ? 2 : 1;
    }
    a0 = 1.0 / a0; a1 *= a0; a2 *= a0; a3 *= a0;
    const double Q = (a1 * a1 - 3 * a2) * (1.0 / 9), R = (2 * a1 * a1 * a1 - 9 * a1 * a2 + 27 * a3) * (1.0 / 54);
    const double Qcubed = Q * Q * Q;
    double d = Qcubed - R * R;
    if (d >= 0) {
        const double theta = std::acos(R / std::sqrt(Qcubed)), sqrtQ = std::sqrt(Q);
        const double t0 = -2 * sqrtQ, t1 = theta * (1.0 / 3), t2 = a1 * (1.0 / 3);
        r[0] = t0 * std::cos(t1) - t2;
        r[1] = t0 * std::cos(t1 + (2.0 * 3.1415926535897932384626433832795 / 3)) - t2;
        r[2] = t0 * std::cos(t1 + (4.0 * 3.1415926535897932384626433832795 / 3)) - t2;
        return 3;
    }
    d = std::sqrt(-d);
    double e = std::pow(d + std::fabs(R), 0.333333333333);
    if (R > 0) e = -e;
    r[0] = (e + Q / e) - a1 * (1.0 / 3);
    return 1;
}
// run7Point on the sampled correspondences: up to three 3x3 models (row-major), count returned
static int fr_seven_point(const float* p1, const float* p2, const int* idx, double* models) {
    double AtA[81];
    for (int e = 0; e < 81; ++e) AtA[e] = 0.0;
    for (int i = 0; i < 7; ++i) {
        const double x0 = p1[2 * idx[i]], y0 = p1[2 * idx[i] + 1], x1 = p2[2 * idx[i]], y1 = p2[2 * idx[i] + 1];
        const double row[9] = {x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1.0};
        for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) AtA[a * 9 + b] += row[a] * row[b];
    }
    // null space = eigenvectors of the two smallest eigenvalues of A^T A (two-sided cyclic Jacobi)
    double V[81];
    for (int e = 0; e < 81; ++e) V[e] = (e % 10 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int a = 0; a < 9; ++a) for (int b = 0; b < 9; ++b) (a == b ? dia : off) += AtA[a * 9 + b] * AtA[a * 9 + b];
        if (!(off > 1e-60 * dia)) break;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                const double apq = AtA[p * 9 + q];
                if (apq == 0.0) continue;
                const double th = (AtA[q * 9 + q] - AtA[p * 9 + p]) / (2.0 * apq);
                const double tn = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(1.0 + th * th));
                const double cs = 1.0 / std::sqrt(1.0 + tn * tn), sn = cs * tn;
                for (int r = 0; r < 9; ++r) { const double a = AtA[r * 9 + p], b = AtA[r * 9 + q]; AtA[r * 9 + p] = cs * a - sn * b; AtA[r * 9 + q] = sn * a + cs * b; }
                for (int r = 0; r < 9; ++r) { const double a = AtA[p * 9 + r], b = AtA[q * 9 + r]; AtA[p * 9 + r] = cs * a - sn * b; AtA[q * 9 + r] = sn * a + cs * b; }
                for (int r = 0; r < 9; ++r) { const double a = V[r * 9 + p], b = V[r * 9 + q]; V[r * 9 + p] = cs * a - sn * b; V[r * 9 + q] = sn * a + cs * b; }
            }
    }
    int i2 = 0, i1 = -1;                                   // i2: smallest eigenvalue, i1: second smallest
    for (int c = 1; c < 9; ++c) if (AtA[c * 10] < AtA[i2 * 10]) i2 = c;
    for (int c = 0; c < 9; ++c) if (c != i2 && (i1 < 0 || AtA[c * 10] < AtA[i1 * 10])) i1 = c;
    double f1[9], f2[9];
    for (int e = 0; e < 9; ++e) { f1[e] = V[e * 9 + i1]; f2[e] = V[e * 9 + i2]; }
    for (int e = 0; e < 9; ++e) f1[e] -= f2[e];
    double c[4];
    double t0 = f2[4] * f2[8] - f2[5] * f2[7], t1 = f2[3] * f2[8] - f2[5] * f2[6], t2 = f2[3] * f2[7] - f2[4] * f2[6];
    c[3] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2;
    c[2] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2 - f1[3] * (f2[1] * f2[8] - f2[2] * f2[7]) + f1[4] * (f2[0] * f2[8] - f2[2] * f2[6]) -
           f1[5] * (f2[0] * f2[7] - f2[1] * f2[6]) + f1[6] * (f2[1] * f2[5] - f2[2] * f2[4]) - f1[7] * (f2[0] * f2[5] - f2[2] * f2[3]) +
           f1[8] * (f2[0] * f2[4] - f2[1] * f2[3]);
    t0 = f1[4] * f1[8] - f1[5] * f1[7]; t1 = f1[3] * f1[8] - f1[5] * f1[6]; t2 = f1[3] * f1[7] - f1[4] * f1[6];
    c[0] = f1[0] * t0 - f1[1] * t1 + f1[2] * t2;
    c[1] = f2[0] * t0 - f2[1] * t1 + f2[2] * t2 - f2[3] * (f1[1] * f1[8] - f1[2] * f1[7]) + f2[4] * (f1[0] * f1[8] - f1[2] * f1[6]) -
           f2[5] * (f1[0] * f1[7] - f1[1] * f1[6]) + f2[6] * (f1[1] * f1[5] - f1[2] * f1[4]) - f2[7] * (f1[0] * f1[5] - f1[2] * f1[3]) +
           f2[8] * (f1[0] * f1[4] - f1[1] * f1[3]);
    double r[3];
    const int nr = fr_solve_cubic(c, r);
    if (nr < 1 || nr > 3) return 0;
    int nm = 0;
    for (int k = 0; k < nr; ++k) {
        double lambda = r[k], mu = 1.0;
        const double sc = f1[8] * r[k] + f2[8];
        double* F = models + 9 * nm;
        if (std::fabs(sc) > 2.220446049250313e-16) { mu = 1.0 / sc; lambda *= mu; F[8] = 1.0; } else F[8] = 0.0;
        for (int e = 0; e < 8; ++e) F[e] = f1[e] * lambda + f2[e] * mu;
        bool finite = true;
        for (int e = 0; e < 9; ++e) finite = finite && (F[e] == F[e]) && std::fabs(F[e]) < 1e300;
        if (finite) ++nm;
    }
    return nm;
}
// canonical comparison of two models of one sample (scale- and sign-free): by the entries of F / (its entry of largest magnitude)
static bool fr_model_before(const double* Fa, const double* Fb) {
    double ma = 0, mb = 0;
    for (int e = 0; e < 9; ++e) { if (std::fabs(Fa[e]) > std::fabs(ma)) ma = Fa[e]; if (std::fabs(Fb[e]) > std::fabs(mb)) mb = Fb[e]; }
    for (int e = 0; e < 9; ++e) {
        const double va = ma != 0 ? Fa[e] / ma : Fa[e], vb = mb != 0 ? Fb[e] / mb : Fb[e];
        if (std::fabs(va - vb) > 1e-6) return va < vb;
    }
    return false;
}
static int fr_update_iters(double p, double ep, int model_points, int max_iters) {      // RANSACUpdateNumIters
    p = std::max(p, 0.0); p = std::min(p, 1.0);
    ep = std::max(ep, 0.0); ep = std::min(ep, 1.0);
    double num = std::max(1.0 - p, 2.2250738585072014e-308);
    double denom = 1.0 - std::pow(1.0 - ep, model_points);
    if (denom < 2.2250738585072014e-308) return 0;
    num = std::log(num); denom = std::log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)std::nearbyint(num / denom);
}

int oracle_fe_reject_with_f(const float* p1, const float* p2, int n, double threshold, uint8_t* status, double* F_out) {
    double bestF[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    bool have = false;
    int ninl = 0;
    std::vector<float> err(n), srt(n);
    CvRng rng((unsigned long long)-1);
    if (n >= 15) {
        const float t2 = (float)(threshold * threshold);
        int niters = 1000, max_good = 0;
        for (int iter = 0; iter < niters; ++iter) {
            int idx[7];
            if (!fr_get_subset(rng, p1, p2, n, idx, 10000)) break;
            double models[27];
            const int nm = fr_seven_point(p1, p2, idx, models);
            int bm = -1, bgood = -1;
            for (int m = 0; m < nm; ++m) {
                int good = 0;
                for (int i = 0; i < n; ++i) good += fr_error(models + 9 * m, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= t2 ? 1 : 0;
                if (good > bgood || (good == bgood && fr_model_before(models + 9 * m, models + 9 * bm))) { bgood = good; bm = m; }
            }
            if (bm >= 0 && bgood > std::max(max_good, 6)) {
                memcpy(bestF, models + 9 * bm, sizeof(bestF));
                have = true;
                max_good = bgood;
                niters = fr_update_iters(0.99, (double)(n - bgood) / n, 7, niters);
            }
        }
        for (int i = 0; i < n; ++i) {
            status[i] = (!have || fr_error(bestF, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= t2) ? 1 : 0;
            ninl += status[i];
        }
    } else {
        int niters = std::max(fr_update_iters(0.99, 0.45, 7, 1000), 3);
        double min_median = 1.7976931348623157e308;
        for (int iter = 0; iter < niters; ++iter) {
            int idx[7];
            if (!fr_get_subset(rng, p1, p2, n, idx, 1000)) break;      // (LMeDS: getSubset's default maxAttempts)
            double models[27];
            const int nm = fr_seven_point(p1, p2, idx, models);
            int bm = -1;
            double bmed = 0.0;
            for (int m = 0; m < nm; ++m) {
                for (int i = 0; i < n; ++i) srt[i] = fr_error(models + 9 * m, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
                std::sort(srt.begin(), srt.end());
                double med = n % 2 != 0 ? (double)srt[n / 2] : (double)(srt[n / 2 - 1] + srt[n / 2]) * 0.5;
                if (!(med == med)) continue;
                // (n <= 13: the median of a model that fits its 7 sample points exactly lies inside the fitted set and is rounding
                //  noise, 1e-20 .. 1e-30 px^2; such medians are snapped to zero so that the FIRST such sample wins instead of noise)
                if (med < 1e-12) med = 0.0;
                if (bm < 0 || med < bmed || (med == bmed && fr_model_before(models + 9 * m, models + 9 * bm))) { bmed = med; bm = m; }
            }
            if (bm >= 0 && bmed < min_median) { min_median = bmed; memcpy(bestF, models + 9 * bm, sizeof(bestF)); have = true; }
        }
        double sigma = 0.0;
        if (have) {
            sigma = 2.5 * 1.4826 * (1 + 5.0 / (n - 7)) * std::sqrt(min_median);
            sigma = std::max(sigma, 0.001);
        }
        const float t2 = (float)(sigma * sigma);
        for (int i = 0; i < n; ++i) {
            status[i] = (!have || fr_error(bestF, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]) <= t2) ? 1 : 0;
            ninl += status[i];
        }
        if (have && ninl < 7) {                       // LMedS reports failure below modelPoints inliers
            for (int i = 0; i < n; ++i) status[i] = 1;
            ninl = n; have = false;
            for (int e = 0; e < 9; ++e) bestF[e] = 0.0;
        }
    }
    if (F_out) memcpy(F_out, bestF, sizeof(bestF));
    return ninl;
}

}  // extern "C"
