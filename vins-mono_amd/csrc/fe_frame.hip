// fe_frame.hip — what FeatureTracker::readImage (feature_tracker/src/feature_tracker.cpp:81-167) does BETWEEN its OpenCV calls, on
// the device, so that vg_fe_read_image (fe_host.hip) runs a frame without handing intermediate results to the host:
//   :115-124  status[i] && inBorder(forw_pts[i]) + reduceVector        -> fe_ri_after_lk_kernel (ordered compaction = reduceVector)
//   :175-188  liftProjective of cur_pts / forw_pts for findFundamentalMat -> the same kernel (PinholeCamera, double, reference order)
//   :191-198  the registrator's sequential bookkeeping over the RANSAC iterations + reduceVector by its mask -> fe_ri_pick_kernel
//   :36-69    setMask's walk in the order the host's sort produced     -> fe_ri_setmask_kernel (+ fe_stamp_kernel of fe_kernels.hip)
//   :144      n_max_cnt = MAX_CNT - forw_pts.size()                    -> written by the same kernel where fe_select_kernel reads it
//   :71-79, :258-268  addPoints + undistortedPoints (liftProjective of the final list) -> fe_ri_finish_kernel
// All of them are single-workgroup kernels on <= a few hundred points: what matters is that they need no round trip, not their
// arithmetic.  Compiled with -ffp-contract=off: the lifting evaluates the reference's double expressions as written.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vg_target.h"
#include "fe_layout.h"

#define FDEV __device__ __forceinline__

// PinholeCamera::liftProjective (PinholeCamera.cc:450-510), recursive distortion model with n = 8: the same expressions as
// fe_lift_kernel (fe_kernels.hip)
FDEV void ri_lift(const RiDev& r, float px, float py, double& mx_u, double& my_u) {
    const double mx_d = (1.0 / r.fx) * (double)px + (-r.cx / r.fx), my_d = (1.0 / r.fy) * (double)py + (-r.cy / r.fy);
    mx_u = mx_d; my_u = my_d;
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const double mx2 = mx_u * mx_u, my2 = my_u * my_u, mxy = mx_u * my_u, rho2 = mx2 + my2;
        const double rad = r.k1 * rho2 + r.k2 * rho2 * rho2;
        const double dx = mx_u * rad + 2.0 * r.pp1 * mxy + r.pp2 * (rho2 + 2.0 * mx2);
        const double dy = my_u * rad + 2.0 * r.pp2 * mxy + r.pp1 * (rho2 + 2.0 * my2);
        mx_u = mx_d - dx; my_u = my_d - dy;
    }
}

// ordered compaction of a flag over [0, n) by one workgroup of 256 threads: dst[rank of i among the set flags] = src ? src[i] : i.
// Returns the number of set flags (uniform).  `wsum` = 4 ints of LDS.
FDEV int ri_compact(const uint8_t* flag, int n, const int* src, int* dst, int* wsum) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int base = 0;
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const bool f = i < n && flag[i] != 0;
        const unsigned long long bal = __ballot(f);
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int before = base;
        for (int q = 0; q < wave; ++q) before += wsum[q];
        if (f) dst[before + __popcll(bal & ((1ull << lane) - 1ull))] = src ? src[i] : i;
        base += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        __syncthreads();
    }
    return base;
}

// After the tracking kernel.  One workgroup.
extern "C" __global__ __launch_bounds__(256) void fe_ri_after_lk_kernel(FeDev d, RiDev r) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x;
    const int n = r.ctl[RI_N], publish = r.ctl[RI_PUBLISH];
    // :115-117  status[i] && inBorder(forw_pts[i])  (BORDER_SIZE 1, cvRound = round half to even)
    for (int i = tid; i < n; i += 256) {
        const float x = d.next_xy[2 * i], y = d.next_xy[2 * i + 1];
        const int ix = __float2int_rn(x), iy = __float2int_rn(y);
        const bool ok = d.status[i] != 0 && 1 <= ix && ix < d.W - 1 && 1 <= iy && iy < d.H - 1;
        r.a_status_lk[i] = ok ? 1 : 0;
        r.a_forw_xy[2 * i] = x; r.a_forw_xy[2 * i + 1] = y;
    }
    __threadfence_block();
    __syncthreads();
    const int n1 = ri_compact(r.a_status_lk, n, nullptr, r.idx1, wsum);
    __threadfence_block();
    __syncthreads();
    const bool ransac = publish && n1 >= 8;                         // rejectWithF: `if (forw_pts.size() >= 8)` (:171)
    for (int k = tid; k < n1; k += 256) {
        const int i = r.idx1[k];
        double ux, uy;
        ri_lift(r, d.next_xy[2 * i], d.next_xy[2 * i + 1], ux, uy);
        if (!publish) {                                             // undistortedPoints of the list this frame ends with (:262-267)
            r.a_un_xy[2 * k] = (float)ux; r.a_un_xy[2 * k + 1] = (float)uy;
        } else {
            if (ransac) {                                           // :176-187: FOCAL_LENGTH * x / z + COL / 2.0, rounded to float by cv::Point2f
                r.p2[2 * k] = (float)(r.focal * ux / 1.0 + r.half_w); r.p2[2 * k + 1] = (float)(r.focal * uy / 1.0 + r.half_h);
                double cx, cy;
                ri_lift(r, r.xy_in[2 * i], r.xy_in[2 * i + 1], cx, cy);
                r.p1[2 * k] = (float)(r.focal * cx / 1.0 + r.half_w); r.p1[2 * k + 1] = (float)(r.focal * cy / 1.0 + r.half_h);
            } else
                r.idx2[k] = i;
        }
    }
    if (tid == 0) {
        r.ctl[RI_N1] = n1;
        r.ctl[RI_RANSAC] = ransac ? 1 : 0;
        int fb = 0;
        if (ransac && n1 < 15) fb |= RI_FB_LMEDS;
        if (ransac && n1 > FE_RANSAC_MAXPTS) fb |= RI_FB_RANGE;
        r.ctl[RI_FALLBACK] = fb;
        r.ctl[RI_BEST] = -1;
        r.ctl[RI_NITERS] = 0;
        if (!ransac) r.ctl[RI_N2] = n1;
        if (!ransac || fb) {                                        // (fb: the host finishes rejectWithF; the header tells it so)
            r.a_hdr[RI_N] = n; r.a_hdr[RI_PUBLISH] = publish; r.a_hdr[RI_N1] = n1; r.a_hdr[RI_N2] = n1; r.a_hdr[RI_FALLBACK] = fb;
            r.a_hdr[RI_RANSAC] = ransac ? 1 : 0; r.a_hdr[RI_BEST] = -1; r.a_hdr[RI_NITERS] = 0;
        }
    }
}

// After fe_ransac7_kernel / fe_ransac_count_kernel: the registrator's loop over the iterations (ptsetreg.cpp RANSACPointSetRegistrator::run
// as restated in fe_ransac.hip: a model replaces the best one if it has more inliers than max(best, 6); after every improvement the
// iteration bound shrinks to RANSACUpdateNumIters(...), read from the host-made table; iterations at or beyond the bound do not
// count), then the mask of the winning model and reduceVector by it.  One workgroup; the loop itself runs on one wavefront with the
// counts of 64 iterations in a register each (v_readlane in sequence: the bound usually ends the loop inside the first chunk).
extern "C" __global__ __launch_bounds__(256) void fe_ri_pick_kernel(RiDev r) {
    __shared__ int wsum[4];
    __shared__ int sbest;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n1 = r.ctl[RI_N1];
    if (r.ctl[RI_PUBLISH] == 0 || r.ctl[RI_RANSAC] == 0) return;
    int fb = r.ctl[RI_FALLBACK];
    if (fb & (RI_FB_LMEDS | RI_FB_RANGE)) return;                   // (header already written by fe_ri_after_lk_kernel)
    if (tid < 64) {
        int niters = FE_RANSAC_MAXIT, max_good = 0, best = -1;
        const int* tab = r.niters_tab + (size_t)n1 * r.tab_stride;
        for (int base = 0; base < FE_RANSAC_MAXIT && base < niters; base += 64) {
            const int mine = base + lane < FE_RANSAC_MAXIT ? r.count[base + lane] : -1;
            for (int j = 0; j < 64 && base + j < niters && base + j < FE_RANSAC_MAXIT; ++j) {
                const int c = __shfl(mine, j);
                if (c > (max_good > 6 ? max_good : 6)) {
                    best = base + j; max_good = c;
                    const int t = tab[c];
                    niters = t < niters ? t : niters;
                }
            }
        }
        if (lane == 0) { sbest = best; r.ctl[RI_BEST] = best; r.ctl[RI_NITERS] = niters; }
    }
    __syncthreads();
    const int best = sbest;
    const int nw = (n1 + 63) >> 6;
    // the mask: inlier set of the winning iteration's model; no model at all -> nothing is rejected (fe_ransac.hip, ASSUMPTIONS F9)
    for (int k = tid; k < n1; k += 256)
        r.a_status_f[k] = best < 0 ? 1 : (uint8_t)((r.words[(size_t)best * nw + (k >> 6)] >> (k & 63)) & 1ull);
    __threadfence_block();
    __syncthreads();
    const int n2 = ri_compact(r.a_status_f, n1, r.idx1, r.idx2, wsum);
    if (tid == 0) {
        fb = r.ctl[RI_FALLBACK];
        r.ctl[RI_N2] = n2;
        r.a_hdr[RI_N] = r.ctl[RI_N]; r.a_hdr[RI_PUBLISH] = 1; r.a_hdr[RI_N1] = n1; r.a_hdr[RI_N2] = n2; r.a_hdr[RI_FALLBACK] = fb;
        r.a_hdr[RI_RANSAC] = 1; r.a_hdr[RI_BEST] = best; r.a_hdr[RI_NITERS] = r.ctl[RI_NITERS];
    }
}

// setMask (:36-69) in a given order: position q of the walk is survivor order[q] (order == nullptr: the list as it stands).  A point is
// kept iff its rounded position is inside the image, the base mask there is 255 and no previously kept point's filled disc covers it
// (fe_setmask_kernel of fe_kernels.hip has the derivation).  One wavefront, 64 positions of the walk at a time, one per lane:
//   1. every lane tests ITS point against the points kept in earlier chunks (their coordinates are read from LDS at a uniform address);
//   2. inside the chunk the walk is sequential over the lanes that are still alive (s_ff1 over the ballot): the first one is kept, its
//      coordinates go to all lanes through v_readlane, every later lane it covers drops out -- ten instructions per kept point instead of
//      an LDS round trip per candidate (48 us -> see DESIGN.md for 150 points);
//   3. the kept lanes store their results side by side (prefix popcount).
// Also: n_max_cnt = MAX_CNT - kept (:144) for the detection.
#define RI_SETMASK_MAX 2048
extern "C" __global__ __launch_bounds__(64) void fe_ri_setmask_kernel(FeDev d, RiDev r) {
    __shared__ short kx[RI_SETMASK_MAX], ky[RI_SETMASK_MAX];
    const int lane = threadIdx.x, W = d.W, H = d.H;
    const int n2 = r.ctl[RI_N2];
    const int r2 = r.radius * r.radius;
    int nk = 0;
    for (int base = 0; base < n2; base += 64) {
        const int q = base + lane;
        int px = 0, py = 0;
        bool alive = false;
        if (q < n2) {
            const int i = r.idx2[r.order ? r.order[q] : q];
            px = __float2int_rn(d.next_xy[2 * i]); py = __float2int_rn(d.next_xy[2 * i + 1]);      // Point2f -> Point: round half to even
            alive = px >= 0 && py >= 0 && px < W && py < H;
            if (alive && r.base_mask) alive = r.base_mask[(size_t)py * W + px] == 255;
        }
        for (int j = 0; j < nk; ++j) {
            const int dx = px - kx[j], dy = py - ky[j];
            alive = alive && !(dx * dx + dy * dy <= r2);
        }
        unsigned long long am = __ballot(alive), keptm = 0ull;
        while (am) {
            const int j = __ffsll((long long)am) - 1;
            keptm |= 1ull << j;
            const int jx = __shfl(px, j), jy = __shfl(py, j);
            const int dx = px - jx, dy = py - jy;
            alive = alive && lane > j && !(dx * dx + dy * dy <= r2);
            am = __ballot(alive);
        }
        if ((keptm >> lane) & 1ull) {
            const int k = nk + __popcll(keptm & ((1ull << lane) - 1ull));
            kx[k] = (short)px; ky[k] = (short)py;
            r.b_kept[k] = q;
            r.kept_xy[2 * k] = px; r.kept_xy[2 * k + 1] = py;
        }
        nk += __popcll(keptm);
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) {
        r.ctl[RI_NK] = nk;
        const int room = r.max_cnt - nk;
        const_cast<int*>(d.max_corners)[0] = room > 0 ? (room < d.max_pts ? room : d.max_pts) : 0;
    }
}

// addPoints (:71-79) + undistortedPoints (:262-267): the list the frame ends with = the kept points in the walk's order, then the new
// corners; every point lifted.
extern "C" __global__ __launch_bounds__(256) void fe_ri_finish_kernel(FeDev d, RiDev r) {
    const int tid = threadIdx.x;
    const int nk = r.ctl[RI_NK];
    const int nc = d.ncorners[0];
    const int nnew = nc < 0 ? 0 : nc;
    for (int k = tid; k < nk + nnew; k += 256) {
        float x, y;
        if (k < nk) {
            const int q = r.b_kept[k];
            const int i = r.idx2[r.order ? r.order[q] : q];
            x = d.next_xy[2 * i]; y = d.next_xy[2 * i + 1];
        } else {
            x = d.corners[2 * (k - nk)]; y = d.corners[2 * (k - nk) + 1];
            r.b_new_xy[2 * (k - nk)] = x; r.b_new_xy[2 * (k - nk) + 1] = y;
        }
        double ux, uy;
        ri_lift(r, x, y, ux, uy);
        r.b_un_xy[2 * k] = (float)ux; r.b_un_xy[2 * k + 1] = (float)uy;
    }
    if (tid == 0) {
        r.ctl[RI_NNEW] = nc;
        r.b_hdr[RI_NK] = nk; r.b_hdr[RI_NNEW] = nc; r.b_hdr[RI_N2] = r.ctl[RI_N2];
    }
}
