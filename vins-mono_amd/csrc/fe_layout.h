// fe_layout.h — device-side descriptor of the front-end state (shared by fe_host.hip and fe_kernels.hip).
#pragma once
#include <stdint.h>
#define FE_MAX_LEVELS 4
#define FE_MAX_CELLS 1024
#define FE_CAND_CAP 65536           // power of two >= number of 3x3 local maxima of a 752x480 frame

#define FE_CNT_STRIDE 64
#define FE_SLACK 256                // bytes in front of / behind the image planes and the frame slots that kernels may read (and ignore)
struct FeDev {
    int W, H, cams, max_level, max_pts, max_count;
    float min_eig_thr;
    double eps2;
    int lw[FE_MAX_LEVELS], lh[FE_MAX_LEVELS];
    const uint8_t* raw;             // [cams][H][W] incoming frames (before CLAHE)
    uint8_t* lut;                   // [cams][64][256]
    uint8_t* const* prev_planes;    // [level * cams + cam] -> plane (previous frame pyramid)
    uint8_t* const* cur_planes;     // [level * cams + cam]          (current frame pyramid)
    const int* npts;                // [cams]
    const float* prev_xy;           // [cams][max_pts][2]
    float* next_xy;                 // [cams][max_pts][2]
    uint8_t* status;                // [cams][max_pts]
    float* err;                     // [cams][max_pts]
    float* eig;                     // [cams][H][W]; allocated by vg_fe_keep_eig
    const uint8_t* mask;            // [cams][H][W]
    int keep_eig;                   // write the map to `eig` (vg_fe_keep_eig; it is an LDS-only intermediate otherwise)
    unsigned* ncand;                // [cams][FE_CNT_STRIDE]: [0] candidate count, [32] ordered-uint eig maximum; one
                                    // 256-byte line per stream (same-line atomics of different streams serialise in L2)
    unsigned long long* keys;       // [cams][FE_CAND_CAP]
    int cand_cap;
    const int* max_corners;         // [cams]
    float* corners;                 // [cams][max_pts][2]
    int* ncorners;                  // [cams]
};

#define FE_RANSAC_MAXIT 1000        // maxIters of cv::findFundamentalMat's RANSAC
#define FE_RANSAC_MAXPTS 1024
// ---- vg_fe_read_image: one call per frame (fe_frame.hip).  Device control block (ints), the head of the uploaded input block:
enum {
    RI_N = 0,          // points handed in (cur_pts)
    RI_PUBLISH,        // PUB_THIS_FRAME
    RI_N1,             // survivors of the tracking + border test
    RI_N2,             // survivors of rejectWithF (= RI_N1 when it did not run)
    RI_FALLBACK,       // bits RI_FB_*: the device could not finish rejectWithF itself
    RI_RANSAC,         // 1: rejectWithF ran (n1 >= 8) and status_f is meaningful
    RI_BEST,           // iteration whose model won (-1: none, nothing rejected)
    RI_NK,             // points setMask kept
    RI_NNEW,           // corners detected (-1: candidate list overflow)
    RI_NITERS,         // iterations that counted
    RI_CTL_INTS = 16
};
#define RI_FB_COLLINEAR 1           // a sample of the point-independent schedule would have been redrawn by OpenCV
#define RI_FB_LMEDS 2               // 8 <= n1 < 15: findFundamentalMat switches to LMedS
#define RI_FB_RANGE 4               // n1 beyond the resident schedule table
struct RiDev {
    int* ctl;                       // [RI_CTL_INTS]
    const float* xy_in;             // [cap][2] cur_pts
    int cap;                        // capacity of every per-point array (= max_pts of the stream)
    int* idx1;                      // [cap] input index of tracking survivor k
    int* idx2;                      // [cap] input index of rejectWithF survivor k
    float* p1;                      // [cap][2] lifted cur / forw points as rejectWithF hands them to findFundamentalMat
    float* p2;
    const int* order;               // [cap] setMask order: position q -> survivor index (into idx2)
    // results, block A (after tracking / rejectWithF) and block B (after setMask / detection): mirrored to the host as they are
    int* a_hdr;                     // [16] copy of ctl
    uint8_t* a_status_lk;           // [cap]
    uint8_t* a_status_f;            // [cap]
    float* a_forw_xy;               // [cap][2]
    float* a_un_xy;                 // [cap][2] lifted survivors (frames that are not published)
    int* b_hdr;                     // [16]
    int* b_kept;                    // [cap] positions (in the setMask order) of the kept points
    float* b_new_xy;                // [cap][2]
    float* b_un_xy;                 // [cap][2] lifted final list: kept points, then the new corners
    // RANSAC
    const int* niters_tab;          // [(FE_RANSAC_MAXPTS + 1)][tab_stride]: RANSACUpdateNumIters(0.99, (n - c) / n, 7, 1000) by (n, c)
    int tab_stride;
    const int* count;               // [FE_RANSAC_MAXIT] inliers of the iteration's best model (-1: no model)
    const unsigned long long* words;  // [FE_RANSAC_MAXIT][ceil(n1 / 64)] its inlier set
    double focal, half_w, half_h;   // FOCAL_LENGTH, COL / 2.0, ROW / 2.0
    double fx, fy, cx, cy, k1, k2, pp1, pp2;
    int max_cnt, radius;
    int* kept_xy;                   // [cap][2] rounded positions of the kept points (fe_stamp_kernel)
    const uint8_t* base_mask;       // fisheye mask or nullptr
};
// device scratch of the fundamental-matrix estimate (fe_ransac.hip), one allocation per handle
struct FeRansacBufs {
    float* p;                      // [2][FE_RANSAC_MAXPTS][2] the two point sets
    double* F;                     // [iteration][9] best model of the iteration
    double* med;                   // [iteration]
    int* cnt;                      // [iteration]
    int* sched;                    // [iteration][7]
    unsigned char* s;              // [FE_RANSAC_MAXPTS]
    double* models;                // [iteration][3][9]
    unsigned long long* words;     // [iteration][ceil(n / 64)] inlier ballots of the iteration's model
};
