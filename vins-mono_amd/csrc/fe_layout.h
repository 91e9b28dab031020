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
