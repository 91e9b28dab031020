// fe_host.hip — host side of the front-end path behind the C-ABI (include/vinsgpu.h, vg_fe_*): owns the per-camera
// pyramids / point / corner buffers in HBM, launches the kernels of fe_kernels.hip on the handle's stream.
// Replaces the OpenCV calls of FeatureTracker::readImage (feature_tracker/src/feature_tracker.cpp:87-93, :113, :149).
#include "vg_range.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include "fe_layout.h"
#include "vg_handle.h"
#include "../../include/vinsgpu.h"

#define HIPCHK(h, expr)                                                                            \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                          \
            return VG_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

extern "C" {
__global__ void fe_clahe_lut_kernel(FeDev d, int clip_limit, float lut_scale);
__global__ void fe_clahe_apply_kernel(FeDev d, uint8_t* const* dst_planes);
__global__ void fe_copy_kernel(FeDev d, uint8_t* const* dst_planes);
__global__ void fe_pyrdown_kernel(const uint8_t* const* src_planes, uint8_t* const* dst_planes, int sw, int sh, int waves_per_strip);
__global__ void fe_pyrdown_tile_kernel(const uint8_t* const* src_planes, uint8_t* const* dst_planes, int sw, int sh);
__global__ void fe_lk_kernel(FeDev d);
__global__ void fe_mineig_kernel(FeDev d, double quality);
hipError_t fe_launch_select(const FeDev& d, double quality, float min_dist, hipStream_t stream);
__global__ void fe_setmask_kernel(FeDev d, const float* pts_xy, const int* track_cnt, const int* npts, const uint8_t* const* base_masks,
                                  int radius, int* kept_index, int* n_kept, int* kept_xy);
__global__ void fe_stamp_kernel(FeDev d, const int* n_kept, const int* kept_xy, int radius);
__global__ void fe_lift_kernel(const float* pts_xy, int n, double fx, double fy, double cx, double cy, double k1, double k2, double p1,
                               double p2, float* out_xy);
// vg_fe_read_image (fe_frame.hip, fe_ransac.hip)
__global__ void fe_ri_after_lk_kernel(FeDev d, RiDev r);
__global__ void fe_ri_pick_kernel(RiDev r);
__global__ void fe_ri_setmask_kernel(FeDev d, RiDev r);
__global__ void fe_ri_finish_kernel(FeDev d, RiDev r);
__global__ void fe_ransac7_kernel(const float* p1, const float* p2, int n, const int* sched, int nsched, double* models, int* ctl);
__global__ void fe_ransac_count_kernel(const float* p1, const float* p2, int n, float thresh2, int lmeds, const double* models, int nsched,
                                       double* Fout, int* count, double* median, unsigned long long* inl_words, const int* ctl);
}
hipError_t fe_ransac_buffers(vg_handle* h, FeRansacBufs* b);
void fe_ransac_tables(int nmax, std::vector<int>& sched, std::vector<int>& niters, int stride);

// resident state of vg_fe_read_image: device arrays (one allocation), their pinned host mirrors, the RANSAC tables
struct RiState {
    int cap = 0;
    char* dev = nullptr;
    char* host = nullptr;              // pinned: [input block | block A | block B | order]
    size_t in_bytes = 0, a_bytes = 0, b_bytes = 0;
    char *d_in = nullptr, *d_a = nullptr, *d_b = nullptr;
    int *d_order = nullptr, *d_sched = nullptr, *d_tab = nullptr;
    uint8_t* d_base = nullptr;         // copy of the caller's fisheye mask
    const uint8_t* base_src = nullptr;
    RiDev r;
};
static void ri_free(RiState* q) {
    if (!q) return;
    (void)hipFree(q->dev); (void)hipFree(q->d_sched); (void)hipFree(q->d_tab); (void)hipFree(q->d_base);
    (void)hipHostFree(q->host);
    delete q;
}

struct FeState {
    FeDev d;
    int W = 0, H = 0, cams = 0, max_pts = 0;
    int flip = 0;                         // which plane set is "current"
    uint8_t* planes[2] = {nullptr, nullptr};   // two pyramids per camera, all levels, contiguous
    uint8_t* planes_alloc[2] = {nullptr, nullptr};   // (what hipMalloc returned: planes[k] - FE_SLACK)
    uint8_t* raw_alloc = nullptr;
    uint8_t** d_ptrs[2] = {nullptr, nullptr};  // device pointer tables [level*cams+cam]
    std::vector<uint8_t*> h_ptrs[2];
    // The same tables with level 0 pointing INTO the frame slot: pyramid set k <-> frame slot k.  A frame that is not equalized
    // is its own level 0 — the upload already put it into HBM, a copy into the pyramid's plane moves 361 KB per frame for
    // nothing (44 of 567 us per 256-stream step).  alias_on[k]: set k currently uses its frame slot as level 0.
    uint8_t** d_ptrs_alias[2] = {nullptr, nullptr};
    std::vector<uint8_t*> h_ptrs_alias[2];
    bool alias_on[2] = {false, false};
    size_t level_off[FE_MAX_LEVELS + 1];
    uint8_t* raw2[2] = {nullptr, nullptr};
    int raw_sel = 1;
    uint8_t *raw = nullptr, *lut = nullptr, *mask = nullptr, *status = nullptr;
    float *prev_xy = nullptr, *next_xy = nullptr, *err = nullptr, *eig = nullptr, *corners = nullptr;
    int *npts = nullptr, *max_corners = nullptr, *ncorners = nullptr;
    unsigned* ncand = nullptr;
    unsigned long long* keys = nullptr;
    std::vector<uint8_t> stage;           // pinned-ish host staging for strided uploads
    std::vector<int> h_npts;
    // setMask / lift, allocated on first use
    float* sm_pts = nullptr;
    int *sm_cnt = nullptr, *sm_n = nullptr, *sm_kidx = nullptr, *sm_nk = nullptr, *sm_kxy = nullptr;
    uint8_t* sm_base = nullptr;
    const uint8_t** sm_base_ptrs = nullptr;
    float *lift_in = nullptr, *lift_out = nullptr;
    bool have_prev = false;
    bool prev_clobbered = false;     // an upload has overwritten the frame slot the PREVIOUS pyramid uses as its level 0: no tracking until the next build
    std::vector<char> pushed_once;
    RiState* ri = nullptr;
};

extern "C" void fe_state_destroy(FeState* s) {
    if (!s) return;
    for (int k = 0; k < 2; ++k) { (void)hipFree(s->planes_alloc[k]); (void)hipFree(s->d_ptrs[k]); (void)hipFree(s->d_ptrs_alias[k]); }
    (void)hipFree(s->raw_alloc); (void)hipFree(s->lut); (void)hipFree(s->mask); (void)hipFree(s->status);
    (void)hipFree(s->prev_xy); (void)hipFree(s->next_xy); (void)hipFree(s->err); (void)hipFree(s->eig);
    (void)hipFree(s->corners); (void)hipFree(s->npts); (void)hipFree(s->max_corners);
    (void)hipFree(s->ncorners); (void)hipFree(s->ncand); (void)hipFree(s->keys);
    (void)hipFree(s->sm_pts); (void)hipFree(s->sm_cnt); (void)hipFree(s->sm_n); (void)hipFree(s->sm_kidx); (void)hipFree(s->sm_nk);
    (void)hipFree(s->sm_kxy); (void)hipFree(s->sm_base); (void)hipFree((void*)s->sm_base_ptrs); (void)hipFree(s->lift_in); (void)hipFree(s->lift_out);
    ri_free(s->ri);
    delete s;
}

static void refresh(FeState* s) {
    const int a = s->flip, b = s->flip ^ 1;
    s->d.cur_planes = s->alias_on[a] ? s->d_ptrs_alias[a] : s->d_ptrs[a];
    s->d.prev_planes = s->alias_on[b] ? s->d_ptrs_alias[b] : s->d_ptrs[b];
}

extern "C" int vg_fe_configure(vg_handle* h, int width, int height, int n_cams, int max_points) {
    if (!h || width < 32 || height < 32 || n_cams < 1 || max_points < 1) return VG_ERR_BAD_ARG;
    if ((width * height) % 4) { h->err = "width*height must be a multiple of 4"; return VG_ERR_UNSUPPORTED; }
    HIPCHK(h, hipSetDevice(h->device));
    if (h->fe) { fe_state_destroy(h->fe); h->fe = nullptr; }
    FeState* s = new FeState();
    h->fe = s;
    s->W = width; s->H = height; s->cams = n_cams; s->max_pts = max_points;
    FeDev& d = s->d;
    memset(&d, 0, sizeof(d));
    d.W = width; d.H = height; d.cams = n_cams; d.max_level = 3; d.max_pts = max_points; d.max_count = 30;
    d.min_eig_thr = 1e-4f; d.eps2 = 0.01 * 0.01;
    // buildOpticalFlowPyramid: levels while both dims stay > winSize (21)
    int lw = width, lh = height, nl = 0;
    size_t off = 0;
    for (int l = 0; l < FE_MAX_LEVELS; ++l) {
        d.lw[l] = lw; d.lh[l] = lh;
        s->level_off[l] = off;
        off += ((size_t)lw * lh + 255) / 256 * 256;
        nl = l;
        const int nw = (lw + 1) / 2, nh = (lh + 1) / 2;
        if (nw <= 21 || nh <= 21) break;
        lw = nw; lh = nh;
    }
    d.max_level = nl;
    s->level_off[nl + 1] = off;
    const size_t per_cam = off, npix = (size_t)width * height;
    for (int k = 0; k < 2; ++k) {
        // (FE_SLACK bytes in front of and behind the planes: fe_pyrdown_kernel's 16-byte windows start 4 bytes before a row and end
        //  up to 12 bytes after it; what they read there is never used)
        HIPCHK(h, hipMalloc((void**)&s->planes_alloc[k], per_cam * n_cams + 2 * FE_SLACK));
        HIPCHK(h, hipMemset(s->planes_alloc[k], 0, per_cam * n_cams + 2 * FE_SLACK));
        s->planes[k] = s->planes_alloc[k] + FE_SLACK;
        s->h_ptrs[k].resize((size_t)(nl + 1) * n_cams);
        for (int l = 0; l <= nl; ++l)
            for (int c = 0; c < n_cams; ++c) s->h_ptrs[k][(size_t)l * n_cams + c] = s->planes[k] + (size_t)c * per_cam + s->level_off[l];
        HIPCHK(h, hipMalloc((void**)&s->d_ptrs[k], sizeof(uint8_t*) * s->h_ptrs[k].size()));
        HIPCHK(h, hipMemcpy(s->d_ptrs[k], s->h_ptrs[k].data(), sizeof(uint8_t*) * s->h_ptrs[k].size(), hipMemcpyHostToDevice));
    }
    // candidate keys per stream: a power of two (the fall-back sort is bitonic) that holds every 3x3 local maximum of a frame without
    // ties (W H / 4) -- 65536 for 752 x 480 -- so that the timing-dependent pruning bound of fe_mineig_kernel cannot overflow it on
    // larger frames either; an overflow (plateaus of equal positive values) is reported, never truncated silently
    {
        size_t cap = FE_CAND_CAP;
        while (cap < npix / 4) cap *= 2;
        d.cand_cap = (int)cap;
    }
    HIPCHK(h, hipMalloc((void**)&s->raw_alloc, 2 * npix * n_cams + 2 * FE_SLACK));
    HIPCHK(h, hipMemset(s->raw_alloc, 0, 2 * npix * n_cams + 2 * FE_SLACK));
    s->raw = s->raw_alloc + FE_SLACK;
    s->raw2[0] = s->raw; s->raw2[1] = s->raw + npix * n_cams;
    for (int k = 0; k < 2; ++k) {
        s->h_ptrs_alias[k] = s->h_ptrs[k];
        for (int c = 0; c < n_cams; ++c) s->h_ptrs_alias[k][c] = s->raw2[k] + (size_t)c * npix;
        HIPCHK(h, hipMalloc((void**)&s->d_ptrs_alias[k], sizeof(uint8_t*) * s->h_ptrs_alias[k].size()));
        HIPCHK(h, hipMemcpy(s->d_ptrs_alias[k], s->h_ptrs_alias[k].data(), sizeof(uint8_t*) * s->h_ptrs_alias[k].size(), hipMemcpyHostToDevice));
        s->alias_on[k] = false;
    }
    HIPCHK(h, hipMalloc((void**)&s->lut, (size_t)n_cams * 64 * 256));
    HIPCHK(h, hipMalloc((void**)&s->mask, npix * n_cams));
    HIPCHK(h, hipMemset(s->mask, 255, npix * n_cams));
    HIPCHK(h, hipMalloc((void**)&s->status, (size_t)n_cams * max_points));
    HIPCHK(h, hipMalloc((void**)&s->prev_xy, sizeof(float) * 2 * n_cams * max_points));
    HIPCHK(h, hipMalloc((void**)&s->next_xy, sizeof(float) * 2 * n_cams * max_points));
    HIPCHK(h, hipMalloc((void**)&s->err, sizeof(float) * n_cams * max_points));
    HIPCHK(h, hipMalloc((void**)&s->corners, sizeof(float) * 2 * n_cams * max_points));
    HIPCHK(h, hipMalloc((void**)&s->npts, sizeof(int) * n_cams));
    HIPCHK(h, hipMemset(s->npts, 0, sizeof(int) * n_cams));
    HIPCHK(h, hipMalloc((void**)&s->max_corners, sizeof(int) * n_cams));
    HIPCHK(h, hipMalloc((void**)&s->ncorners, sizeof(int) * n_cams));
    HIPCHK(h, hipMalloc((void**)&s->ncand, sizeof(unsigned) * FE_CNT_STRIDE * n_cams));
    HIPCHK(h, hipMalloc((void**)&s->keys, sizeof(unsigned long long) * (size_t)d.cand_cap * n_cams));
    d.raw = s->raw2[s->raw_sel]; d.lut = s->lut; d.npts = s->npts; d.prev_xy = s->prev_xy; d.next_xy = s->next_xy; d.status = s->status;
    d.err = s->err; d.eig = nullptr; d.keep_eig = 0; d.mask = s->mask; d.ncand = s->ncand; d.keys = s->keys;
    d.max_corners = s->max_corners; d.corners = s->corners; d.ncorners = s->ncorners;
    s->h_npts.assign(n_cams, 0);
    s->pushed_once.assign(n_cams, 0);
    refresh(s);
    return VG_OK;
}

static int fe_upload_async(vg_handle* h, const uint8_t* const* imgs, int stride) {
    if (!h || !h->fe || !imgs) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    const size_t npix = (size_t)s->W * s->H;
    for (int c = 0; c < s->cams; ++c)                 // validate before touching any state
        if (!imgs[c]) { h->err = "vg_fe_upload_frames: every stream needs a frame (batched streams advance together)"; return VG_ERR_BAD_ARG; }
    if (stride < s->W) { h->err = "vg_fe_upload_frames: stride smaller than the frame width"; return VG_ERR_BAD_ARG; }
    // into the frame slot of the pyramid set the next build fills (set k <-> slot k): never the slot the CURRENT pyramid may be
    // using as its level 0 -- that image is the "previous" one of the next tracking step
    s->raw_sel = s->flip ^ 1;
    s->d.raw = s->raw2[s->raw_sel];
    // That slot may be level 0 of the PREVIOUS pyramid (frame n - 1, aliased): uploading frame n + 1 before frame n has been tracked
    // would make vg_fe_track* read the new image as the old one.  The order upload -> build -> track is fine (the build rotates the
    // sets); upload -> track is refused until a build has run.
    s->prev_clobbered = s->have_prev && s->alias_on[s->flip ^ 1];
    // frames that follow each other in memory without padding (one buffer for all streams: a ring of a capture driver, the bench's
    // staging area) travel as ONE copy; otherwise one copy per stream (2-D when the rows are padded)
    bool contiguous = stride == s->W;
    for (int c = 1; c < s->cams && contiguous; ++c) contiguous = imgs[c] == imgs[c - 1] + npix;
    if (contiguous) {
        HIPCHK(h, hipMemcpyAsync(s->raw2[s->raw_sel], imgs[0], npix * s->cams, hipMemcpyHostToDevice, h->stream));
    } else {
        for (int c = 0; c < s->cams; ++c) {
            if (stride == s->W) HIPCHK(h, hipMemcpyAsync(s->raw2[s->raw_sel] + (size_t)c * npix, imgs[c], npix, hipMemcpyHostToDevice, h->stream));
            else HIPCHK(h, hipMemcpy2DAsync(s->raw2[s->raw_sel] + (size_t)c * npix, s->W, imgs[c], stride, s->W, s->H, hipMemcpyHostToDevice, h->stream));
        }
    }
    return VG_OK;
}

extern "C" int vg_fe_upload_frames(vg_handle* h, const uint8_t* const* imgs, int stride) {
    VG_RANGE("vg_fe_upload_frames");
    const int rc = fe_upload_async(h, imgs, stride);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return VG_OK;
}

extern "C" int vg_fe_frame_slot(vg_handle* h) {
    return (h && h->fe) ? h->fe->raw_sel : -1;
}

extern "C" int vg_fe_select_frames(vg_handle* h, int slot) {
    if (!h || !h->fe || slot < 0 || slot > 1) return VG_ERR_BAD_ARG;
    h->fe->raw_sel = slot;
    h->fe->d.raw = h->fe->raw2[slot];
    return VG_OK;
}

extern "C" int vg_fe_build_async(vg_handle* h, int equalize) {
    VG_RANGE("vg_fe_build_async");
    if (!h || !h->fe) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    if (equalize && (s->d.W % 8 || s->d.H % 8)) { h->err = "CLAHE needs width and height divisible by 8"; return VG_ERR_UNSUPPORTED; }
    // (fe_clahe_apply_kernel lays the interpolation cells of a row over the 256 threads of a workgroup: W / 8 / 4 + 3 of them)
    if (equalize && s->d.W / 8 / 4 + 3 > 256) { h->err = "CLAHE: frame wider than the interpolation-cell tiling (W / 32 + 3 > 256)"; return VG_ERR_UNSUPPORTED; }
    const bool first = !s->have_prev;
    s->prev_clobbered = false;
    s->flip ^= 1;                                     // previous <- current (only after the arguments are known to be valid)
    // level 0 = the frame slot itself when the frame is used as it is and sits in the slot that belongs to this pyramid set
    // (the normal alternation: upload -> build -> upload -> build); otherwise it is written into the set's own plane
    const bool alias = !equalize && !first && s->raw_sel == s->flip;
    s->alias_on[s->flip] = alias;
    refresh(s);
    const FeDev& d = s->d;
    uint8_t* const* cur0 = alias ? s->d_ptrs_alias[s->flip] : s->d_ptrs[s->flip];
    if (equalize) {
        const int area = (d.W / 8) * (d.H / 8);
        int clip = (int)(3.0 * area / 256);
        clip = clip < 1 ? 1 : clip;
        hipLaunchKernelGGL(fe_clahe_lut_kernel, dim3(64, d.cams), dim3(256), 0, h->stream, d, clip, 255.f / area);
        hipLaunchKernelGGL(fe_clahe_apply_kernel, dim3(9, 9, d.cams), dim3(256), 0, h->stream, d, cur0);      // one workgroup per interpolation cell
    } else if (!alias) {
        hipLaunchKernelGGL(fe_copy_kernel, dim3(256, d.cams), dim3(256), 0, h->stream, d, cur0);
    }
    for (int l = 1; l <= d.max_level; ++l) {
        const int sw = d.lw[l - 1], sh = d.lh[l - 1];
        if ((sw & 3) == 0) {
            // thread = 4 output columns x 8 output rows, a wavefront = 256 columns of one strip of rows (fe_kernels.hip)
            const int dw = sw / 2, dh = (sh + 1) / 2, wps = ((dw + 3) / 4 + 63) / 64, strips = (dh + 7) / 8;
            hipLaunchKernelGGL(fe_pyrdown_kernel, dim3((strips * wps + 3) / 4, 1, d.cams), dim3(256), 0, h->stream,
                               (const uint8_t* const*)(cur0 + (size_t)(l - 1) * d.cams), cur0 + (size_t)l * d.cams, sw, sh, wps);
        } else {
            hipLaunchKernelGGL(fe_pyrdown_tile_kernel, dim3(((sw + 1) / 2 + 63) / 64, ((sh + 1) / 2 + 15) / 16, d.cams), dim3(256), 0, h->stream,
                               (const uint8_t* const*)(cur0 + (size_t)(l - 1) * d.cams), cur0 + (size_t)l * d.cams, sw, sh);
        }
    }
    HIPCHK(h, hipGetLastError());
    if (first) {
        // prev_img = cur_img = forw_img = img on the first frame (feature_tracker.cpp:97-100)
        const size_t bytes = s->level_off[d.max_level + 1] * d.cams;
        HIPCHK(h, hipMemcpyAsync(s->planes[s->flip ^ 1], s->planes[s->flip], bytes, hipMemcpyDeviceToDevice, h->stream));
        s->have_prev = true;
    }
    return VG_OK;
}

extern "C" int vg_fe_push_frames(vg_handle* h, const uint8_t* const* imgs, int stride, int equalize) {
    VG_RANGE("vg_fe_push_frames");
    int rc = vg_fe_upload_frames(h, imgs, stride);
    if (rc) return rc;
    rc = vg_fe_build_async(h, equalize);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return VG_OK;
}

extern "C" int vg_fe_track_upload(vg_handle* h, const float* prev_xy, const int* n) {
    if (!h || !h->fe || !prev_xy || !n) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    for (int c = 0; c < s->cams; ++c) {
        if (n[c] < 0 || n[c] > s->max_pts) { h->err = "point count out of range"; return VG_ERR_BAD_ARG; }
        s->h_npts[c] = n[c];
    }
    HIPCHK(h, hipMemcpyAsync(s->prev_xy, prev_xy, sizeof(float) * 2 * s->cams * s->max_pts, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(s->npts, n, sizeof(int) * s->cams, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return VG_OK;
}

extern "C" int vg_fe_track_async(vg_handle* h) {
    VG_RANGE("vg_fe_track_async");
    if (!h || !h->fe) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    if (s->prev_clobbered) {
        h->err = "vg_fe_track: a frame was uploaded into the slot of the previous image since the last build (upload -> build -> track)";
        return VG_ERR_BAD_ARG;
    }
    int nmax = 0;
    for (int c = 0; c < s->cams; ++c) nmax = nmax > s->h_npts[c] ? nmax : s->h_npts[c];
    if (nmax == 0) return VG_OK;
    hipLaunchKernelGGL(fe_lk_kernel, dim3(nmax, s->cams), dim3(64), 0, h->stream, s->d);
    HIPCHK(h, hipGetLastError());
    return VG_OK;
}

extern "C" int vg_fe_track_download(vg_handle* h, float* next_xy, uint8_t* status, float* err) {
    if (!h || !h->fe) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    const size_t n = (size_t)s->cams * s->max_pts;
    if (next_xy) HIPCHK(h, hipMemcpyAsync(next_xy, s->next_xy, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, h->stream));
    if (status) HIPCHK(h, hipMemcpyAsync(status, s->status, n, hipMemcpyDeviceToHost, h->stream));
    if (err) HIPCHK(h, hipMemcpyAsync(err, s->err, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return VG_OK;
}

extern "C" int vg_fe_track(vg_handle* h, int cam, const float* prev_xy, int n, float* next_xy, uint8_t* status, float* err) {
    VG_RANGE("vg_fe_track");
    if (!h || !h->fe || cam < 0 || cam >= h->fe->cams || n < 0 || n > h->fe->max_pts || (n && (!prev_xy || !next_xy || !status)))
        return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    if (n == 0) return VG_OK;
    const size_t o = (size_t)cam * s->max_pts;
    std::vector<int> saved = s->h_npts;
    for (int c = 0; c < s->cams; ++c) s->h_npts[c] = (c == cam) ? n : 0;
    HIPCHK(h, hipMemcpyAsync(s->prev_xy + o * 2, prev_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(s->npts, s->h_npts.data(), sizeof(int) * s->cams, hipMemcpyHostToDevice, h->stream));
    int rc = vg_fe_track_async(h);
    if (rc) { s->h_npts = saved; return rc; }
    HIPCHK(h, hipMemcpyAsync(next_xy, s->next_xy + o * 2, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(status, s->status + o, n, hipMemcpyDeviceToHost, h->stream));
    if (err) HIPCHK(h, hipMemcpyAsync(err, s->err + o, sizeof(float) * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    // the single-stream call borrowed the batched point counts: put the other streams' counts back (host and device)
    s->h_npts = saved;
    HIPCHK(h, hipMemcpy(s->npts, s->h_npts.data(), sizeof(int) * s->cams, hipMemcpyHostToDevice));
    return VG_OK;
}

extern "C" int vg_fe_detect_upload(vg_handle* h, const uint8_t* const* masks, const int* max_corners) {
    if (!h || !h->fe || !max_corners) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    const size_t npix = (size_t)s->W * s->H;
    for (int c = 0; c < s->cams; ++c) {
        if (max_corners[c] < 0 || max_corners[c] > s->max_pts) { h->err = "max_corners out of range"; return VG_ERR_BAD_ARG; }
        if (masks && masks[c]) HIPCHK(h, hipMemcpyAsync(s->mask + (size_t)c * npix, masks[c], npix, hipMemcpyHostToDevice, h->stream));
        else HIPCHK(h, hipMemsetAsync(s->mask + (size_t)c * npix, 255, npix, h->stream));
    }
    HIPCHK(h, hipMemcpyAsync(s->max_corners, max_corners, sizeof(int) * s->cams, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return VG_OK;
}

extern "C" int vg_fe_detect_async(vg_handle* h, double quality, double min_dist) {
    VG_RANGE("vg_fe_detect_async");
    if (!h || !h->fe) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    const FeDev& d = s->d;
    const int cell = (int)std::lrint(min_dist) < 1 ? 1 : (int)std::lrint(min_dist);
    if (((d.W + cell - 1) / cell) * ((d.H + cell - 1) / cell) > FE_MAX_CELLS) { h->err = "min_dist too small for the cell grid"; return VG_ERR_UNSUPPORTED; }
    HIPCHK(h, hipMemsetAsync(s->ncand, 0, sizeof(unsigned) * FE_CNT_STRIDE * d.cams, h->stream));
    // min-eigenvalue map + candidates in one kernel (64 x 16 tiles), then the exact threshold, the sort and the min-distance walk
    const dim3 g((d.W + 63) / 64, (d.H + 15) / 16, d.cams);
    hipLaunchKernelGGL(fe_mineig_kernel, g, dim3(256), 0, h->stream, d, quality);
    HIPCHK(h, fe_launch_select(d, quality, (float)min_dist, h->stream));
    HIPCHK(h, hipGetLastError());
    return VG_OK;
}

extern "C" int vg_fe_detect_download(vg_handle* h, float* out_xy, int* out_n) {
    if (!h || !h->fe || !out_xy || !out_n) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    HIPCHK(h, hipMemcpyAsync(out_xy, s->corners, sizeof(float) * 2 * s->cams * s->max_pts, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(out_n, s->ncorners, sizeof(int) * s->cams, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int c = 0; c < s->cams; ++c)
        if (out_n[c] < 0) { out_n[c] = 0; h->err = "goodFeaturesToTrack: candidate list overflow (more 3x3 maxima than the key buffer holds)"; return VG_ERR_UNSUPPORTED; }
    return VG_OK;
}

extern "C" int vg_fe_detect(vg_handle* h, int cam, const uint8_t* mask, int max_corners, double quality, double min_dist,
                            float* out_xy, int* out_n) {
    VG_RANGE("vg_fe_detect");
    if (!h || !h->fe || cam < 0 || cam >= h->fe->cams || !out_xy || !out_n || max_corners < 0 || max_corners > h->fe->max_pts)
        return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    *out_n = 0;
    if (max_corners == 0) return VG_OK;
    const size_t npix = (size_t)s->W * s->H;
    std::vector<int> mc(s->cams, 0);
    mc[cam] = max_corners;
    if (mask) HIPCHK(h, hipMemcpyAsync(s->mask + (size_t)cam * npix, mask, npix, hipMemcpyHostToDevice, h->stream));
    else HIPCHK(h, hipMemsetAsync(s->mask + (size_t)cam * npix, 255, npix, h->stream));
    HIPCHK(h, hipMemcpyAsync(s->max_corners, mc.data(), sizeof(int) * s->cams, hipMemcpyHostToDevice, h->stream));
    int rc = vg_fe_detect_async(h, quality, min_dist);
    if (rc) return rc;
    int n = 0;
    HIPCHK(h, hipMemcpyAsync(&n, s->ncorners + cam, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n < 0) { h->err = "goodFeaturesToTrack: candidate list overflow (more 3x3 maxima than the key buffer holds)"; return VG_ERR_UNSUPPORTED; }
    if (n > 0) HIPCHK(h, hipMemcpy(out_xy, s->corners + (size_t)cam * s->max_pts * 2, sizeof(float) * 2 * n, hipMemcpyDeviceToHost));
    *out_n = n;
    return VG_OK;
}

// ---- FeatureTracker::setMask on the device (feature_tracker.cpp:36-69)
extern "C" int vg_fe_set_mask(vg_handle* h, const float* pts_xy, const int* track_cnt, const int* n, const uint8_t* const* base_masks,
                              int radius, int* kept_index, int* n_kept) {
    VG_RANGE("vg_fe_set_mask");
    if (!h || !h->fe || !pts_xy || !track_cnt || !n || !kept_index || !n_kept || radius < 0) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    if (s->max_pts > 2048) { h->err = "vg_fe_set_mask: max_points > 2048"; return VG_ERR_UNSUPPORTED; }
    for (int c = 0; c < s->cams; ++c)
        if (n[c] < 0 || n[c] > s->max_pts) { h->err = "vg_fe_set_mask: point count out of range"; return VG_ERR_BAD_ARG; }
    const size_t npix = (size_t)s->W * s->H, cap = (size_t)s->cams * s->max_pts;
    if (!s->sm_pts) {
        HIPCHK(h, hipMalloc((void**)&s->sm_pts, sizeof(float) * 2 * cap));
        HIPCHK(h, hipMalloc((void**)&s->sm_cnt, sizeof(int) * cap));
        HIPCHK(h, hipMalloc((void**)&s->sm_n, sizeof(int) * s->cams));
        HIPCHK(h, hipMalloc((void**)&s->sm_kidx, sizeof(int) * cap));
        HIPCHK(h, hipMalloc((void**)&s->sm_nk, sizeof(int) * s->cams));
        HIPCHK(h, hipMalloc((void**)&s->sm_kxy, sizeof(int) * 2 * cap));
        HIPCHK(h, hipMalloc((void**)&s->sm_base_ptrs, sizeof(uint8_t*) * s->cams));
    }
    bool any_base = false;
    for (int c = 0; c < s->cams; ++c) any_base = any_base || (base_masks && base_masks[c]);
    if (any_base && !s->sm_base) HIPCHK(h, hipMalloc((void**)&s->sm_base, npix * s->cams));
    std::vector<const uint8_t*> ptrs(s->cams, nullptr);
    for (int c = 0; c < s->cams; ++c) {
        uint8_t* dm = s->mask + (size_t)c * npix;
        if (base_masks && base_masks[c]) {
            HIPCHK(h, hipMemcpyAsync(s->sm_base + (size_t)c * npix, base_masks[c], npix, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(dm, s->sm_base + (size_t)c * npix, npix, hipMemcpyDeviceToDevice, h->stream));
            ptrs[c] = s->sm_base + (size_t)c * npix;
        } else {
            HIPCHK(h, hipMemsetAsync(dm, 255, npix, h->stream));
        }
    }
    HIPCHK(h, hipMemcpyAsync((void*)s->sm_base_ptrs, ptrs.data(), sizeof(uint8_t*) * s->cams, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(s->sm_pts, pts_xy, sizeof(float) * 2 * cap, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(s->sm_cnt, track_cnt, sizeof(int) * cap, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(s->sm_n, n, sizeof(int) * s->cams, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(fe_setmask_kernel, dim3(s->cams), dim3(256), 0, h->stream, s->d, s->sm_pts, s->sm_cnt, s->sm_n,
                       any_base ? s->sm_base_ptrs : (const uint8_t* const*)nullptr, radius, s->sm_kidx, s->sm_nk, s->sm_kxy);
    hipLaunchKernelGGL(fe_stamp_kernel, dim3(s->max_pts, s->cams), dim3(256), 0, h->stream, s->d, s->sm_nk, s->sm_kxy, radius);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(kept_index, s->sm_kidx, sizeof(int) * cap, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(n_kept, s->sm_nk, sizeof(int) * s->cams, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return VG_OK;
}

// goodFeaturesToTrack with the mask vg_fe_set_mask left on the device (no mask upload)
extern "C" int vg_fe_detect_masked(vg_handle* h, int cam, int max_corners, double quality, double min_dist, float* out_xy, int* out_n) {
    VG_RANGE("vg_fe_detect_masked");
    if (!h || !h->fe || cam < 0 || cam >= h->fe->cams || !out_xy || !out_n || max_corners < 0 || max_corners > h->fe->max_pts)
        return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    *out_n = 0;
    if (max_corners == 0) return VG_OK;
    std::vector<int> mc(s->cams, 0);
    mc[cam] = max_corners;
    HIPCHK(h, hipMemcpyAsync(s->max_corners, mc.data(), sizeof(int) * s->cams, hipMemcpyHostToDevice, h->stream));
    int rc = vg_fe_detect_async(h, quality, min_dist);
    if (rc) return rc;
    int n = 0;
    HIPCHK(h, hipMemcpyAsync(&n, s->ncorners + cam, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n < 0) { h->err = "goodFeaturesToTrack: candidate list overflow (more 3x3 maxima than the key buffer holds)"; return VG_ERR_UNSUPPORTED; }
    if (n > 0) HIPCHK(h, hipMemcpy(out_xy, s->corners + (size_t)cam * s->max_pts * 2, sizeof(float) * 2 * n, hipMemcpyDeviceToHost));
    *out_n = n;
    return VG_OK;
}

extern "C" int vg_fe_get_mask(vg_handle* h, int cam, uint8_t* out) {
    if (!h || !h->fe || cam < 0 || cam >= h->fe->cams || !out) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    const size_t npix = (size_t)s->W * s->H;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, s->mask + (size_t)cam * npix, npix, hipMemcpyDeviceToHost));
    return VG_OK;
}

// PinholeCamera::liftProjective for a batch of points (feature_tracker.cpp:258-271)
extern "C" int vg_fe_undistort(vg_handle* h, const float* pts_xy, int n, const double* intr, float* out_xy) {
    if (!h || !h->fe || n < 0 || !intr || (n > 0 && (!pts_xy || !out_xy))) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    if (n == 0) return VG_OK;
    if ((size_t)n > (size_t)s->cams * s->max_pts) { h->err = "vg_fe_undistort: more points than configured"; return VG_ERR_BAD_ARG; }
    if (!s->lift_in) {
        HIPCHK(h, hipMalloc((void**)&s->lift_in, sizeof(float) * 2 * s->cams * s->max_pts));
        HIPCHK(h, hipMalloc((void**)&s->lift_out, sizeof(float) * 2 * s->cams * s->max_pts));
    }
    HIPCHK(h, hipMemcpyAsync(s->lift_in, pts_xy, sizeof(float) * 2 * n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(fe_lift_kernel, dim3((n + 255) / 256), dim3(256), 0, h->stream, s->lift_in, n, intr[0], intr[1], intr[2], intr[3],
                       intr[4], intr[5], intr[6], intr[7], s->lift_out);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out_xy, s->lift_out, sizeof(float) * 2 * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return VG_OK;
}

extern "C" int vg_fe_get_level(vg_handle* h, int cam, int which, int level, uint8_t* out, int* w, int* hgt) {
    if (!h || !h->fe || cam < 0 || cam >= h->fe->cams || level < 0 || level > h->fe->d.max_level || !out) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    const int set = which ? (s->flip ^ 1) : s->flip;
    const int lw = s->d.lw[level], lh = s->d.lh[level];
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const std::vector<uint8_t*>& tab = s->alias_on[set] ? s->h_ptrs_alias[set] : s->h_ptrs[set];
    HIPCHK(h, hipMemcpy(out, tab[(size_t)level * s->cams + cam], (size_t)lw * lh, hipMemcpyDeviceToHost));
    if (w) *w = lw;
    if (hgt) *hgt = lh;
    return VG_OK;
}

// The min-eigenvalue map is an LDS-only intermediate of the detection since round 4; a caller that wants to look at it (the parity
// tests do) asks for it before the detection: the map of every following detection is then written to HBM as well.
extern "C" int vg_fe_keep_eig(vg_handle* h, int on) {
    if (!h || !h->fe) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    if (on && !s->eig) {
        HIPCHK(h, hipMalloc((void**)&s->eig, sizeof(float) * (size_t)s->W * s->H * s->cams));
        HIPCHK(h, hipMemset(s->eig, 0, sizeof(float) * (size_t)s->W * s->H * s->cams));       // (vg_fe_get_eig before any detection: zeros, not stale memory)
    }
    s->d.eig = s->eig;
    s->d.keep_eig = on ? 1 : 0;
    return VG_OK;
}

extern "C" int vg_fe_get_eig(vg_handle* h, int cam, float* out) {
    if (!h || !h->fe || cam < 0 || cam >= h->fe->cams || !out) return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    if (!s->d.keep_eig) { h->err = "vg_fe_get_eig: call vg_fe_keep_eig(h, 1) before the detection whose map is wanted"; return VG_ERR_BAD_ARG; }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpy(out, s->eig + (size_t)cam * s->W * s->H, sizeof(float) * s->W * s->H, hipMemcpyDeviceToHost));
    return VG_OK;
}

// ================================================================================================ one call per frame
static size_t ri_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int ri_build(vg_handle* h, FeState* s, RiState* q);

// (the state is attached to the stream only when it is complete: a failed allocation leaves nothing half-made behind)
static int ri_ensure(vg_handle* h, FeState* s) {
    if (s->ri) return VG_OK;
    RiState* q = new RiState();
    const int rc = ri_build(h, s, q);
    if (rc != VG_OK) { ri_free(q); return rc; }
    s->ri = q;
    return VG_OK;
}

static int ri_build(vg_handle* h, FeState* s, RiState* q) {
    const size_t cap = (size_t)s->max_pts;
    q->cap = (int)cap;
    // input block: control ints | cur_pts;  block A: header | status_lk | status_f | forw_xy | un_xy;  block B: header | kept | new_xy | un_xy
    q->in_bytes = ri_up(sizeof(int) * RI_CTL_INTS + sizeof(float) * 2 * cap, 256);
    const size_t a_st = 64, a_sf = a_st + ri_up(cap, 16), a_fw = a_sf + ri_up(cap, 16), a_un = a_fw + sizeof(float) * 2 * cap;
    q->a_bytes = ri_up(a_un + sizeof(float) * 2 * cap, 256);
    const size_t b_k = 64, b_nw = b_k + sizeof(int) * cap, b_un = b_nw + sizeof(float) * 2 * cap;
    q->b_bytes = ri_up(b_un + sizeof(float) * 2 * cap, 256);
    const size_t o_idx1 = q->in_bytes + q->a_bytes + q->b_bytes, o_idx2 = o_idx1 + sizeof(int) * cap, o_p1 = o_idx2 + sizeof(int) * cap;
    const size_t o_p2 = o_p1 + sizeof(float) * 2 * cap, o_ord = o_p2 + sizeof(float) * 2 * cap, o_kxy = o_ord + sizeof(int) * cap;
    const size_t total = o_kxy + sizeof(int) * 2 * cap;
    HIPCHK(h, hipMalloc((void**)&q->dev, total));
    HIPCHK(h, hipMemset(q->dev, 0, total));
    HIPCHK(h, hipHostMalloc((void**)&q->host, q->in_bytes + q->a_bytes + q->b_bytes + sizeof(int) * cap, 0));
    memset(q->host, 0, q->in_bytes + q->a_bytes + q->b_bytes + sizeof(int) * cap);
    q->d_in = q->dev; q->d_a = q->dev + q->in_bytes; q->d_b = q->d_a + q->a_bytes;
    q->d_order = (int*)(q->dev + o_ord);
    RiDev& r = q->r;
    memset(&r, 0, sizeof(r));
    r.ctl = (int*)q->d_in; r.xy_in = (const float*)(q->d_in + sizeof(int) * RI_CTL_INTS); r.cap = (int)cap;
    r.idx1 = (int*)(q->dev + o_idx1); r.idx2 = (int*)(q->dev + o_idx2); r.p1 = (float*)(q->dev + o_p1); r.p2 = (float*)(q->dev + o_p2);
    r.kept_xy = (int*)(q->dev + o_kxy);
    r.a_hdr = (int*)q->d_a; r.a_status_lk = (uint8_t*)(q->d_a + a_st); r.a_status_f = (uint8_t*)(q->d_a + a_sf);
    r.a_forw_xy = (float*)(q->d_a + a_fw); r.a_un_xy = (float*)(q->d_a + a_un);
    r.b_hdr = (int*)q->d_b; r.b_kept = (int*)(q->d_b + b_k); r.b_new_xy = (float*)(q->d_b + b_nw); r.b_un_xy = (float*)(q->d_b + b_un);
    // the point-independent part of OpenCV's sample schedule and the iteration bounds, for every point count the stream can have
    const int nmax = (int)std::min<size_t>(cap, FE_RANSAC_MAXPTS);
    std::vector<int> sched, tab;
    r.tab_stride = nmax + 1;
    fe_ransac_tables(nmax, sched, tab, r.tab_stride);
    HIPCHK(h, hipMalloc((void**)&q->d_sched, sizeof(int) * std::max<size_t>(sched.size(), 1)));
    HIPCHK(h, hipMalloc((void**)&q->d_tab, sizeof(int) * tab.size()));
    if (!sched.empty()) HIPCHK(h, hipMemcpy(q->d_sched, sched.data(), sizeof(int) * sched.size(), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(q->d_tab, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice));
    r.niters_tab = q->d_tab;
    return VG_OK;
}

extern "C" int vg_fe_read_image(vg_handle* h, const vg_fe_frame_in* in, vg_fe_frame_out* out) {
    VG_RANGE("vg_fe_read_image");
    if (!h || !h->fe || !in || !out || in->struct_size != (int)sizeof(vg_fe_frame_in) || !in->img || in->n < 0 || (in->n && !in->cur_xy))
        return VG_ERR_BAD_ARG;
    FeState* s = h->fe;
    if (s->cams != 1) { h->err = "vg_fe_read_image: one stream per handle (vg_fe_configure with n_cams == 1)"; return VG_ERR_UNSUPPORTED; }
    if (s->max_pts > 2048) { h->err = "vg_fe_read_image: max_points > 2048"; return VG_ERR_UNSUPPORTED; }
    if (in->n > s->max_pts || in->max_cnt < 0 || in->max_cnt > s->max_pts) { h->err = "vg_fe_read_image: n / max_cnt beyond max_points of vg_fe_configure"; return VG_ERR_BAD_ARG; }
    if (in->publish && (in->min_dist < 0 || !(in->f_threshold > 0) || !(in->quality > 0))) { h->err = "vg_fe_read_image: min_dist / f_threshold / quality"; return VG_ERR_BAD_ARG; }
    // (everything that can be refused from the arguments is refused before the frame goes up and the pyramid ring turns)
    if (in->n > 0 && !s->have_prev) { h->err = "vg_fe_read_image: points to track but no previous frame"; return VG_ERR_BAD_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc = ri_ensure(h, s);
    if (rc) return rc;
    RiState* q = s->ri;
    const size_t npix = (size_t)s->W * s->H, cap = (size_t)q->cap;
    memset(out, 0, sizeof(*out));
    RiDev r = q->r;
    r.focal = in->focal_length; r.half_w = s->W / 2.0; r.half_h = s->H / 2.0;
    r.fx = in->intr[0]; r.fy = in->intr[1]; r.cx = in->intr[2]; r.cy = in->intr[3];
    r.k1 = in->intr[4]; r.k2 = in->intr[5]; r.pp1 = in->intr[6]; r.pp2 = in->intr[7];
    r.max_cnt = in->max_cnt; r.radius = in->min_dist;
    FeRansacBufs rb;
    { const hipError_t e = fe_ransac_buffers(h, &rb); if (e != hipSuccess) { h->err = std::string("vg_fe_read_image: ") + hipGetErrorString(e); return VG_ERR_HIP; } }
    r.count = rb.cnt; r.words = rb.words;
    // the fisheye mask travels once (the reference loads it at start-up, feature_tracker_node.cpp)
    if (in->publish && in->base_mask && in->base_mask != q->base_src) {
        if (!q->d_base) HIPCHK(h, hipMalloc((void**)&q->d_base, npix));
        HIPCHK(h, hipMemcpyAsync(q->d_base, in->base_mask, npix, hipMemcpyHostToDevice, h->stream));
        q->base_src = in->base_mask;
    }
    r.base_mask = (in->publish && in->base_mask) ? q->d_base : nullptr;
    // ---- upload: the frame, then ONE block with the control ints and cur_pts
    const uint8_t* planes[1] = {in->img};
    rc = fe_upload_async(h, planes, in->stride);
    if (rc) return rc;
    int* hctl = (int*)q->host;
    memset(hctl, 0, sizeof(int) * RI_CTL_INTS);
    hctl[RI_N] = in->n; hctl[RI_PUBLISH] = in->publish ? 1 : 0; hctl[RI_BEST] = -1;
    if (in->n) memcpy(q->host + sizeof(int) * RI_CTL_INTS, in->cur_xy, sizeof(float) * 2 * in->n);
    HIPCHK(h, hipMemcpyAsync(q->d_in, q->host, sizeof(int) * RI_CTL_INTS + sizeof(float) * 2 * in->n, hipMemcpyHostToDevice, h->stream));
    rc = vg_fe_build_async(h, in->equalize);
    if (rc) return rc;
    FeDev d = s->d;
    d.npts = r.ctl + RI_N; d.prev_xy = r.xy_in;
    const bool track = in->n > 0;
    if (track) hipLaunchKernelGGL(fe_lk_kernel, dim3(in->n, 1), dim3(64), 0, h->stream, d);
    hipLaunchKernelGGL(fe_ri_after_lk_kernel, dim3(1), dim3(256), 0, h->stream, d, r);
    char* hA = q->host + q->in_bytes;
    char* hB = hA + q->a_bytes;
    int* horder = (int*)(hB + q->b_bytes);
    const int* ahdr = (const int*)hA;
    out->status_lk = (const uint8_t*)(hA + ((const char*)r.a_status_lk - q->d_a));
    out->status_f = (const uint8_t*)(hA + ((const char*)r.a_status_f - q->d_a));
    out->forw_xy = (const float*)(hA + ((const char*)r.a_forw_xy - q->d_a));
    if (!in->publish) {
        HIPCHK(h, hipGetLastError());
        HIPCHK(h, hipMemcpyAsync(hA, q->d_a, q->a_bytes, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        out->n1 = out->n2 = out->n_final = ahdr[RI_N1];
        out->un_xy = (const float*)(hA + ((const char*)r.a_un_xy - q->d_a));
        out->ransac_best = -1;
        return VG_OK;
    }
    // ---- rejectWithF (:169-202); the kernels leave at once when fewer than 15 points survived the tracking
    if (in->n >= 15) {
        const float thresh2 = (float)(in->f_threshold * in->f_threshold);
        hipLaunchKernelGGL(fe_ransac7_kernel, dim3((FE_RANSAC_MAXIT + 6) / 7), dim3(64), 0, h->stream, (const float*)r.p1, (const float*)r.p2, 0,
                           (const int*)q->d_sched, FE_RANSAC_MAXIT, rb.models, r.ctl);
        hipLaunchKernelGGL(fe_ransac_count_kernel, dim3(FE_RANSAC_MAXIT), dim3(64), 0, h->stream, (const float*)r.p1, (const float*)r.p2, 0, thresh2, 0,
                           (const double*)rb.models, FE_RANSAC_MAXIT, rb.F, rb.cnt, rb.med, rb.words, (const int*)r.ctl);
        hipLaunchKernelGGL(fe_ri_pick_kernel, dim3(1), dim3(256), 0, h->stream, r);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(hA, q->d_a, (size_t)((const char*)r.a_un_xy - q->d_a), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    out->n1 = ahdr[RI_N1]; out->n2 = ahdr[RI_N2]; out->ransac_ran = ahdr[RI_RANSAC]; out->fallback = ahdr[RI_FALLBACK];
    out->ransac_best = ahdr[RI_BEST]; out->ransac_niters = ahdr[RI_NITERS];
    if (out->n1 < 0 || out->n1 > in->n || out->n2 < 0 || out->n2 > out->n1) { h->err = "vg_fe_read_image: inconsistent counts from the device"; return VG_ERR_NUMERIC; }
    if (out->fallback) {
        // The device could not finish the estimate by itself (LMedS range, or a sample OpenCV would have redrawn): the lifted point
        // sets come back, vg_fe_reject_with_f runs the exact schedule, and the survivor list goes up again.
        const int n1 = out->n1;
        std::vector<float> pp((size_t)4 * n1);
        HIPCHK(h, hipMemcpyAsync(pp.data(), r.p1, sizeof(float) * 2 * n1, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(pp.data() + 2 * n1, r.p2, sizeof(float) * 2 * n1, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        uint8_t* sf = (uint8_t*)(hA + ((const char*)r.a_status_f - q->d_a));
        rc = vg_fe_reject_with_f(h, pp.data(), pp.data() + 2 * n1, n1, in->f_threshold, sf, nullptr, nullptr);
        if (rc) return rc;
        std::vector<int> idx2;
        for (int i = 0, k = 0; i < in->n; ++i)
            if (out->status_lk[i]) { if (sf[k]) idx2.push_back(i); ++k; }
        out->n2 = (int)idx2.size();
        if (out->n2) HIPCHK(h, hipMemcpy(r.idx2, idx2.data(), sizeof(int) * idx2.size(), hipMemcpyHostToDevice));
        HIPCHK(h, hipMemcpy(r.ctl + RI_N2, &out->n2, sizeof(int), hipMemcpyHostToDevice));
        out->ransac_best = -1; out->ransac_niters = 0;
    }
    // ---- setMask (:36-69): the order of the walk is the caller's (see include/vinsgpu.h)
    const int n2 = out->n2;
    if (in->order && n2 > 0) {
        for (int k = 0; k < n2; ++k) horder[k] = -1;
        if (in->order(in->user, out, horder) != 0) { h->err = "vg_fe_read_image: the order callback failed"; return VG_ERR_BAD_ARG; }
        std::vector<char> seen((size_t)n2, 0);
        for (int k = 0; k < n2; ++k) {
            if (horder[k] < 0 || horder[k] >= n2 || seen[horder[k]]) { h->err = "vg_fe_read_image: the order callback did not return a permutation"; return VG_ERR_BAD_ARG; }
            seen[horder[k]] = 1;
        }
        HIPCHK(h, hipMemcpyAsync(q->d_order, horder, sizeof(int) * n2, hipMemcpyHostToDevice, h->stream));
        r.order = q->d_order;
    } else
        r.order = nullptr;
    if (r.base_mask) HIPCHK(h, hipMemcpyAsync(s->mask, q->d_base, npix, hipMemcpyDeviceToDevice, h->stream));
    else HIPCHK(h, hipMemsetAsync(s->mask, 255, npix, h->stream));
    hipLaunchKernelGGL(fe_ri_setmask_kernel, dim3(1), dim3(64), 0, h->stream, d, r);
    if (n2 > 0) hipLaunchKernelGGL(fe_stamp_kernel, dim3(n2, 1), dim3(256), 0, h->stream, d, (const int*)(r.ctl + RI_NK), (const int*)r.kept_xy, in->min_dist);
    // ---- goodFeaturesToTrack(forw_img, n_pts, MAX_CNT - forw_pts.size(), 0.01, MIN_DIST, mask) (:144-149) + addPoints + undistortedPoints
    rc = vg_fe_detect_async(h, in->quality, (double)in->min_dist);
    if (rc) return rc;
    hipLaunchKernelGGL(fe_ri_finish_kernel, dim3(1), dim3(256), 0, h->stream, d, r);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(hB, q->d_b, q->b_bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int* bhdr = (const int*)hB;
    if (bhdr[RI_NNEW] < 0) { h->err = "goodFeaturesToTrack: candidate list overflow (more 3x3 maxima than the key buffer holds)"; return VG_ERR_UNSUPPORTED; }
    out->n_kept = bhdr[RI_NK]; out->n_new = bhdr[RI_NNEW]; out->n_final = out->n_kept + out->n_new;
    out->kept = (const int*)(hB + ((const char*)r.b_kept - q->d_b));
    out->new_xy = (const float*)(hB + ((const char*)r.b_new_xy - q->d_b));
    out->un_xy = (const float*)(hB + ((const char*)r.b_un_xy - q->d_b));
    return VG_OK;
}
