// fe_host.hip — host side of the front-end path behind the C-ABI (include/vinsgpu.h, vg_fe_*): owns the per-camera
// pyramids / point / corner buffers in HBM, launches the kernels of fe_kernels.hip on the handle's stream.
// Replaces the OpenCV calls of FeatureTracker::readImage (feature_tracker/src/feature_tracker.cpp:87-93, :113, :149).
#include "vg_range.h"
#include <hip/hip_runtime.h>
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

extern "C" int vg_fe_upload_frames(vg_handle* h, const uint8_t* const* imgs, int stride) {
    VG_RANGE("vg_fe_upload_frames");
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
