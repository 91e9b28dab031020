// fe_kernels.hip — feature_tracker front end on gfx950: the arithmetic under FeatureTracker::readImage
// (feature_tracker/src/feature_tracker.cpp:81-167), i.e. what the reference delegates to OpenCV:
//   CLAHE (call site :87-93), calcOpticalFlowPyrLK (:113: pyrDown pyramid, Scharr derivative, LK iterations),
//   goodFeaturesToTrack (:149: Sobel min-eigenvalue map, threshold + 3x3 NMS, sort, min-distance selection).
// OpenCV is third-party and absent: algorithms restated as recorded in oracle/ASSUMPTIONS.md.
//
// All integer paths are bit-exact by construction; the float32 paths evaluate the same IEEE expressions in the same
// order as oracle/fe_cpu.cpp (this file is compiled with -ffp-contract=off and correctly rounded sqrt / divide), so
// corner lists, LK status AND positions are bit-identical to the oracle.
//
// Layout: one image plane per (camera, pyramid level), row-major u8, rows contiguous (coalesced row segments);
// LK: ONE WAVEFRONT PER TRACK (block = 64 threads); the 24x24 template neighbourhood and a 32x32 search region live in LDS
// (1.6 KB per track), a lane's seven template values / gradients in registers (the Scharr field is formed there, round 4).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "vg_target.h"
#include "fe_layout.h"

#define FDEV __device__ __forceinline__

FDEV int reflect101(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }
// exact product of two integers known to fit 24 bits (pixels, 14-bit bilinear weights, int16 derivatives, 14-bit differences)
// whose product fits 32: v_mul_i32_i24 runs at full rate, v_mul_lo_u32 at a quarter of it -- and the LK kernel is bound by
// VALU issue
FDEV int mul24(int a, int b) { return vg_mul24(a, b); }
FDEV int cv_round(float v) { return __float2int_rn(v); }
FDEV int cv_floor(float v) { return (int)floorf(v); }
// monotone map float -> unsigned (total order incl. negatives)
FDEV unsigned ford(float f) { const unsigned b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
FDEV float funord(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// ================================================================================================ CLAHE
// grid (64 tiles, cams), block 256: per-tile histogram -> clip / redistribute -> LUT (clahe.cpp CLAHE_CalcLut_Body)
extern "C" __global__ __launch_bounds__(256) void fe_clahe_lut_kernel(FeDev d, int clip_limit, float lut_scale) {
    __shared__ int hist[256];
    const int cam = blockIdx.y, tile = blockIdx.x, tx = tile % 8, ty = tile / 8;
    const int tw = d.W / 8, th = d.H / 8;
    const uint8_t* src = d.raw + (size_t)cam * d.W * d.H;
    hist[threadIdx.x] = 0;
    __syncthreads();
    // thread = (tile row, quarter of the row): no division per pixel.  Round 6: the quarter row (<= 32 bytes for tiles up to 128 wide)
    // is REQUESTED in one batch -- eight dwords from a 2-byte aligned address, clamped to the row -- before the first histogram
    // update; the loop "load two pixels, add them" waited for every load (twelve dependent HBM round trips per thread: most of the
    // kernel).  Wider tiles keep the loop.
    {
        const glb_u8* tsrc = (const glb_u8*)src + (size_t)(ty * th) * d.W + tx * tw;
        const int seg = threadIdx.x & 3, qw = (tw + 3) >> 2;                 // columns [seg * qw, min(tw, (seg + 1) * qw))
        const int c0 = seg * qw, c1 = (c0 + qw) < tw ? (c0 + qw) : tw;
        for (int y = threadIdx.x >> 2; y < th; y += 64) {
            const glb_u8* row = tsrc + (size_t)y * d.W;
            if (((tw | qw) & 1) == 0 && qw <= 32 && (d.W & 1) == 0) {
                // dword k of the quarter covers columns c0 + 4 k .. c0 + 4 k + 3; a dword that would reach past the END OF THE IMAGE ROW is
                // read two bytes earlier and shifted (the row start is 2-byte aligned, c1 - c0 is even)
                unsigned v[8];
                const int rowend = d.W - tx * tw;                             // columns of this tile row left in the image row
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int x = c0 + 4 * k;
                    v[k] = 0;
                    if (x < c1) {
                        if (x + 4 <= rowend) v[k] = *(glb_u32_a2*)(row + x);
                        else v[k] = (unsigned)(*(const glb_u16*)(row + x));     // (two pixels left in the image row)
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int x = c0 + 4 * k;
                    if (x < c1) {
                        atomicAdd(&hist[v[k] & 255u], 1);
                        atomicAdd(&hist[(v[k] >> 8) & 255u], 1);
                        if (x + 2 < c1) {
                            atomicAdd(&hist[(v[k] >> 16) & 255u], 1);
                            atomicAdd(&hist[v[k] >> 24], 1);
                        }
                    }
                }
            } else if (((tw | qw) & 1) == 0) {
                for (int x = c0; x < c1; x += 2) {
                    const unsigned v = *(const glb_u16*)(row + x);
                    atomicAdd(&hist[v & 255u], 1);
                    atomicAdd(&hist[v >> 8], 1);
                }
            } else {
                for (int x = c0; x < c1; ++x) atomicAdd(&hist[row[x]], 1);
            }
        }
    }
    __syncthreads();
    // clipped excess: its sum over the 256 bins, then the inclusive scan of the redistributed histogram -- both inside the wavefronts
    // (shuffles) with one exchange of the four wavefront totals through LDS each (round 4: they were two LDS trees with a workgroup
    // barrier per step, ~25 barriers per tile; integer sums, any order)
    __shared__ int wtot[4], wsum[4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int hv = hist[threadIdx.x];
    int ex = hv > clip_limit ? hv - clip_limit : 0;
    hv = hv > clip_limit ? clip_limit : hv;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ex += __shfl_xor(ex, o, 64);
    if (lane == 0) wtot[wv] = ex;
    __syncthreads();
    const int clipped = (wtot[0] + wtot[1]) + (wtot[2] + wtot[3]);
    const int batch = clipped / 256;
    const int residual = clipped - batch * 256;
    hv += batch;
    if (residual != 0) {
        const int step = 256 / residual > 1 ? 256 / residual : 1;
        // for (i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++
        if ((int)threadIdx.x % step == 0 && (int)threadIdx.x / step < residual) hv += 1;
    }
    // inclusive scan -> LUT
    int sc = hv;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(sc, o, 64); sc += lane >= o ? v : 0; }
    if (lane == 63) wsum[wv] = sc;
    __syncthreads();
    for (int q = 0; q < wv; ++q) sc += wsum[q];
    int lv = cv_round((float)sc * lut_scale);
    lv = lv < 0 ? 0 : (lv > 255 ? 255 : lv);
    d.lut[((size_t)cam * 64 + tile) * 256 + threadIdx.x] = (uint8_t)lv;
}

// per-pixel bilinear blend of the four tile LUTs (CLAHE_Interpolation_Body); writes pyramid level 0 of `cur`.
// Round 4: one workgroup per INTERPOLATION CELL -- the pixels whose four LUTs are the same: (tx1, ty1) = (floor(x / tw - 0.5),
// floor(y / th - 0.5)) before clamping, 9 x 9 cells per frame.  The cell's four LUTs sit in LDS interleaved (one 4-byte read per pixel
// gives all four entries of its grey value; they were four byte gathers from HBM per pixel: 260 us per 256 frames at 0.7 TB/s), a
// thread handles four adjacent pixels (dword load / store).  A pixel belongs to the cell its OWN float expressions name, not to an
// integer range: the workgroup walks a range one dword wider than the cell and every pixel tests (tx1, ty1) == (cx - 1, cy - 1), so
// each pixel is written by exactly one workgroup and the arithmetic is the former kernel's, bit for bit.
extern "C" __global__ __launch_bounds__(256) void fe_clahe_apply_kernel(FeDev d, uint8_t* const* dst_planes) {
    __shared__ uint32_t lut4[256];
    const int cam = blockIdx.z, cx = blockIdx.x, cy = blockIdx.y;           // cell: tx1 = cx - 1, ty1 = cy - 1 (unclamped)
    const int W = d.W, H = d.H, tw = W / 8, th = H / 8;
    {
        const int tx1 = cx - 1 < 0 ? 0 : cx - 1, tx2 = cx > 7 ? 7 : cx, ty1 = cy - 1 < 0 ? 0 : cy - 1, ty2 = cy > 7 ? 7 : cy;
        const uint8_t* lut = d.lut + (size_t)cam * 64 * 256;
        const int v = threadIdx.x;
        lut4[v] = (uint32_t)lut[(ty1 * 8 + tx1) * 256 + v] | ((uint32_t)lut[(ty1 * 8 + tx2) * 256 + v] << 8) |
                  ((uint32_t)lut[(ty2 * 8 + tx1) * 256 + v] << 16) | ((uint32_t)lut[(ty2 * 8 + tx2) * 256 + v] << 24);
    }
    __syncthreads();
    // pixel ranges that certainly contain the cell ((cx - 0.5) tw <= x < (cx + 0.5) tw up to float rounding), x on dwords
    int x_lo = ((2 * cx - 1) * tw) / 2 - 2, x_hi = ((2 * cx + 1) * tw + 1) / 2 + 2;
    int y_lo = ((2 * cy - 1) * th) / 2 - 2, y_hi = ((2 * cy + 1) * th + 1) / 2 + 2;
    x_lo = (x_lo < 0 ? 0 : x_lo) & ~3; x_hi = x_hi > W ? W : x_hi;
    y_lo = y_lo < 0 ? 0 : y_lo; y_hi = y_hi > H ? H : y_hi;
    const int ngx = (x_hi - x_lo + 3) >> 2;                                // dwords per row of the walk (<= W / 32 + 2)
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    const glb_u8* src = (const glb_u8*)(d.raw + (size_t)cam * W * H);
    glb_u8* dst = (glb_u8*)dst_planes[cam];
    // thread = (dword column g, row r0 of a pass): its four x positions are the same in every row it visits, so what depends on x
    // alone -- cell membership and the horizontal weights -- is formed once
    const int rows_per_pass = 256 / ngx, g = threadIdx.x % ngx, r0 = threadIdx.x / ngx;
    if (r0 >= rows_per_pass) return;
    const int x = x_lo + 4 * g;                                           // (W % 8 == 0: the dword lies inside the row)
    float xa[4], xa1[4];
    unsigned mine = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float txf = (x + q) * inv_tw - 0.5f;
        const int tx1 = cv_floor(txf);
        xa[q] = txf - tx1; xa1[q] = 1.0f - xa[q];
        if (tx1 == cx - 1) mine |= 1u << q;
    }
    if (mine == 0u) return;
    // (round 6: the dwords of up to eight of the thread's rows are requested together -- the loop waited for every row's load before it
    //  asked for the next one: ~7 dependent HBM round trips per thread)
    for (int yb = y_lo + r0; yb < y_hi; yb += 8 * rows_per_pass) {
        uint32_t in8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int y = yb + u * rows_per_pass;
            in8[u] = *(const glb_u32*)(src + (size_t)(y < y_hi ? y : yb) * W + x);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int y = yb + u * rows_per_pass;
            if (y >= y_hi) continue;
            const float tyf = y * inv_th - 0.5f;
            const int ty1 = cv_floor(tyf);
            if (ty1 != cy - 1) continue;
            const float ya = tyf - ty1, ya1 = 1.0f - ya;
            const uint32_t in4 = in8[u];
            uint32_t out4 = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t l = lut4[(in4 >> (8 * q)) & 255u];
                const float l11 = (float)(l & 255u), l12 = (float)((l >> 8) & 255u), l21 = (float)((l >> 16) & 255u), l22 = (float)(l >> 24);
                const float res = (l11 * xa1[q] + l12 * xa[q]) * ya1 + (l21 * xa1[q] + l22 * xa[q]) * ya;
                int o = cv_round(res);
                o = o < 0 ? 0 : (o > 255 ? 255 : o);
                out4 |= (uint32_t)o << (8 * q);
            }
            glb_u8* po = dst + (size_t)y * W + x;
            if (mine == 15u) *(glb_u32*)po = out4;
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (mine & (1u << q)) po[q] = (uint8_t)(out4 >> (8 * q));
            }
        }
    }
}

extern "C" __global__ __launch_bounds__(256) void fe_copy_kernel(FeDev d, uint8_t* const* dst_planes) {
    const int cam = blockIdx.y;
    const size_t n = (size_t)d.W * d.H;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n / 4; k += (size_t)gridDim.x * 256)
        ((uint32_t*)dst_planes[cam])[k] = ((const uint32_t*)(d.raw + (size_t)cam * n))[k];
}

// ================================================================================================ pyrDown
// [1 4 6 4 1] x [1 4 6 4 1], BORDER_REFLECT_101, (sum + 128) >> 8  (pyramids.cpp).  Block = 64 x 16 output tile: the 35 x 136
// input tile is staged in LDS (aligned dwords wherever the dword lies inside the image row, per-byte reflection for the one or
// two dwords that straddle an edge), filtered horizontally once (35 x 64 partial sums) and then vertically: ~3x fewer
// instructions than 25 reflected global byte loads per output pixel, same integers.
// Round 4: the kernel was bound by VALU issue, not by HBM (~370 instructions per thread for 4 outputs): the horizontal pass now
// forms FOUR adjacent sums from one 16-byte LDS read (their 11 input bytes start at a multiple of 8: 6 instead of 18
// instructions per sum), the staging loop walks (row, dword) without a division or 64-bit address arithmetic per element, and
// border tiles (42 % of the tiles of level 0, 75 % of level 1, all of level 2) no longer stage their whole input per byte.
#define PD_ROWS 16                              // output rows per workgroup (64 x 16 outputs = 4 per thread)
#define PD_IN (2 * PD_ROWS + 3)                 // input rows a workgroup reads
#define PD_TW 144                               // bytes per staged row (136 used: input columns 2 bx0 - 4 .. 2 bx0 + 131)
extern "C" __global__ __launch_bounds__(256) void fe_pyrdown_tile_kernel(const uint8_t* const* src_planes, uint8_t* const* dst_planes, int sw, int sh) {
    __shared__ alignas(16) uint8_t tile[PD_IN][PD_TW];
    __shared__ alignas(16) int hs[PD_IN][64];
    const int cam = blockIdx.z;
    const int dw = (sw + 1) / 2, dh = (sh + 1) / 2;
    const int bx0 = blockIdx.x * 64, by0 = blockIdx.y * PD_ROWS;
    const glb_u8* s = (const glb_u8*)src_planes[cam];     // (the planes are HBM: global_load, not flat_load)
    const int ix0 = 2 * bx0 - 4, iy0 = 2 * by0 - 2;         // input coordinates of tile[0][0] (ix0 is 4-byte aligned)
    // Columns / rows further out than one reflection are never used by an output inside the image; they are clamped so that the
    // read stays inside the plane, and what lies beyond sw + 4 / sh + 4 is not read at all.
    if ((sw & 3) == 0) {
        const int r0 = (int)threadIdx.x / 34, cdw = (int)threadIdx.x - 34 * r0;      // 7 rows x 34 dwords per pass
        if (r0 < 7) {
            const int gx = ix0 + 4 * cdw;
            const bool whole = gx >= 0 && gx + 4 <= sw, skipx = gx >= sw + 4;
            // all five loads of a thread are issued before the first LDS store (a load -> store loop is five dependent HBM round
            // trips per workgroup)
            uint32_t v[5];
#pragma unroll
            for (int u = 0; u < 5; ++u) {
                const int r = r0 + 7 * u;
                int ry = reflect101(iy0 + r, sh);
                ry = ry < 0 ? 0 : (ry >= sh ? sh - 1 : ry);
                const glb_u8* row = s + (unsigned)(ry * sw);
                v[u] = 0;
                if (whole) v[u] = *(const glb_u32*)(row + gx);
                else if (!skipx && iy0 + r < sh + 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        int rx = reflect101(gx + q, sw);
                        rx = rx < 0 ? 0 : (rx >= sw ? sw - 1 : rx);
                        v[u] |= (uint32_t)row[rx] << (8 * q);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 5; ++u) *(uint32_t*)&tile[r0 + 7 * u][4 * cdw] = v[u];
        }
    } else {
        for (int k = threadIdx.x; k < PD_IN * 136; k += 256) {
            const int r = k / 136, cc = k - 136 * r;
            int ry = reflect101(iy0 + r, sh), rx = reflect101(ix0 + cc, sw);
            ry = ry < 0 ? 0 : (ry >= sh ? sh - 1 : ry);
            rx = rx < 0 ? 0 : (rx >= sw ? sw - 1 : rx);
            tile[r][cc] = s[(size_t)ry * sw + rx];
        }
    }
    __syncthreads();
    // horizontal: thread = (row, 4 adjacent sums).  Sum lx reads tile columns 2 lx + 2 .. 2 lx + 6: for lx = 4 j .. 4 j + 3 the
    // bytes 8 j + 2 .. 8 j + 12 of the row
    {
        const int j = threadIdx.x & 15;
        for (int r = threadIdx.x >> 4; r < PD_IN; r += 16) {
            const uint2 lo = *(const uint2*)&tile[r][8 * j], hi = *(const uint2*)&tile[r][8 * j + 8];
            const uint32_t w[4] = {lo.x, lo.y, hi.x, hi.y};
            int4 o;
            int* po = &o.x;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int acc = 0;
                const int wgt[5] = {1, 4, 6, 4, 1};
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const int byte = 2 + 2 * i + t;
                    acc += wgt[t] * (int)((w[byte >> 2] >> (8 * (byte & 3))) & 255u);
                }
                po[i] = acc;
            }
            *(int4*)&hs[r][4 * j] = o;
        }
    }
    __syncthreads();
    // vertical: thread = 4 horizontally adjacent outputs of one row: five 16-byte LDS reads, one 4-byte store (a byte store per
    // thread moves 64 bytes per wavefront instruction)
    const int lq = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int x = bx0 + 4 * lq, y = by0 + ly;
    if (x >= dw || y >= dh) return;
    int acc[4] = {0, 0, 0, 0};
    const int wgt[5] = {1, 4, 6, 4, 1};
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int4 v = *(const int4*)&hs[2 * ly + r][4 * lq];
        acc[0] += wgt[r] * v.x; acc[1] += wgt[r] * v.y; acc[2] += wgt[r] * v.z; acc[3] += wgt[r] * v.w;
    }
    uint8_t* o = dst_planes[cam] + (size_t)y * dw + x;
    if (x + 3 < dw && (dw & 3) == 0) {
        *(uint32_t*)o = (uint32_t)((acc[0] + 128) >> 8) | ((uint32_t)((acc[1] + 128) >> 8) << 8) | ((uint32_t)((acc[2] + 128) >> 8) << 16) |
                        ((uint32_t)((acc[3] + 128) >> 8) << 24);
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (x + q < dw) o[q] = (uint8_t)((acc[q] + 128) >> 8);
    }
}

// The same filter for rows that start on a dword (sw % 4 == 0: every level of a 752-wide frame) WITHOUT LDS and without barriers
// (round 4; the tile kernel above stays for other widths): thread = 4 adjacent output columns x PD_R output rows.  An input row
// costs the thread ONE 16-byte load (input columns 8 tx - 4 .. 8 tx + 11; neighbouring threads overlap by half, which the L1 / TA
// absorbs) and its four horizontal sums in two registers of two 16-bit lanes (v_perm_b32 cuts the (column, column + 2) pairs out
// of the window, v_pk_add / v_pk_mad_u16 weigh them: 18 instructions per row, every partial sum < 2^16); the vertical filter runs
// over the thread's own 2 PD_R + 3 rows of sums, again on 16-bit lanes ((sum + 128) >> 8 <= 255: 13 instructions per 4 outputs).
// All loads of a thread are issued before the first use.  Edges: the row index is reflected per row (wave-uniform); the first
// thread of a row takes columns -2, -1 from columns 2, 1 of its own window, the last one column sw from column sw - 2 (one byte
// each: no output inside the image reads further out); the 4 bytes in front of a row / up to 12 behind it that the window of
// those threads covers are read and ignored (the planes are allocated with slack on both sides).
#define PD_R 8
#define PD_NIN (2 * PD_R + 3)
template <int A> FDEV unsigned pd_pair(const unsigned (&d)[4]) {          // window bytes A and A + 2 as two 16-bit lanes
    return vg_perm<(unsigned)(A & 3) | (0x0Cu << 8) | ((unsigned)((A & 3) + 2) << 16) | (0x0Cu << 24)>(d[(A >> 2) + 1 > 3 ? 3 : (A >> 2) + 1], d[A >> 2]);
}
template <int K> FDEV unsigned pd_hsum(const unsigned (&d)[4]) {           // [1 4 6 4 1] at window bytes K .. K + 4 and K + 2 .. K + 6
    const unsigned e = vg_pk_add(pd_pair<K>(d), pd_pair<K + 4>(d));
    const unsigned o = vg_pk_add(pd_pair<K + 1>(d), pd_pair<K + 3>(d));
    return vg_pk_mad(pd_pair<K + 2>(d), 6, vg_pk_mad(o, 4, e));
}
extern "C" __global__ __launch_bounds__(256) void fe_pyrdown_kernel(const uint8_t* const* src_planes, uint8_t* const* dst_planes, int sw, int sh,
                                                                    int waves_per_strip) {
    const int cam = blockIdx.z;
    const int dw = sw >> 1, dh = (sh + 1) >> 1;
    const int ntx = (dw + 3) >> 2;
    const int wave = uni((int)(blockIdx.x * 4 + (threadIdx.x >> 6))), lane = threadIdx.x & 63;
    const int strip = wave / waves_per_strip, tx = (wave - strip * waves_per_strip) * 64 + lane;
    const int y0 = strip * PD_R;
    if (y0 >= dh || tx >= ntx) return;
    const glb_u8* s = (const glb_u8*)src_planes[cam];
    const int c0 = 8 * tx - 4;                     // input column of window byte 0
    const int isw = sw - c0;                       // window index of column sw (8 or 12 in the last thread of a row, larger elsewhere)
    const bool left = tx == 0, fix8 = isw == 8, fix12 = isw == 12;
    uint4 w[PD_NIN];
#pragma unroll
    for (int r = 0; r < PD_NIN; ++r) {
        int ry = reflect101(2 * y0 - 2 + r, sh);
        ry = ry < 0 ? 0 : (ry >= sh ? sh - 1 : ry);
        w[r] = vg_load16_unaligned(s + (unsigned)(ry * sw) + c0);
    }
    unsigned h[PD_NIN][2];
#pragma unroll
    for (int r = 0; r < PD_NIN; ++r) {
        unsigned d[4] = {w[r].x, w[r].y, w[r].z, w[r].w};
        const unsigned l0 = vg_perm<0x05060100u>(d[1], d[0]);      // bytes 2, 3 <- window bytes 6, 5 (columns 2, 1)
        const unsigned r3 = vg_perm<0x07060502u>(d[3], d[2]);      // byte 12 <- window byte 10
        const unsigned r2 = vg_perm<0x07060502u>(d[2], d[1]);      // byte 8 <- window byte 6
        d[0] = left ? l0 : d[0];
        d[3] = fix12 ? r3 : d[3];
        d[2] = fix8 ? r2 : d[2];
        h[r][0] = pd_hsum<2>(d);
        h[r][1] = pd_hsum<6>(d);
    }
    glb_u8* dst = (glb_u8*)dst_planes[cam];
    const int x = 4 * tx;
#pragma unroll
    for (int i = 0; i < PD_R; ++i) {
        const int y = y0 + i;
        if (y >= dh) break;
        unsigned o[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const unsigned e = vg_pk_add(h[2 * i][m], h[2 * i + 4][m]);
            const unsigned od = vg_pk_add(h[2 * i + 1][m], h[2 * i + 3][m]);
            const unsigned a = vg_pk_mad(h[2 * i + 2][m], 6, vg_pk_mad(od, 4, e));
            o[m] = vg_pk_shr(vg_pk_add(a, 0x00800080u), 8);
        }
        const unsigned px = vg_perm<0x06040200u>(o[1], o[0]);     // the four results, one byte each
        glb_u8* q = dst + (unsigned)(y * dw + x);
        if (x + 3 < dw && (dw & 3) == 0) *(glb_u32*)q = px;
        else {
#pragma unroll
            for (int k = 0; k < 4; ++k) if (x + k < dw) q[k] = (uint8_t)(px >> (8 * k));
        }
    }
}

// ================================================================================================ LK
#define LK_WIN 21
#define LK_WBITS 14

#define LK_JR 32
#define LK_JS ((LK_JR - 22) / 2)
struct LkLds {
    alignas(16) uint8_t jreg[LK_JR * LK_JR];    // search REGION (reflect-101), origin (jox, joy): holds every 22x22 search window
                                    // whose origin lies within +-LK_JS px of the window it was staged for
    uint8_t ipatch[24 * 24 + 8];    // template neighbourhood (reflect-101), origin (ipx-1, ipy-1) (+ 8: a lane reads its rows as 12 bytes)
};

// Exact integer wavefront sums on the DPP network (VALU only; a __shfl_down tree is dependent LDS-crossbar round trips): quad
// swaps -> half-row mirror give every lane its 8-lane total.  Integer addition is associative, so the result is independent of
// the order.
// the same for lane values whose 8-lane totals still fit 32 bits (|v| < 2^28): three DPP stages in 32 bits (old = 0 + bound_ctrl
// lets the compiler fold each move into its add: one v_add_u32_dpp per stage instead of copy + v_mov_dpp + add), then the eight
// 8-lane totals through SGPRs — sign extension and the 64-bit adds are SALU work, which the VALU-bound kernel has spare
FDEV int dpp_add_i32_0xB1(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true); }
FDEV int dpp_add_i32_0x4E(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true); }
FDEV int dpp_add_i32_0x141(int v) { return v + __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true); }
FDEV long long wave_sum_i32(int v) {
    v = dpp_add_i32_0xB1(v);
    v = dpp_add_i32_0x4E(v);
    v = dpp_add_i32_0x141(v);
    long long t = 0;
#pragma unroll
    for (int g = 0; g < 8; ++g) t += (long long)__builtin_amdgcn_readlane(v, 8 * g);
    return t;
}
// lane values below 2^27 in magnitude: the 16-lane totals of a DPP row still fit 32 bits -- one more stage (row_mirror), four
// v_readlane instead of eight
FDEV long long wave_sum_i27(int v) {
    v = dpp_add_i32_0xB1(v);
    v = dpp_add_i32_0x4E(v);
    v = dpp_add_i32_0x141(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    long long t = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) t += (long long)__builtin_amdgcn_readlane(v, 16 * g);
    return t;
}
// lane values whose WAVEFRONT total fits 32 bits
FDEV int wave_sum_small(int v) {
    v = dpp_add_i32_0xB1(v);
    v = dpp_add_i32_0x4E(v);
    v = dpp_add_i32_0x141(v);
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
FDEV void lk_weights(float a, float b, int& w00, int& w01, int& w10, int& w11) {
    w00 = cv_round((1.f - a) * (1.f - b) * (float)(1 << LK_WBITS));
    w01 = cv_round(a * (1.f - b) * (float)(1 << LK_WBITS));
    w10 = cv_round((1.f - a) * b * (float)(1 << LK_WBITS));
    w11 = (1 << LK_WBITS) - w00 - w01 - w10;
}

// Image planes are reached through pointer tables in HBM: a loaded pointer is generic and its accesses would compile to
// flat_load (which also ties the LDS counter to the image loads); the planes are HBM, say so (glb_u8, vg_target.h).

// The bilinear taps of a lane's seven pixels from its 2 x 8 search-window bytes (row a = ra, row b = rb), round 6.  A tap is
//     t_q = w00 a_q + w01 a_(q+1) + w10 b_q + w11 b_(q+1) + 2^(W-1),   value = t_q >> W,   W = LK_WBITS - 5,
// and what the iteration needs is value - template.  Three things make it short (VALU issue bounds this kernel):
//  * the 16-bit pairs are cut by COLUMN, V_q = (a_q, b_q): tap q is V_q . (w00, w10) + V_(q+1) . (w01, w11), so a pair serves two taps
//    (eight v_perm_b32 instead of fourteen row pairs);
//  * the template rides in the addend: with c_q = 2^(W-1) - 2^W template_q the shift delivers value - template directly
//    (floor((x - 2^W i) / 2^W) = floor(x / 2^W) - i exactly; |c_q| < 2^23 and |t_q| < 2^22: no overflow), so the rounding constant
//    costs no v_mov and the difference no v_sub;
//  * integer sums are exact in any order: same values as the row-pair form bit for bit.
// Signed lanes: the fourth weight is 2^14 minus the three rounded ones and can be -1 or -2.
template <int Q> FDEV unsigned lk_col_pair(const uint2& ra, const uint2& rb) {
    // (a_Q | b_Q << 16): byte Q & 3 of the dword of row a and of row b
    return Q < 4 ? vg_perm<(unsigned)(Q & 3) | (0x0Cu << 8) | ((unsigned)(4 + (Q & 3)) << 16) | (0x0Cu << 24)>(rb.x, ra.x)
                 : vg_perm<(unsigned)(Q & 3) | (0x0Cu << 8) | ((unsigned)(4 + (Q & 3)) << 16) | (0x0Cu << 24)>(rb.y, ra.y);
}
// d[q] = (tap q >> W) - template q  for q = 0 .. 6;  cq[q] = 2^(W-1) - 2^W template q;  wl = (w00 | w10 << 16), wr = (w01 | w11 << 16)
FDEV void lk_diffs7(const uint2& ra, const uint2& rb, unsigned wl, unsigned wr, const int* cq, int* d) {
    const unsigned v0 = lk_col_pair<0>(ra, rb), v1 = lk_col_pair<1>(ra, rb), v2 = lk_col_pair<2>(ra, rb), v3 = lk_col_pair<3>(ra, rb);
    const unsigned v4 = lk_col_pair<4>(ra, rb), v5 = lk_col_pair<5>(ra, rb), v6 = lk_col_pair<6>(ra, rb), v7 = lk_col_pair<7>(ra, rb);
    d[0] = vg_sdot2(v1, wr, vg_sdot2_keep(v0, wl, cq[0])) >> (LK_WBITS - 5);
    d[1] = vg_sdot2(v2, wr, vg_sdot2_keep(v1, wl, cq[1])) >> (LK_WBITS - 5);
    d[2] = vg_sdot2(v3, wr, vg_sdot2_keep(v2, wl, cq[2])) >> (LK_WBITS - 5);
    d[3] = vg_sdot2(v4, wr, vg_sdot2_keep(v3, wl, cq[3])) >> (LK_WBITS - 5);
    d[4] = vg_sdot2(v5, wr, vg_sdot2_keep(v4, wl, cq[4])) >> (LK_WBITS - 5);
    d[5] = vg_sdot2(v6, wr, vg_sdot2_keep(v5, wl, cq[5])) >> (LK_WBITS - 5);
    d[6] = vg_sdot2(v7, wr, vg_sdot2_keep(v6, wl, cq[6])) >> (LK_WBITS - 5);
}
// two differences (|d| <= 255 * 32 fits 16 bits) as one register of two int16 lanes: one v_perm_b32
FDEV unsigned lk_pack16(int lo, int hi) { return vg_perm<0x05040100u>((unsigned)hi, (unsigned)lo); }

// Stage the LK_JR x LK_JR search region with origin (ox, oy) of plane J into LDS (reflect-101 outside the image, exactly
// the pixels the per-window gather of LKTrackerInvoker reads).  lane = (row, 16-byte half): one unaligned 16-byte load
// per lane when the region lies inside the image, per-byte reflection at the borders.
FDEV void lk_stage_region(uint8_t* reg, const glb_u8* J, int lw, int lh, int ox, int oy, int lane) {
    __syncthreads();
    const int row = lane >> 1, hf = lane & 1;
    const bool inside = ox >= 0 && ox + LK_JR <= lw && oy >= 0 && oy + LK_JR <= lh;      // uniform
    if (inside) {
        const glb_u8* pj = J + (size_t)(oy + row) * lw + ox + 16 * hf;      // unaligned 16 bytes
        const uint4 v = vg_load16_unaligned(pj);
        *(uint4*)(reg + row * LK_JR + 16 * hf) = v;
    } else {
        // (the +-LK_JS margin may reach more than one image size outside a tiny level; those pixels are never part of a
        //  window — a window origin is >= -LK_WIN — so they are clamped instead of folded twice)
        const int ry = reflect101(oy + row, lh);
        const glb_u8* src = J + (size_t)(ry < 0 ? 0 : (ry >= lh ? lh - 1 : ry)) * lw;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int rx = reflect101(ox + 16 * hf + q, lw);
            reg[row * LK_JR + 16 * hf + q] = src[rx < 0 ? 0 : (rx >= lw ? lw - 1 : rx)];
        }
    }
    __syncthreads();
}

// grid (max_points, cams), block 64 (one wavefront = one track).  LKTrackerInvoker, levels max_level .. 0.
// (six wavefronts per SIMD: the register allocator is asked for <= 80 VGPRs instead of the 86 it takes unasked -- 73, no spills;
//  -3 % per launch.  Eight would need <= 64 and was measured slower.)
VG_WAVES_PER_EU(6)
extern "C" __global__ __launch_bounds__(64) void fe_lk_kernel(FeDev d) {
    __shared__ LkLds s;
    const int cam = blockIdx.y, t = blockIdx.x, lane = threadIdx.x;
    if (t >= d.npts[cam]) return;
    const float px = d.prev_xy[((size_t)cam * d.max_pts + t) * 2], py = d.prev_xy[((size_t)cam * d.max_pts + t) * 2 + 1];
    float nx = 0.f, ny = 0.f, err = 0.f;
    int status = 1;
    const float half = (LK_WIN - 1) * 0.5f;
    const float FLT_SCALE = 1.f / (1 << 20);
    const int ly = lane / 3, x0 = 7 * (lane - 3 * ly);       // 21 rows x 3 segments of 7 pixels = 63 lanes
    const bool act = lane < 63;
    const int lyc = act ? ly : 0;
    for (int level = d.max_level; level >= 0; --level) {
        const int lw = d.lw[level], lh = d.lh[level];
        const glb_u8* I = (const glb_u8*)d.prev_planes[level * d.cams + cam];
        const glb_u8* J = (const glb_u8*)d.cur_planes[level * d.cams + cam];
        float prevx = px * (float)(1. / (1 << level)), prevy = py * (float)(1. / (1 << level));
        float nextx, nexty;
        if (level == d.max_level) { nextx = prevx; nexty = prevy; }
        else { nextx = nx * 2.f; nexty = ny * 2.f; }
        nx = nextx; ny = nexty;
        prevx -= half; prevy -= half;
        const int ipx = cv_floor(prevx), ipy = cv_floor(prevy);
        if (ipx < -LK_WIN || ipx >= lw || ipy < -LK_WIN || ipy >= lh) {
            if (level == 0) { status = 0; err = 0.f; }
            continue;
        }
        int w00, w01, w10, w11;
        lk_weights(prevx - ipx, prevy - ipy, w00, w01, w10, w11);
        __syncthreads();
        // 24x24 neighbourhood of the template window: 8-byte row chunks in the interior, per-byte reflect-101 at borders
        if (ipx - 1 >= 0 && ipx + 23 <= lw && ipy - 1 >= 0 && ipy + 23 <= lh) {              // uniform
            for (int k = lane; k < 24 * 3; k += 64) {
                const int yy = k / 3, ch = k - 3 * yy;
                uint2 v;
                __builtin_memcpy(&v, I + (size_t)(ipy - 1 + yy) * lw + (ipx - 1) + 8 * ch, 8);
                *(uint2*)(s.ipatch + yy * 24 + 8 * ch) = v;
            }
        } else {
            for (int k = lane; k < 24 * 24; k += 64) {
                const int yy = k / 24, xx = k % 24;
                s.ipatch[k] = I[(size_t)reflect101(ipy - 1 + yy, lh) * lw + reflect101(ipx - 1 + xx, lw)];
            }
        }
        // (the patch was written by other lanes of this wavefront: LDS operations of one wavefront complete in order on the
        //  hardware; the wave barrier states the dependency for the compiler and for the CPU emulation of tests/simt)
        __builtin_amdgcn_wave_barrier();
        // lane = (window row ly, 7-pixel segment): the template values and gradients of a lane's seven pixels stay in registers
        // for all iterations of the level (VALU issue, not latency, bounds this kernel at full occupancy).
        // Round 4: the lane forms the Scharr derivatives (calcSharrDeriv; constant-0 border outside the image; NB the x+-1 / y+-1
        // taps reflect at the IMAGE edge, which is what the reflect-101 patch holds) of ITS 2 x 8 lattice positions from its 4 x 10
        // patch bytes in registers -- column sums t0 = 3 (up + down) + 10 mid and t1 = down - up are shared by neighbouring
        // positions (9 instructions per position) -- instead of a 22 x 22 field through LDS (18 per position on 44 lanes, then 16
        // packed dword reads and 8 unpacks per pixel): one barrier and 1.9 KB of LDS less, -2.5 % per launch.
        // (a lane's seven products of two gradients are < 7 * 4080^2 = 1.2e8 < 2^27: lane sums in 32 bits, see wave_sum_i32)
        int s11 = 0, s12 = 0, s22 = 0;
        int cq[7];                    // 2^(W-1) - 2^W template value (the addend of a tap, lk_diffs7)
        unsigned gxp[3], gyp[3];      // the gradients of pixels (2 j, 2 j + 1) as two int16 lanes (|Ix|, |Iy| <= 16 * 255)
        int gx6, gy6;
        {
            // The lane's 4 x 10 patch bytes (rows ly .. ly + 3, columns x0 .. x0 + 9) as three dwords per row; everything below
            // works on TWO 16-bit lanes per register (two neighbouring columns; v_perm_b32 cuts the pairs out, v_pk_add / v_pk_sub /
            // v_pk_mad_u16 do the Scharr arithmetic modulo 2^16: |values| <= 4080).
            const uint8_t* pp = s.ipatch + lyc * 24 + x0;
            unsigned R[4][3];
#pragma unroll
            for (int r = 0; r < 4; ++r) __builtin_memcpy(R[r], pp + r * 24, 12);
            // E[r][j] = columns (2 j, 2 j + 1) of patch row r
            unsigned E[4][5];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                E[r][0] = vg_perm<0x0C010C00u>(R[r][0], R[r][0]); E[r][1] = vg_perm<0x0C030C02u>(R[r][0], R[r][0]);
                E[r][2] = vg_perm<0x0C010C00u>(R[r][1], R[r][1]); E[r][3] = vg_perm<0x0C030C02u>(R[r][1], R[r][1]);
                E[r][4] = vg_perm<0x0C010C00u>(R[r][2], R[r][2]);
            }
            // lattice position (k, c), k = 0, 1, c = 0 .. 7  <->  patch (k + 1, c + 1)  <->  image (ipy + ly + k, ipx + x0 + c);
            // DX[k][j] / DY[k][j] = the derivatives at (k, 2 j) and (k, 2 j + 1):  dx(c) = t0(c + 2) - t0(c),
            // dy(c) = 3 (t1(c) + t1(c + 2)) + 10 t1(c + 1)  with the column sums t0 = 3 (up + down) + 10 mid, t1 = down - up
            unsigned DX[2][4], DY[2][4];
            const bool allin = ipx >= 0 && ipx + 22 <= lw && ipy >= 0 && ipy + 22 <= lh;       // uniform: no position outside
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                unsigned T0[5], T1[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    T0[j] = vg_pk_mad(vg_pk_add(E[k][j], E[k + 2][j]), 3, vg_pk_mad(E[k + 1][j], 10, 0u));
                    T1[j] = vg_pk_sub(E[k + 2][j], E[k][j]);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    DX[k][j] = vg_pk_sub(T0[j + 1], T0[j]);
                    const unsigned mid = vg_perm<0x05040302u>(T1[j + 1], T1[j]);              // (t1(2 j + 1), t1(2 j + 2))
                    DY[k][j] = vg_pk_mad(vg_pk_add(T1[j], T1[j + 1]), 3, vg_pk_mad(mid, 10, 0u));
                }
            }
            if (!allin) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int Y = ipy + lyc + k;
                    const bool yin = Y >= 0 && Y < lh;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int X = ipx + x0 + 2 * j;
                        const unsigned keep = ((yin && X >= 0 && X < lw) ? 0x0000ffffu : 0u) | ((yin && X + 1 >= 0 && X + 1 < lw) ? 0xffff0000u : 0u);
                        DX[k][j] &= keep; DY[k][j] &= keep;
                    }
                }
            }
            // bilinear taps as signed 16-bit dot products (weights in [-2, 2^14]: the fourth is 2^14 minus the three rounded ones;
            // pixels < 2^8, derivatives |.| <= 4080): two v_dot2_i32_i16 per tap, the rounding constant as the addend of the first
            // (three-operand form: one register holds the constant for all taps).  Gradients: ROW pairs (q, q + 1) -- a register as
            // it stands for even q, one v_perm_b32 of two neighbouring registers for odd q -- against (w00, w01) / (w10, w11);
            // template pixels: COLUMN pairs (row k = 0, row k = 1) of patch columns 1 .. 8 against (w00, w10) / (w01, w11), a pair
            // serves two taps (lk_col_pair).  The idle lane 63 gets zero gradient weights: its gradients are (2^13 >> 14) = 0.
            // The values fit 16 bits (|gradient tap| <= 4080, template tap <= 255 * 32), as OpenCV's short deriv / patch buffers hold them.
            const unsigned wtg = act ? ((unsigned)w00 | ((unsigned)w01 << 16)) : 0u, wbg = act ? ((unsigned)w10 | ((unsigned)w11 << 16)) : 0u;
            const unsigned wl = (unsigned)w00 | ((unsigned)w10 << 16), wr = (unsigned)w01 | ((unsigned)w11 << 16);
            const int rnd_g = 1 << (LK_WBITS - 1), rnd_i = 1 << (LK_WBITS - 5 - 1);
            unsigned PX[2][7], PY[2][7];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int q = 0; q < 7; ++q) {
                    PX[k][q] = (q & 1) ? vg_perm<0x05040302u>(DX[k][(q + 1) / 2], DX[k][q / 2]) : DX[k][q / 2];
                    PY[k][q] = (q & 1) ? vg_perm<0x05040302u>(DY[k][(q + 1) / 2], DY[k][q / 2]) : DY[k][q / 2];
                }
            }
            // VI[c - 1] = (patch row 1 byte c | patch row 2 byte c << 16), c = 1 .. 8: byte c & 3 of dword c >> 2 of either row
            unsigned VI[8];
            VI[0] = vg_perm<0x0C050C01u>(R[2][0], R[1][0]); VI[1] = vg_perm<0x0C060C02u>(R[2][0], R[1][0]); VI[2] = vg_perm<0x0C070C03u>(R[2][0], R[1][0]);
            VI[3] = vg_perm<0x0C040C00u>(R[2][1], R[1][1]); VI[4] = vg_perm<0x0C050C01u>(R[2][1], R[1][1]); VI[5] = vg_perm<0x0C060C02u>(R[2][1], R[1][1]);
            VI[6] = vg_perm<0x0C070C03u>(R[2][1], R[1][1]); VI[7] = vg_perm<0x0C040C00u>(R[2][2], R[1][2]);
            int ixv[7], iyv[7];
#pragma unroll
            for (int q = 0; q < 7; ++q) {
                const int ival = vg_sdot2(VI[q + 1], wr, vg_sdot2_keep(VI[q], wl, rnd_i)) >> (LK_WBITS - 5);
                ixv[q] = vg_sdot2(PX[1][q], wbg, vg_sdot2_keep(PX[0][q], wtg, rnd_g)) >> LK_WBITS;
                iyv[q] = vg_sdot2(PY[1][q], wbg, vg_sdot2_keep(PY[0][q], wtg, rnd_g)) >> LK_WBITS;
                cq[q] = rnd_i - (1 << (LK_WBITS - 5)) * ival;
            }
#pragma unroll
            for (int j = 0; j < 3; ++j) { gxp[j] = lk_pack16(ixv[2 * j], ixv[2 * j + 1]); gyp[j] = lk_pack16(iyv[2 * j], iyv[2 * j + 1]); }
            gx6 = ixv[6]; gy6 = iyv[6];
            // sums of gradient products over the lane's seven pixels: two pixels per v_dot2_i32_i16 (< 7 * 4080^2 = 1.2e8 < 2^27)
            s11 = mul24(gx6, gx6); s12 = mul24(gx6, gy6); s22 = mul24(gy6, gy6);
#pragma unroll
            for (int j = 0; j < 3; ++j) { s11 = vg_sdot2(gxp[j], gxp[j], s11); s12 = vg_sdot2(gxp[j], gyp[j], s12); s22 = vg_sdot2(gyp[j], gyp[j], s22); }
        }
        const long long a11 = wave_sum_i27(s11), a12 = wave_sum_i27(s12), a22 = wave_sum_i27(s22);
        const float A11 = (float)a11 * FLT_SCALE, A12 = (float)a12 * FLT_SCALE, A22 = (float)a22 * FLT_SCALE;
        float D = A11 * A22 - A12 * A12;
        const float minEig = (A22 + A11 - sqrtf((A11 - A22) * (A11 - A22) + 4.f * A12 * A12)) / (2 * LK_WIN * LK_WIN);
        if (minEig < d.min_eig_thr || D < 1.1920929e-07f) {
            if (level == 0) status = 0;
            continue;
        }
        D = 1.f / D;
        nextx -= half; nexty -= half;
        float pdx = 0.f, pdy = 0.f;
        bool staged = false;
        int jox = 0, joy = 0;
        for (int j = 0; j < d.max_count; ++j) {
            const int inx = cv_floor(nextx), iny = cv_floor(nexty);
            if (inx < -LK_WIN || inx >= lw || iny < -LK_WIN || iny >= lh) {
                if (level == 0) status = 0;
                break;
            }
            int r00, r01, r10, r11;
            lk_weights(nextx - inx, nexty - iny, r00, r01, r10, r11);
            int rx = inx - jox, ry = iny - joy;
            if (!staged || rx < 0 || rx > LK_JR - 22 || ry < 0 || ry > LK_JR - 22) {      // uniform
                jox = inx - LK_JS; joy = iny - LK_JS; staged = true; rx = LK_JS; ry = LK_JS;
                lk_stage_region(s.jreg, J, lw, lh, jox, joy, lane);
            }
            long long b1, b2;
            {
                // the lane's 2 x 8 search-window bytes as four dwords -> seven (value - template) differences (lk_diffs7), then
                // sum diff Ix and sum diff Iy as v_dot2_i32_i16 over pairs of pixels (the differences packed two to a register)
                const uint8_t* p0 = s.jreg + (ry + lyc) * LK_JR + rx + x0;
                uint2 ra, rb;
                __builtin_memcpy(&ra, p0, 8); __builtin_memcpy(&rb, p0 + LK_JR, 8);
                const unsigned wl = (unsigned)r00 | ((unsigned)r10 << 16), wr = (unsigned)r01 | ((unsigned)r11 << 16);
                // a lane's seven products fit 32 bits with room to spare (|diff| <= 255 * 32, |Ix|, |Iy| <= 16 * 255: < 2.4e8 in
                // total), so the lane sums are formed in 32 bits and widened once; the wavefront sums stay exact int64
                int dq[7];
                lk_diffs7(ra, rb, wl, wr, cq, dq);
                int s1 = mul24(dq[6], gx6), s2 = mul24(dq[6], gy6);
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const unsigned dp = lk_pack16(dq[2 * j], dq[2 * j + 1]);
                    s1 = vg_sdot2(dp, gxp[j], s1); s2 = vg_sdot2(dp, gyp[j], s2);
                }
                b1 = wave_sum_i32(s1); b2 = wave_sum_i32(s2);       // (|s| <= 7 * 8160 * 4080 = 2.3e8 < 2^28)
            }
            const float fb1 = (float)b1 * FLT_SCALE, fb2 = (float)b2 * FLT_SCALE;
            const float dx = (A12 * fb2 - A22 * fb1) * D, dy = (A12 * fb1 - A11 * fb2) * D;
            nextx += dx; nexty += dy;
            nx = nextx + half; ny = nexty + half;
            if ((double)dx * dx + (double)dy * dy <= d.eps2) break;
            if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                nx -= dx * 0.5f; ny -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (status && level == 0) {
            const float ex = nx - half, ey = ny - half;
            const int inx = cv_floor(ex), iny = cv_floor(ey);
            if (inx < -LK_WIN || inx >= lw || iny < -LK_WIN || iny >= lh) { status = 0; continue; }
            int r00, r01, r10, r11;
            lk_weights(ex - inx, ey - iny, r00, r01, r10, r11);
            int rx = inx - jox, ry = iny - joy;
            if (!staged || rx < 0 || rx > LK_JR - 22 || ry < 0 || ry > LK_JR - 22) {
                jox = inx - LK_JS; joy = iny - LK_JS; staged = true; rx = LK_JS; ry = LK_JS;
                lk_stage_region(s.jreg, J, lw, lh, jox, joy, lane);
            }
            int e = 0;               // (|diff| <= 8160: the wavefront total is < 2^22)
            {
                const uint8_t* p0 = s.jreg + (ry + lyc) * LK_JR + rx + x0;
                uint2 ra, rb;
                __builtin_memcpy(&ra, p0, 8); __builtin_memcpy(&rb, p0 + LK_JR, 8);
                const unsigned wl = (unsigned)r00 | ((unsigned)r10 << 16), wr = (unsigned)r01 | ((unsigned)r11 << 16);
                int dq[7];
                lk_diffs7(ra, rb, wl, wr, cq, dq);
#pragma unroll
                for (int q = 0; q < 7; ++q) e += act ? (dq[q] < 0 ? -dq[q] : dq[q]) : 0;
            }
            e = wave_sum_small(e);
            err = ((float)e * 1.f) / (float)(32 * LK_WIN * LK_WIN);      // a division, as OpenCV's expression parses (F3)
        }
    }
    if (lane == 0) {
        const size_t o = (size_t)cam * d.max_pts + t;
        d.next_xy[o * 2] = nx; d.next_xy[o * 2 + 1] = ny;
        d.status[o] = (uint8_t)status;
        d.err[o] = err;
    }
}

// ================================================================================================ GFTT
// min-eigenvalue map (corner.cpp cornerMinEigenVal, blockSize 3, Sobel 3) AND the corner candidates in one pass (round 4: the map
// used to go to HBM, 1.44 MB per frame, and come back nine times per pixel in a second kernel).  Tile = 64 x 16 pixels; the map of
// the tile and of a one-pixel ring around it lives in LDS, so the 3 x 3 comparison of goodFeaturesToTrack needs no second pass:
//   threshold (THRESH_TOZERO at maxVal * quality) + 3x3 dilate equality + mask   ==   value > threshold, value >= its 8 neighbours
// (a neighbour above the value is above the threshold too), and the second condition does not depend on the threshold.  The threshold
// does depend on the maximum over the whole (masked) image, which no tile knows: a tile prunes with a LOWER bound of it -- quality x
// the larger of its own masked maximum and the running maximum the tiles before it have left in HBM (an atomicMax, so the bound and
// with it the length of the list depend on timing) -- and fe_select_kernel applies the exact threshold before it sorts: the list that
// survives is the list of the two-pass form whatever the timing.  (Values <= 0 are never corners: the exact threshold is >= 0 whenever
// the maximum is, and a masked maximum below zero does not occur for the minimum eigenvalue of a sum of outer products beyond
// rounding; oracle/ASSUMPTIONS.md F10.)
// key = (ordered float bits << 32) | linear index   (sort descending = value desc, then index desc)
// The map itself reaches HBM only when the caller asked for it (vg_fe_keep_eig: tests compare it bit by bit).
#define ME_R 16
extern "C" __global__ __launch_bounds__(256) void fe_mineig_kernel(FeDev d, double quality) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[ME_R + 6][72];   // rows y0-3 .. y0+R+2, cols x0-4 .. x0+67 (reflected coordinates)
    __shared__ __attribute__((aligned(16))) float gx[ME_R + 4][68], gy[ME_R + 4][68];   // gradients on rows y0-2 .. y0+R+1, cols x0-2 .. x0+65
    __shared__ float eg[ME_R + 2][66];                            // the map on rows y0-1 .. y0+R, cols x0-1 .. x0+64
    __shared__ __attribute__((aligned(16))) uint8_t mk[ME_R][64];         // the mask under the tile's own pixels (0 outside the image)

    __shared__ float bmax[4];
    __shared__ unsigned wcount[4], wbase[4], lbound;
    const int cam = blockIdx.z, W = d.W, H = d.H;
    const int x0 = blockIdx.x * 64, y0 = blockIdx.y * ME_R;
    const glb_u8* img = (const glb_u8*)d.cur_planes[cam];          // level 0
    const glb_u8* gmask = (const glb_u8*)(d.mask + (size_t)cam * W * H);
    const float k1 = (float)(1.0 / 3060.0), k2 = (float)(2.0 / 3060.0);
    // A tile whose halo lies inside the image (280 of the 360 tiles of a 752 x 480 frame) needs no reflection anywhere: its pixels and its
    // mask arrive as aligned 32-bit words (x0 is a multiple of 64, W of 4) and a thread forms FOUR neighbouring gradients from six words
    // of the tile -- the same float expressions as the general path below, evaluated on the same values (round 5: the byte-by-byte
    // tile load and the per-pixel gradient with its reflection tests were 500 of the ~1150 instructions of a thread).
    const bool interior = (W & 3) == 0 && x0 >= 4 && x0 + 68 <= W && y0 >= 3 && y0 + ME_R + 3 <= H;
    if (interior) {
        const glb_u8* src = img + (size_t)(y0 - 3) * W + (x0 - 4);
        for (int k = threadIdx.x; k < (ME_R + 6) * 18; k += 256) {
            const int yy = k / 18, wx = k - 18 * yy;
            const unsigned v = *(const glb_u32*)(src + (size_t)yy * W + 4 * wx);
            __builtin_memcpy(&tile[yy][4 * wx], &v, 4);
        }
        {
            const int ly = threadIdx.x >> 4, wx = threadIdx.x & 15;
            const unsigned v = *(const glb_u32*)(gmask + (size_t)(y0 + ly) * W + x0 + 4 * wx);
            __builtin_memcpy(&mk[ly][4 * wx], &v, 4);
        }
        __syncthreads();
        for (int k = threadIdx.x; k < (ME_R + 4) * 17; k += 256) {
            const int yy = k / 17, g = k - 17 * yy;                // gradient row yy, columns 4g .. 4g+3 = tile row yy + 1, columns 4g+2 .. 4g+5
            int A[6], B[6], C[6];                                  // tile columns 4g+1 .. 4g+6 of the rows above / at / below
            {
                unsigned w[6];
                __builtin_memcpy(&w[0], &tile[yy][4 * g], 8); __builtin_memcpy(&w[2], &tile[yy + 1][4 * g], 8); __builtin_memcpy(&w[4], &tile[yy + 2][4 * g], 8);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const int sh = 8 * ((j + 1) & 3), hi = (j + 1) >> 2;
                    A[j] = (int)((w[0 + hi] >> sh) & 255u); B[j] = (int)((w[2 + hi] >> sh) & 255u); C[j] = (int)((w[4 + hi] >> sh) & 255u);
                }
            }
            float dxv[4], dyv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float r0 = (float)(A[q + 2] - A[q]);
                const float r1 = (float)(B[q + 2] - B[q]);
                const float r2 = (float)(C[q + 2] - C[q]);
                dxv[q] = k2 * r1 + k1 * (r0 + r2);
                const float s0 = (float)A[q + 1] * k2 + (float)(A[q] + A[q + 2]) * k1;
                const float s2 = (float)C[q + 1] * k2 + (float)(C[q] + C[q + 2]) * k1;
                dyv[q] = s2 - s0;
            }
            __builtin_memcpy(&gx[yy][4 * g], dxv, 16); __builtin_memcpy(&gy[yy][4 * g], dyv, 16);
        }
    } else {
    auto PX = [&](int y, int x) {                                  // image pixel at reflected coordinates (clamped beyond one reflection:
        int ry = reflect101(y, H), rx = reflect101(x, W);          //  nothing an in-image result depends on lies that far out)
        ry = ry < 0 ? 0 : (ry >= H ? H - 1 : ry);
        rx = rx < 0 ? 0 : (rx >= W ? W - 1 : rx);
        return (int)img[(size_t)ry * W + rx];
    };
    for (int k = threadIdx.x; k < (ME_R + 6) * 70; k += 256) {
        const int yy = k / 70, xx = k - 70 * yy;
        // NB: REFLECT_101 is applied per filter stage in OpenCV (Sobel on the image, then boxFilter on cov); the halo
        // below holds image pixels at reflected coordinates, gradients are evaluated at reflected positions too.
        tile[yy][xx + 1] = (uint8_t)PX(y0 - 3 + yy, x0 - 3 + xx);
    }
    for (int k = threadIdx.x; k < ME_R * 64; k += 256) {
        const int ly = k >> 6, lx = k & 63, y = y0 + ly, x = x0 + lx;
        mk[ly][lx] = (x < W && y < H) ? (uint8_t)gmask[(size_t)y * W + x] : (uint8_t)0;
    }
    __syncthreads();
    // gradients at (y0-2+yy, x0-2+xx): position may lie outside the image -> the boxFilter's reflect-101 wants the
    // gradient AT THE REFLECTED POSITION, which is not the gradient computed from reflected pixels; handled below by
    // recomputing from global memory for those few halo positions.
    for (int k = threadIdx.x; k < (ME_R + 4) * 68; k += 256) {
        const int yy = k / 68, xx = k - 68 * yy;
        int Y = y0 - 2 + yy, X = x0 - 2 + xx;
        float dxv, dyv;
        const int Yr = reflect101(Y, H), Xr = reflect101(X, W);
        if (Yr == Y && Xr == X) {
            const uint8_t(*t)[72] = tile;
            const int ty = yy + 1, tx = xx + 2;       // tile index of (Y, X)
            const float r0 = (float)(t[ty - 1][tx + 1] - t[ty - 1][tx - 1]);
            const float r1 = (float)(t[ty][tx + 1] - t[ty][tx - 1]);
            const float r2 = (float)(t[ty + 1][tx + 1] - t[ty + 1][tx - 1]);
            dxv = k2 * r1 + k1 * (r0 + r2);
            const float s0 = (float)t[ty - 1][tx] * k2 + (float)(t[ty - 1][tx - 1] + t[ty - 1][tx + 1]) * k1;
            const float s2 = (float)t[ty + 1][tx] * k2 + (float)(t[ty + 1][tx - 1] + t[ty + 1][tx + 1]) * k1;
            dyv = s2 - s0;
        } else {
            // gradient at the reflected (in-image) position, from global memory
            Y = Yr < 0 ? 0 : (Yr >= H ? H - 1 : Yr); X = Xr < 0 ? 0 : (Xr >= W ? W - 1 : Xr);
            const float r0 = (float)(PX(Y - 1, X + 1) - PX(Y - 1, X - 1));
            const float r1 = (float)(PX(Y, X + 1) - PX(Y, X - 1));
            const float r2 = (float)(PX(Y + 1, X + 1) - PX(Y + 1, X - 1));
            dxv = k2 * r1 + k1 * (r0 + r2);
            const float s0 = (float)PX(Y - 1, X) * k2 + (float)(PX(Y - 1, X - 1) + PX(Y - 1, X + 1)) * k1;
            const float s2 = (float)PX(Y + 1, X) * k2 + (float)(PX(Y + 1, X - 1) + PX(Y + 1, X + 1)) * k1;
            dyv = s2 - s0;
        }
        gx[yy][xx] = dxv; gy[yy][xx] = dyv;
    }
    }
    __syncthreads();
    // The map on the tile and its ring; the masked maximum over the tile's own pixels.  The 3 x 3 box is SEPARABLE, as cv::boxFilter
    // runs it (RowSum<float, double>: ((p[x-1] + p[x]) + p[x+1]) per channel, then ColumnSum<double, float>: ((r[y-1] + r[y]) + r[y+1]);
    // oracle/fe_cpu.cpp, round 5): a thread walks DOWN a column of the map over a strip of six rows and keeps the last three row sums
    // in registers -- per pixel 12 products / conversions and 14 FP64 adds instead of 27 + 27, no second LDS array.
    float m = -INFINITY;
    if (threadIdx.x < 3 * 66) {
        const int strip = threadIdx.x / 66, ex = threadIdx.x - 66 * strip;
        const int ey0 = 6 * strip;                                 // map rows ey0 .. ey0+5 of (ME_R + 2) = 18: gradient rows ey0 .. ey0+7
        const int x = x0 - 1 + ex;
        double r0[3] = {0, 0, 0}, r1[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int yy = ey0 + j;
            double rs[3];
            {
                const float a0 = gx[yy][ex], b0 = gy[yy][ex], a1 = gx[yy][ex + 1], b1 = gy[yy][ex + 1], a2 = gx[yy][ex + 2], b2 = gy[yy][ex + 2];
                rs[0] = ((double)(a0 * a0) + (double)(a1 * a1)) + (double)(a2 * a2);
                rs[1] = ((double)(a0 * b0) + (double)(a1 * b1)) + (double)(a2 * b2);
                rs[2] = ((double)(b0 * b0) + (double)(b1 * b1)) + (double)(b2 * b2);
            }
            if (j >= 2) {
                const int ey = yy - 2, y = y0 - 1 + ey;
                float val = -INFINITY;
                if (x >= 0 && y >= 0 && x < W && y < H) {
                    const double sxx = (r0[0] + r1[0]) + rs[0], sxy = (r0[1] + r1[1]) + rs[1], syy = (r0[2] + r1[2]) + rs[2];
                    const float a = (float)sxx * 0.5f, b = (float)sxy, c = (float)syy * 0.5f;
                    val = (float)((a + c) - sqrtf((a - c) * (a - c) + b * b));
                    if (ey >= 1 && ey <= ME_R && ex >= 1 && ex <= 64) {             // the tile's own pixel
                        if (d.keep_eig) d.eig[(size_t)cam * W * H + (size_t)y * W + x] = val;
                        if (mk[ey - 1][ex - 1]) m = fmaxf(m, val);
                    }
                }
                eg[ey][ex] = val;
            }
#pragma unroll
            for (int c3 = 0; c3 < 3; ++c3) { r0[c3] = r1[c3]; r1[c3] = rs[c3]; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_down(m, o, 64));
    if ((threadIdx.x & 63) == 0) bmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float bm = fmaxf(fmaxf(bmax[0], bmax[1]), fmaxf(bmax[2], bmax[3]));
        // per-stream maximum for minMaxLoc (max is order-independent -> deterministic); slot zeroed with ncand
        unsigned seen = 0u;
        if (bm != -INFINITY) { const unsigned mine = ford(bm); seen = atomicMax(&d.ncand[FE_CNT_STRIDE * cam + 32], mine); seen = seen > mine ? seen : mine; }
        else seen = atomicMax(&d.ncand[FE_CNT_STRIDE * cam + 32], 0u);          // (a read)
        lbound = seen;
    }
    __syncthreads();
    // lower bound of the threshold: the same expression as the exact one (fe_select_kernel), monotone in the maximum
    const unsigned sb = lbound;
    const float lbf = sb ? funord(sb) : -INFINITY;
    const float thr_lb = (float)(((lbf == -INFINITY) ? 0.0 : (double)lbf) * quality);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bool is_cand[ME_R / 4];
    unsigned my[ME_R / 4], tot = 0;
#pragma unroll
    for (int i = 0; i < ME_R / 4; ++i) {
        const int ly = wv + 4 * i, x = x0 + lane, y = y0 + ly;
        bool cnd = false;
        if (x >= 1 && y >= 1 && x < W - 1 && y < H - 1) {
            const float raw = eg[ly + 1][lane + 1];
            if (raw > 0.f && raw > thr_lb && mk[ly][lane]) {
                float mx = raw;
#pragma unroll
                for (int v = 0; v < 3; ++v)
#pragma unroll
                    for (int u = 0; u < 3; ++u) mx = fmaxf(mx, eg[ly + v][lane + u]);
                cnd = (raw == mx);
            }
        }
        const unsigned long long bal = __ballot(cnd);
        is_cand[i] = cnd;
        my[i] = tot + __popcll(bal & ((1ull << lane) - 1ull));
        tot += __popcll(bal);
    }
    // block-aggregated append: one global atomic per workgroup (the final order comes from the sort)
    if (lane == 0) wcount[wv] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned all = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        const unsigned base = all ? atomicAdd(&d.ncand[FE_CNT_STRIDE * cam], all) : 0u;
        wbase[0] = base; wbase[1] = base + wcount[0]; wbase[2] = wbase[1] + wcount[1]; wbase[3] = wbase[2] + wcount[2];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ME_R / 4; ++i) {
        if (!is_cand[i]) continue;
        const int ly = wv + 4 * i, x = x0 + lane, y = y0 + ly;
        const unsigned slot = wbase[wv] + my[i];
        if (slot < (unsigned)d.cand_cap)
            d.keys[(size_t)cam * d.cand_cap + slot] = ((unsigned long long)ford(eg[ly + 1][lane + 1]) << 32) | (unsigned)(y * W + x);
    }
}

// one workgroup (1024 threads) per camera: in-place bitonic sort (descending) of the candidate keys, padded with 0
// ================================================================================================ select
// goodFeaturesToTrack's "sort all corners by quality, then walk them greedily with the min-distance grid"
// (featureselect.cpp) — but the walk stops after max_corners acceptances, typically inside the best few hundred of
// tens of thousands of candidates.  One workgroup per stream:
//   1. histogram of the candidates over 4096 value bins (linear between the threshold and the maximum; monotone)
//   2. take bins from the top until ~FE_SEL_CAP candidates are covered, compact them into LDS, bitonic-sort (full 64-bit
//      key: value desc, then index desc = OpenCV's pointer tie-break), wave 0 walks them with the cell grid
//   3. repeat with the next bins while corners are still missing.
// The visiting order is exactly the order of the full sort, so the output is identical.  A single bin holding more than
// FE_SEL_CAP candidates (flat images) falls back to the full bitonic sort in global memory.
#define FE_SEL_CAP 4096
#define FE_SEL_BINS 4096
struct SelWalk {
    unsigned short (*cellxy)[7][2];
    unsigned char* cellcnt;
    int W, H, cell, gw, gh, maxc;
    double md2;
    bool use_dist;
};
// wave 0 only: visit `cnt` sorted keys, append accepted corners; returns the new acceptance count
FDEV int sel_walk(const SelWalk& w, const unsigned long long* keys, unsigned cnt, int nacc, float* corners, int lane) {
    // The walk over the sorted candidates is a serial chain on one wavefront (a candidate is tested against everything accepted before
    // it).  Round 6 takes out of the chain what does not belong there: every lane decodes ITS candidate of the pass once (two integer
    // divisions by the image width and the cell size were ~150 instructions per visit), and tests it against the grid AS IT STANDS AT THE
    // START OF THE PASS -- 64 candidates in parallel; a candidate too close to a corner accepted in an earlier pass stays rejected
    // whatever this pass adds (the grid only grows), so the serial loop visits the survivors only and gives the same list.
    for (unsigned base = 0; base < cnt && nacc < w.maxc; base += 64) {
        const unsigned long long mykey = (base + lane < cnt) ? keys[base + lane] : 0ull;
        const int m = (cnt - base) < 64u ? (int)(cnt - base) : 64;
        const unsigned myidx = (unsigned)(mykey & 0xffffffffull);
        const int my = (int)(myidx / (unsigned)w.W), mx = (int)(myidx - (unsigned)my * (unsigned)w.W);
        const int mxc = mx / w.cell, myc = my / w.cell;
        bool pre_bad = false;
        if (w.use_dist && lane < m) {
            for (int nb = 0; nb < 9; ++nb) {
                const int cx = mxc - 1 + nb % 3, cy = myc - 1 + nb / 3;
                if (cx < 0 || cy < 0 || cx >= w.gw || cy >= w.gh) continue;
                const int c = cy * w.gw + cx;
                const int kc = w.cellcnt[c];
                for (int slot = 0; slot < kc; ++slot) {
                    const float dx = (float)(mx - (int)w.cellxy[c][slot][0]), dy = (float)(my - (int)w.cellxy[c][slot][1]);
                    pre_bad = pre_bad || (double)(dx * dx + dy * dy) < w.md2;
                }
            }
        }
        unsigned long long todo = __ballot(lane < m && !pre_bad);
        while (todo && nacc < w.maxc) {
            const int j = (int)__builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int x = __builtin_amdgcn_readlane(mx, j), y = __builtin_amdgcn_readlane(my, j);
            const int xc = __builtin_amdgcn_readlane(mxc, j), yc = __builtin_amdgcn_readlane(myc, j);
            bool bad = false;
            if (w.use_dist && lane < 63) {
                const int nb = lane / 7, slot = lane % 7;          // 9 neighbour cells x 7 slots
                const int cx = xc - 1 + nb % 3, cy = yc - 1 + nb / 3;
                if (cx >= 0 && cy >= 0 && cx < w.gw && cy < w.gh) {
                    const int c = cy * w.gw + cx;
                    if (slot < w.cellcnt[c]) {
                        const float dx = (float)(x - (int)w.cellxy[c][slot][0]), dy = (float)(y - (int)w.cellxy[c][slot][1]);
                        bad = (double)(dx * dx + dy * dy) < w.md2;
                    }
                }
            }
            if (!__any(bad)) {
                if (lane == 0) {
                    const int c = yc * w.gw + xc;
                    const int k = w.cellcnt[c];
                    if (k < 7) { w.cellxy[c][k][0] = (unsigned short)x; w.cellxy[c][k][1] = (unsigned short)y; w.cellcnt[c] = (unsigned char)(k + 1); }
                    corners[nacc * 2] = (float)x; corners[nacc * 2 + 1] = (float)y;
                }
                ++nacc;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // lane 0's cell update before the next visit
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    return nacc;
}
FDEV void sel_bitonic_lds(unsigned long long* sk, unsigned np) {
    for (unsigned size = 2; size <= np; size <<= 1)
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned t = threadIdx.x; t < np / 2; t += 1024) {
                const unsigned lo = 2 * t - (t & (stride - 1));      // index with bit `stride` cleared
                const unsigned hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = sk[lo], b = sk[hi];
                if ((a < b) == desc) { sk[lo] = b; sk[hi] = a; }
            }
            __syncthreads();
        }
}
extern "C" __global__ __launch_bounds__(1024) void fe_select_kernel(FeDev d, double quality, float min_dist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sel_smem[];
    unsigned long long* sk = (unsigned long long*)sel_smem;                               // [FE_SEL_CAP]
    unsigned* hist = (unsigned*)(sk + FE_SEL_CAP);                                        // [FE_SEL_BINS]
    unsigned* grp = hist + FE_SEL_BINS;                                                   // [64] group sums
    int* ctl = (int*)(grp + 64);                                                          // [8]
    unsigned short (*cellxy)[7][2] = (unsigned short (*)[7][2])(ctl + 8);                 // [FE_MAX_CELLS][7][2]
    unsigned char* cellcnt = (unsigned char*)(cellxy + FE_MAX_CELLS);                     // [FE_MAX_CELLS]
    const int cam = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    SelWalk w;
    w.cellxy = cellxy; w.cellcnt = cellcnt; w.W = d.W; w.H = d.H;
    w.maxc = d.max_corners[cam];
    w.cell = __float2int_rn(min_dist) < 1 ? 1 : __float2int_rn(min_dist);
    w.gw = (d.W + w.cell - 1) / w.cell; w.gh = (d.H + w.cell - 1) / w.cell;
    w.md2 = (double)min_dist * (double)min_dist;
    w.use_dist = min_dist >= 1.f;
    for (int k = tid; k < w.gw * w.gh; k += 1024) cellcnt[k] = 0;
    unsigned nall = d.ncand[FE_CNT_STRIDE * cam];
    // fe_mineig_kernel prunes with a timing-dependent LOWER bound of the threshold, so on a frame whose strong corners are found late
    // the list can run long; past the capacity it dropped keys (arbitrary ones): the result would no longer be the reference's
    // list.  Not silently: the stream reports -1 corners and the host calls return VG_ERR_UNSUPPORTED.
    if (nall > (unsigned)d.cand_cap) {
        if (tid == 0) d.ncorners[cam] = -1;
        return;
    }
    unsigned long long* keys = d.keys + (size_t)cam * d.cand_cap;
    float* corners = d.corners + (size_t)cam * d.max_pts * 2;
    int nacc = 0;
    // The exact threshold (THRESH_TOZERO at maxVal * quality; minMaxLoc over an empty mask leaves maxVal = 0 in the reference).
    // fe_mineig_kernel pruned with a lower bound of it: whatever it let through in excess goes here, so the surviving list does not
    // depend on the order in which the tiles ran.  The survivors are compacted to the front of the list (their order is irrelevant:
    // everything below sorts).
    const unsigned smax = d.ncand[FE_CNT_STRIDE * cam + 32];
    const float mf = smax ? funord(smax) : -INFINITY;
    const double maxVal = (mf == -INFINITY) ? 0.0 : (double)mf;
    const float thr = (float)(maxVal * quality);
    if (tid == 0) ctl[5] = 0;
    __syncthreads();
    for (unsigned i0 = 0; i0 < nall; i0 += 1024) {
        // (in place: a chunk's survivors land at or before the chunk's own start, and every key of the chunk is in a register
        //  before the first store of the chunk)
        const unsigned i = i0 + tid;
        const unsigned long long key = i < nall ? keys[i] : 0ull;
        const bool keep = i < nall && funord((unsigned)(key >> 32)) > thr;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) grp[wave] = __popcll(bal);
        __syncthreads();
        unsigned before = (unsigned)ctl[5];
        for (int q = 0; q < wave; ++q) before += grp[q];
        if (keep) keys[before + __popcll(bal & ((1ull << lane) - 1ull))] = key;
        __syncthreads();
        if (tid == 0) { unsigned t = 0; for (int q = 0; q < 16; ++q) t += grp[q]; ctl[5] += (int)t; }
        __syncthreads();
    }
    const unsigned n = (unsigned)ctl[5];
    __threadfence_block();
    __syncthreads();
    // (round 6) the walk stops at max_corners accepted corners and rarely looks past the first few hundred candidates, so neither path
    // sorts more than it has to: lists up to `target` keys are sorted whole, longer ones leave in chunks of about `target` keys from
    // the top value bins (the next chunk only if the walk runs out of candidates; the order of the walk, and with it the result, does
    // not depend on where the chunks are cut)
    const unsigned target = (unsigned)(8 * w.maxc < 512 ? 512 : (8 * w.maxc > FE_SEL_CAP ? FE_SEL_CAP : 8 * w.maxc));
    if (n <= target) {
        unsigned np = 1;
        while (np < n) np <<= 1;
        for (unsigned i = tid; i < np; i += 1024) sk[i] = i < n ? keys[i] : 0ull;
        __syncthreads();
        sel_bitonic_lds(sk, np);
        if (wave == 0) { nacc = sel_walk(w, sk, n, 0, corners, lane); if (lane == 0) d.ncorners[cam] = nacc; }
        return;
    }
    // value bins: linear between the threshold and the maximum
    const float span = (float)maxVal - thr;
    const float scale = span > 0.f ? (float)(FE_SEL_BINS - 1) / span : 0.f;
    auto bin_of = [&](unsigned long long key) {
        const float v = funord((unsigned)(key >> 32));
        int b = (int)((v - thr) * scale);
        return b < 0 ? 0 : (b > FE_SEL_BINS - 1 ? FE_SEL_BINS - 1 : b);
    };
    for (int k = tid; k < FE_SEL_BINS; k += 1024) hist[k] = 0u;
    __syncthreads();
    for (unsigned i = tid; i < n; i += 1024) atomicAdd(&hist[bin_of(keys[i])], 1u);
    __syncthreads();
    int b_hi = FE_SEL_BINS - 1;
    for (;;) {
        // ---- next chunk of bins [b_lo, b_hi]: wave 0 sums groups of 64 bins, lane 0 packs
        if (wave == 0) {
            unsigned g = 0;
            const int top = b_hi - 64 * lane;                 // lane covers bins top-63 .. top (descending groups)
            for (int q = 0; q < 64; ++q) { const int b = top - q; g += b >= 0 ? hist[b] : 0u; }
            grp[lane] = g;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                unsigned tot = 0;
                int b = b_hi, gi = 0;
                while (gi < 64 && b - 63 >= 0 && tot + grp[gi] <= target) { tot += grp[gi]; b -= 64; ++gi; }
                while (b >= 0 && tot + hist[b] <= target) { tot += hist[b]; --b; }
                if (b == b_hi && hist[b] <= FE_SEL_CAP) { tot = hist[b]; --b; }      // (a single bin above the target, below the capacity)
                ctl[0] = b + 1;                               // b_lo
                ctl[1] = (int)tot;
                ctl[2] = (b == b_hi) ? 1 : 0;                 // a single bin exceeds the capacity -> fallback
                ctl[3] = 0;                                   // compaction counter
            }
        }
        __syncthreads();
        const int b_lo = ctl[0];
        if (ctl[2]) break;
        for (unsigned i = tid; i < n; i += 1024) {
            const unsigned long long key = keys[i];
            const int b = bin_of(key);
            if (b >= b_lo && b <= b_hi) sk[atomicAdd((unsigned*)&ctl[3], 1u)] = key;
        }
        __syncthreads();
        const unsigned cnt = (unsigned)ctl[1];
        unsigned np = 1;
        while (np < cnt) np <<= 1;
        for (unsigned i = cnt + tid; i < np; i += 1024) sk[i] = 0ull;
        __syncthreads();
        sel_bitonic_lds(sk, np);
        if (wave == 0) {
            nacc = sel_walk(w, sk, cnt, nacc, corners, lane);
            if (lane == 0) ctl[4] = nacc;
        }
        __syncthreads();
        nacc = ctl[4];
        if (nacc >= w.maxc || b_lo == 0) { if (tid == 0) d.ncorners[cam] = nacc; return; }
        b_hi = b_lo - 1;
        __syncthreads();
    }
    // ---- fallback: full bitonic sort in global memory, then one walk (restarted from scratch: the chunks walked so
    //      far are a prefix of the same order, so re-walking them reproduces the same acceptances)
    for (int k = tid; k < w.gw * w.gh; k += 1024) cellcnt[k] = 0;
    unsigned np = 1;
    while (np < n) np <<= 1;
    for (unsigned i = n + tid; i < np; i += 1024) keys[i] = 0ull;
    __syncthreads();
    for (unsigned size = 2; size <= np; size <<= 1)
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned t = tid; t < np / 2; t += 1024) {
                const unsigned lo = 2 * t - (t & (stride - 1));
                const unsigned hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
            __threadfence_block();
            __syncthreads();
        }
    if (wave == 0) { nacc = sel_walk(w, keys, n, 0, corners, lane); if (lane == 0) d.ncorners[cam] = nacc; }
}
// ================================================================================================ setMask / lift
// FeatureTracker::setMask (feature_tracker.cpp:36-69), SURVEY.md 8(f) row 1.  One workgroup per stream:
//   1. stable order by track_cnt descending = ascending bitonic sort of (INT_MAX - cnt) << 32 | index in LDS;
//   2. wave 0 walks the points; a point is kept iff its rounded position is inside the image, the base mask there is
//      255 and no previously kept point's filled disc covers it — exactly "mask.at(pt) == 255" of :59 without
//      touching the mask plane (lanes test 64 kept points at a time);
//   3. the mask plane of the stream is rebuilt by fe_stamp_kernel (base mask or 255, then every kept disc in parallel:
//      the union does not depend on the order).
// Disc rule: dx^2 + dy^2 <= r^2 (the restatement of cv::circle(..., -1), oracle/ASSUMPTIONS.md F7).
#define FE_SETMASK_MAX 2048
extern "C" __global__ __launch_bounds__(256) void fe_setmask_kernel(FeDev d, const float* __restrict__ pts_xy, const int* __restrict__ track_cnt,
                                                                    const int* __restrict__ npts, const uint8_t* const* __restrict__ base_masks,
                                                                    int radius, int* __restrict__ kept_index, int* __restrict__ n_kept,
                                                                    int* __restrict__ kept_xy) {
    __shared__ unsigned long long sk[FE_SETMASK_MAX];
    __shared__ short kx[FE_SETMASK_MAX], ky[FE_SETMASK_MAX];
    const int cam = blockIdx.x, tid = threadIdx.x, lane = tid & 63, W = d.W, H = d.H;
    const int n = npts[cam];
    const float* p = pts_xy + (size_t)cam * d.max_pts * 2;
    const int* tc = track_cnt + (size_t)cam * d.max_pts;
    const uint8_t* bm = base_masks ? base_masks[cam] : nullptr;
    unsigned np = 1;
    while (np < (unsigned)n) np <<= 1;
    for (unsigned i = tid; i < np; i += 256)
        sk[i] = i < (unsigned)n ? (((unsigned long long)(unsigned)(0x7fffffff - tc[i]) << 32) | i) : ~0ull;
    __syncthreads();
    for (unsigned size = 2; size <= np; size <<= 1)
        for (unsigned stride = size >> 1; stride > 0; stride >>= 1) {
            for (unsigned t = tid; t < np / 2; t += 256) {
                const unsigned lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool asc = ((lo & size) == 0);
                const unsigned long long a = sk[lo], b = sk[hi];
                if ((a > b) == asc) { sk[lo] = b; sk[hi] = a; }
            }
            __syncthreads();
        }
    if (tid < 64) {
        const int r2 = radius * radius;
        int nk = 0;
        for (int q = 0; q < n; ++q) {
            const int i = (int)(unsigned)(sk[q] & 0xffffffffull);
            const int px = __float2int_rn(p[2 * i]), py = __float2int_rn(p[2 * i + 1]);      // Point2f -> Point: round half to even
            bool ok = px >= 0 && py >= 0 && px < W && py < H;
            if (ok && bm) ok = bm[(size_t)py * W + px] == 255;
            if (ok) {
                bool cov = false;
                for (int base = 0; base < nk; base += 64) {
                    const int j = base + lane;
                    if (j < nk) { const int dx = px - kx[j], dy = py - ky[j]; cov = cov || (dx * dx + dy * dy <= r2); }
                }
                ok = !__any(cov);
            }
            if (ok) {
                if (lane == 0) {
                    kx[nk] = (short)px; ky[nk] = (short)py;
                    kept_index[(size_t)cam * d.max_pts + nk] = i;
                    kept_xy[((size_t)cam * d.max_pts + nk) * 2] = px; kept_xy[((size_t)cam * d.max_pts + nk) * 2 + 1] = py;
                }
                ++nk;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) n_kept[cam] = nk;
    }
}
// grid (max_pts + 1, cams): block 0 of a stream resets its mask plane to the base mask / 255 ... done by the host with
// memcpy/memset; blocks stamp the disc of kept point blockIdx.x
extern "C" __global__ __launch_bounds__(256) void fe_stamp_kernel(FeDev d, const int* __restrict__ n_kept, const int* __restrict__ kept_xy, int radius) {
    const int cam = blockIdx.y, k = blockIdx.x;
    if (k >= n_kept[cam]) return;
    const int cx = kept_xy[((size_t)cam * d.max_pts + k) * 2], cy = kept_xy[((size_t)cam * d.max_pts + k) * 2 + 1];
    uint8_t* m = const_cast<uint8_t*>(d.mask) + (size_t)cam * d.W * d.H;
    const int side = 2 * radius + 1, r2 = radius * radius;
    for (int t = threadIdx.x; t < side * side; t += 256) {
        const int dy = t / side - radius, dx = t % side - radius;
        const int x = cx + dx, y = cy + dy;
        if (x >= 0 && y >= 0 && x < d.W && y < d.H && dx * dx + dy * dy <= r2) m[(size_t)y * d.W + x] = 0;
    }
}
// PinholeCamera::liftProjective, recursive distortion model with n = 8 (PinholeCamera.cc:450-510, :646-661); thread per
// point, double arithmetic in the reference's order (this file is compiled with -ffp-contract=off)
extern "C" __global__ __launch_bounds__(256) void fe_lift_kernel(const float* __restrict__ pts_xy, int n, double fx, double fy, double cx, double cy,
                                                                 double k1, double k2, double p1, double p2, float* __restrict__ out_xy) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double mx_d = (1.0 / fx) * (double)pts_xy[2 * i] + (-cx / fx), my_d = (1.0 / fy) * (double)pts_xy[2 * i + 1] + (-cy / fy);
    double mx_u = mx_d, my_u = my_d;
#pragma unroll 1
    for (int it = 0; it < 8; ++it) {
        const double mx2 = mx_u * mx_u, my2 = my_u * my_u, mxy = mx_u * my_u, rho2 = mx2 + my2;
        const double rad = k1 * rho2 + k2 * rho2 * rho2;
        const double dx = mx_u * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2);
        const double dy = my_u * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2);
        mx_u = mx_d - dx; my_u = my_d - dy;
    }
    out_xy[2 * i] = (float)mx_u; out_xy[2 * i + 1] = (float)my_u;
}

#define FE_SEL_LDS_BYTES (FE_SEL_CAP * 8 + FE_SEL_BINS * 4 + 64 * 4 + 8 * 4 + FE_MAX_CELLS * 7 * 2 * 2 + FE_MAX_CELLS)

extern "C" hipError_t fe_launch_select(const FeDev& d, double quality, float min_dist, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)fe_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FE_SEL_LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(fe_select_kernel, dim3(d.cams), dim3(1024), FE_SEL_LDS_BYTES, stream, d, quality, min_dist);
    return hipGetLastError();
}
