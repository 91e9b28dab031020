// tests/simt/hip/hip_runtime.h — TEST INFRASTRUCTURE, never part of the product.
//
// A CPU stand-in for <hip/hip_runtime.h> that lets the *unchanged* kernel sources of vins-mono_amd/csrc be compiled
// with g++ and executed in this GPU-less container: every thread of a workgroup is a fiber (ucontext) of one OS thread;
// fibers run until they reach __syncthreads(), a wavefront collective (readlane / DPP / shuffle / ballot / MFMA) or a
// wave barrier and are resumed when their workgroup / wavefront has arrived.  Workgroups run one after another.
// The scheduling ORDER of the fibers is selectable (SIMT_ORDER=forward|reverse|shuffle): a missing barrier shows up as
// a result that depends on it.  There is no implicit wavefront lock-step: lanes that exchange data through LDS must
// be separated by a barrier or __builtin_amdgcn_wave_barrier(), as the tests check by running every order.
//
// Only what the csrc sources use is modelled: see tests/simt/README.md.  libvinsgpu_simt.so built from this is loaded
// by tests/test_simt_*.py only; the package (vins-mono_amd/__init__.py) can not load it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <vector>

#define SIMT_EMULATION 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __shared__ thread_local            // all fibers share one OS thread => one instance per process
#define __launch_bounds__(...)
#define __HIP_DEVICE_COMPILE__ 1

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_simt { unsigned x, y, z; };

namespace simt {
struct Lane {
    uint3_simt tid, bid;
    int flat, lane, wave;
    unsigned xcnt;            // collective counter of this lane (selects the exchange buffer)
};
extern thread_local Lane* cur;
extern thread_local dim3 g_block, g_grid;
void sync_block();
void sync_wave();
uint64_t* xchg(unsigned parity);              // exchange buffer [64] of the current wave
void launch(dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body);
long long clock();
inline int nlanes() { return 64; }
// one collective = deposit, wave barrier, read.  Buffers alternate, so the next collective's deposit cannot clobber
// values a slower lane still has to read.
template <typename F> inline uint64_t collective(uint64_t mine, F&& pick) {
    Lane* me = cur;
    uint64_t* b = xchg(me->xcnt++ & 1);
    b[me->lane] = mine;
    sync_wave();
    return pick(b, me->lane);
}
}  // namespace simt

#define threadIdx (simt::cur->tid)
#define blockIdx (simt::cur->bid)
#define blockDim (simt::g_block)
#define gridDim (simt::g_grid)
static const int warpSize = 64;

inline void __syncthreads() { simt::sync_block(); }
inline void __builtin_amdgcn_s_barrier() { simt::sync_block(); }
inline void __builtin_amdgcn_wave_barrier() { simt::sync_wave(); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)      /* (instruction-scheduling fence: nothing to emulate) */
inline void __threadfence_block() {}
inline void __threadfence() {}
inline long long clock64() { return simt::clock(); }
inline long long wall_clock64() { return simt::clock(); }

// ---- HIP vector types (members x y z w, natural alignment of the whole vector)
#define SIMT_VEC2(T, N) struct alignas(2 * sizeof(T)) N { T x, y; }; inline N make_##N(T x, T y) { return N{x, y}; }
#define SIMT_VEC4(T, N) struct alignas(4 * sizeof(T) > 16 ? 16 : 4 * sizeof(T)) N { T x, y, z, w; }; inline N make_##N(T x, T y, T z, T w) { return N{x, y, z, w}; }
SIMT_VEC2(int, int2) SIMT_VEC2(unsigned, uint2) SIMT_VEC2(float, float2) SIMT_VEC2(double, double2) SIMT_VEC2(short, short2)
SIMT_VEC2(unsigned short, ushort2) SIMT_VEC2(unsigned char, uchar2) SIMT_VEC2(long long, longlong2) SIMT_VEC2(unsigned long long, ulonglong2)
SIMT_VEC4(int, int4) SIMT_VEC4(unsigned, uint4) SIMT_VEC4(float, float4) SIMT_VEC4(double, double4) SIMT_VEC4(short, short4)
SIMT_VEC4(unsigned short, ushort4) SIMT_VEC4(unsigned char, uchar4)
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
inline int __mulhi(int a, int b) { return (int)(((long long)a * b) >> 32); }
inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }

// ---- bit casts / integer helpers
inline int __double2loint(double v) { uint64_t u; memcpy(&u, &v, 8); return (int)(uint32_t)u; }
inline int __double2hiint(double v) { uint64_t u; memcpy(&u, &v, 8); return (int)(uint32_t)(u >> 32); }
inline double __hiloint2double(int hi, int lo) { uint64_t u = ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo; double v; memcpy(&v, &u, 8); return v; }
inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
inline long long __double_as_longlong(double v) { long long u; memcpy(&u, &v, 8); return u; }
inline double __longlong_as_double(long long u) { double v; memcpy(&v, &u, 8); return v; }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
inline double __builtin_amdgcn_rsq(double x) { return 1.0 / std::sqrt(x); }
inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / std::sqrt(x); }
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
using std::fma; using std::fmax; using std::fmin; using std::fabs; using std::sqrt; using std::log1p; using std::atan2;
using std::sin; using std::cos; using std::floor; using std::ceil; using std::lrintf; using std::min; using std::max;
inline float __fsqrt_rn(float x) { return std::sqrt(x); }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline int __float2int_rn(float x) { return (int)std::nearbyint(x); }

// ---- wavefront collectives
inline int __builtin_amdgcn_readlane(int v, int lane) {
    return (int)(uint32_t)simt::collective((uint32_t)v, [&](uint64_t* b, int) { return b[lane & 63]; });
}
inline int __builtin_amdgcn_readfirstlane(int v) {
    // (all lanes of the wavefront are assumed active, which is what the kernels guarantee at their call sites)
    return (int)(uint32_t)simt::collective((uint32_t)v, [&](uint64_t* b, int) { return b[0]; });
}
inline int simt_dpp_src(int lane, int ctrl) {
    const int row = lane & ~15, r = lane & 15;
    if (ctrl >= 0x000 && ctrl <= 0x0FF) return (lane & ~3) + ((ctrl >> (2 * (lane & 3))) & 3);       // quad_perm
    if (ctrl >= 0x101 && ctrl <= 0x10F) { const int s = r + (ctrl & 15); return s < 16 ? row + s : -1; }   // row_shl
    if (ctrl >= 0x111 && ctrl <= 0x11F) { const int s = r - (ctrl & 15); return s >= 0 ? row + s : -1; }   // row_shr
    if (ctrl >= 0x121 && ctrl <= 0x12F) return row + ((r - (ctrl & 15)) & 15);                         // row_ror
    if (ctrl == 0x140) return row + (15 - r);                                                         // row_mirror
    if (ctrl == 0x141) return (lane & ~7) + (7 - (lane & 7));                                         // row_half_mirror
    fprintf(stderr, "simt: unmodelled DPP control 0x%x\n", ctrl);
    abort();
}
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    return (int)(uint32_t)simt::collective((uint32_t)src, [&](uint64_t* b, int lane) -> uint64_t {
        const int s = simt_dpp_src(lane, ctrl);
        const bool en = ((row_mask >> (lane >> 4)) & 1) && ((bank_mask >> ((lane >> 2) & 3)) & 1);
        if (!en) return (uint32_t)old;
        if (s < 0) return bound_ctrl ? 0u : (uint32_t)old;
        return b[s];
    });
}
inline int __builtin_amdgcn_mov_dpp(int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    return __builtin_amdgcn_update_dpp(0, src, ctrl, row_mask, bank_mask, bound_ctrl);
}
template <typename T> inline T simt_shfl(T v, int src_lane_fn(int, int, int), int arg, int width) {
    static_assert(sizeof(T) <= 8, "shuffle of <= 8-byte values");
    uint64_t u = 0; memcpy(&u, &v, sizeof(T));
    const uint64_t r = simt::collective(u, [&](uint64_t* b, int lane) {
        const int s = src_lane_fn(lane, arg, width);
        return b[s < 0 ? lane : s];
    });
    T o; memcpy(&o, &r, sizeof(T)); return o;
}
inline int simt_src_down(int lane, int d, int w) { const int s = lane + d; return (s / w == lane / w && s < 64) ? s : -1; }
inline int simt_src_up(int lane, int d, int w) { const int s = lane - d; return (s >= 0 && s / w == lane / w) ? s : -1; }
inline int simt_src_xor(int lane, int m, int w) { const int s = lane ^ m; return (s / w == lane / w && s < 64) ? s : -1; }
inline int simt_src_idx(int lane, int i, int w) { return (lane / w) * w + (i & (w - 1)); }
template <typename T> inline T __shfl_down(T v, unsigned d, int w = 64) { return simt_shfl(v, simt_src_down, (int)d, w); }
template <typename T> inline T __shfl_up(T v, unsigned d, int w = 64) { return simt_shfl(v, simt_src_up, (int)d, w); }
template <typename T> inline T __shfl_xor(T v, int m, int w = 64) { return simt_shfl(v, simt_src_xor, m, w); }
template <typename T> inline T __shfl(T v, int i, int w = 64) { return simt_shfl(v, simt_src_idx, i, w); }
inline unsigned long long __ballot(int pred) {
    return simt::collective(pred ? 1u : 0u, [&](uint64_t* b, int) {
        uint64_t m = 0;
        for (int l = 0; l < 64; ++l) m |= (uint64_t)(b[l] & 1) << l;
        return m;
    });
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) { return __ballot(pred) == ~0ull; }

// v_mfma_f64_16x16x4_f64: A[i][k] from lane (i = lane & 15, k = lane >> 4), B[k][j] from lane (j = lane & 15, k = lane >> 4),
// D[(lane >> 4) + 4 * reg][lane & 15] in register `reg` — the layout the kernels were validated with on the hardware.
typedef double simt_double4 __attribute__((vector_size(32)));
inline simt_double4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, simt_double4 c, int, int, int) {
    uint64_t ua, ub; memcpy(&ua, &a, 8); memcpy(&ub, &b, 8);
    simt::Lane* me = simt::cur;
    uint64_t* ba = simt::xchg(me->xcnt++ & 1);
    ba[me->lane] = ua;
    simt::sync_wave();
    double A[4][1];
    (void)A;
    double av[16][4];
    for (int l = 0; l < 64; ++l) memcpy(&av[l & 15][l >> 4], &ba[l], 8);
    uint64_t* bb = simt::xchg(me->xcnt++ & 1);
    bb[me->lane] = ub;
    simt::sync_wave();
    double bv[4][16];
    for (int l = 0; l < 64; ++l) memcpy(&bv[l >> 4][l & 15], &bb[l], 8);
    const int col = me->lane & 15;
    for (int reg = 0; reg < 4; ++reg) {
        const int row = (me->lane >> 4) + 4 * reg;
        double s = c[reg];
        for (int k = 0; k < 4; ++k) s = std::fma(av[row][k], bv[k][col], s);
        c[reg] = s;
    }
    return c;
}

// ---- atomics (fibers of one OS thread: plain read-modify-write is atomic between yields)
template <typename T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

// ---- host runtime (memory is host memory; streams execute immediately; events are wall-clock stamps)
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct simt_stream* hipStream_t;
typedef struct simt_event* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0, hipEventDisableTiming = 2 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t n, unsigned flags = 0);
hipError_t hipHostFree(void* p);
enum { hipHostRegisterDefault = 0 };
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }      // (host memory is host memory here)
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamCreate(hipStream_t* s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipGetLastError();
const char* hipGetErrorString(hipError_t e);
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v);

// stream capture / graphs: while a stream is capturing, launches are recorded (grid, block, LDS size, argument values) instead
// of executed; a graph launch replays them in order (other stream operations are not part of an emulated graph)
typedef struct simt_graph* hipGraph_t;
typedef struct simt_graph* hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1, hipStreamCaptureModeRelaxed = 2 };
hipError_t hipStreamBeginCapture(hipStream_t s, hipStreamCaptureMode mode);
hipError_t hipStreamEndCapture(hipStream_t s, hipGraph_t* g);
hipError_t hipGraphInstantiate(hipGraphExec_t* x, hipGraph_t g, void*, void*, size_t);
hipError_t hipGraphDestroy(hipGraph_t g);
hipError_t hipGraphExecDestroy(hipGraphExec_t x);
hipError_t hipGraphLaunch(hipGraphExec_t x, hipStream_t s);
namespace simt { bool capture_launch(hipStream_t s, dim3 grid, dim3 block, size_t lds_bytes, const std::function<void()>& body); }

// The explicit kernel arguments laid out as the kernarg segment of a real launch holds them (natural alignment, in order); the
// hidden arguments would follow: __builtin_amdgcn_implicitarg_ptr() points there, and device code that rebuilds its context from
// the kernarg segment (phase_ctx in ba_pipeline.hip) finds its arguments at the same negative offsets as on the GPU.
namespace simt {
extern thread_local const char* kernarg_end;
template <typename T> inline void pack_kernarg(std::vector<char>& b, const T& v) {
    const size_t o = (b.size() + alignof(T) - 1) / alignof(T) * alignof(T);
    b.resize(o + sizeof(T));
    memcpy(b.data() + o, &v, sizeof(T));
}
}  // namespace simt
inline const char* __builtin_amdgcn_implicitarg_ptr() { return simt::kernarg_end; }

template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args) {
    auto ka = std::make_shared<std::vector<char>>();
    (simt::pack_kernarg<KArgs>(*ka, static_cast<KArgs>(args)), ...);
    ka->resize((ka->size() + 7) / 8 * 8);
    std::function<void()> body = [=]() { simt::kernarg_end = ka->data() + ka->size(); kernel(args...); };
    if (simt::capture_launch(st, grid, block, lds, body)) return;
    simt::launch(grid, block, lds, body);
}
