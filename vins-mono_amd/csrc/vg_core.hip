// vg_core.hip — handle lifecycle, stream and HIP-event stopwatch of the C-ABI (include/vinsgpu.h).
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "vg_handle.h"
#include "vg_target.h"
#include "../../include/vinsgpu.h"

extern "C" void fe_state_destroy(FeState* s);
extern "C" void ba_graph_release(vg_handle* h);
extern "C" void ba_seq_release(vg_handle* h);
// (weak: builds without csrc/vg_rccl.hip — the CPU emulation of tests/simt — have no communicator to release)
extern "C" int vg_ba_rccl_finalize(vg_handle* h) __attribute__((weak));

extern "C" int vg_abi_version(void) { return VG_ABI_VERSION; }

extern "C" int vg_create(vg_handle** out) {
    if (!out) return VG_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return VG_ERR_NO_DEVICE;
    vg_handle* h = new vg_handle();
    if (hipGetDevice(&h->device) != hipSuccess) { delete h; return VG_ERR_HIP; }
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess ||
        hipEventCreate(&h->ev2) != hipSuccess || hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) {
        delete h;
        return VG_ERR_HIP;
    }
    *out = h;
    return VG_OK;
}

extern "C" int vg_create_config(const vg_config* cfg, vg_handle** out) {
    if (!cfg || !out || cfg->struct_size < (int)(2 * sizeof(int))) return VG_ERR_BAD_ARG;
    vg_config c;
    memset(&c, 0, sizeof(c));                                  // fields beyond the caller's struct_size: their defaults (all zero)
    memcpy(&c, cfg, (size_t)cfg->struct_size < sizeof(c) ? (size_t)cfg->struct_size : sizeof(c));
    if (c.launch_mode != 0 && c.launch_mode != VG_LAUNCH_GRAPH + 1 && c.launch_mode != VG_LAUNCH_DIRECT + 1) return VG_ERR_BAD_ARG;
    if (c.marg_mode != VG_MARG_SQRT && c.marg_mode != VG_MARG_EIGEN) return VG_ERR_BAD_ARG;
    if (c.fused_min_windows < -1 || c.pack_threads < 0 || c.pack_threads > 64) return VG_ERR_BAD_ARG;
    if (c.imu_info_mode != VG_IMU_INFO_FACTOR && c.imu_info_mode != VG_IMU_INFO_REFERENCE) return VG_ERR_BAD_ARG;
    // device: 0 (what a zero-initialised struct holds) and negative values = the CURRENT device -- a rank that chose its GPU with
    // hipSetDevice(local_rank) stays there --, k + 1 = device k (the encoding of launch_mode; ABI 12)
    if (c.device > 0) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return VG_ERR_NO_DEVICE;
        if (c.device - 1 >= ndev) return VG_ERR_BAD_ARG;
        if (hipSetDevice(c.device - 1) != hipSuccess) return VG_ERR_HIP;
    }
    const int rc = vg_create(out);
    if (rc != VG_OK) return rc;
    vg_handle* h = *out;
    h->ba.no_env = true;                                        // every switch below is explicit: the environment is not consulted
    h->ba.launch_mode = c.launch_mode ? c.launch_mode - 1 : VG_LAUNCH_DEFAULT;
    h->ba.marg_mode = c.marg_mode;
    h->ba.imu_info_mode = c.imu_info_mode;
    h->ba.fused_min = c.fused_min_windows == 0 ? 32 : (c.fused_min_windows < 0 ? 0 : c.fused_min_windows);
    h->ba.pack_threads = c.pack_threads ? c.pack_threads : 8;
    return VG_OK;
}

extern "C" int vg_destroy(vg_handle* h) {
    if (!h) return VG_ERR_BAD_ARG;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    if (vg_ba_rccl_finalize) (void)vg_ba_rccl_finalize(h);
    ba_graph_release(h);
    ba_seq_release(h);
    BaPtrs& P = h->ba.P;
    (void)hipFree(P.iarr); (void)hipFree(P.din); (void)hipFree(P.scr); (void)hipFree(P.out); (void)hipFree(P.iout);
    (void)hipFree(h->ba.dL);
    if (h->ba.h_ia) (void)hipHostFree(h->ba.h_ia);
    if (h->ba.h_di) (void)hipHostFree(h->ba.h_di);
    if (h->ba.h_pri) (void)hipHostFree(h->ba.h_pri);
    (void)hipFree(P.pri);
    h->ba.h_out.release(); h->ba.h_iout.release(); h->ba.h_mout.release(); h->ba.h_miout.release();
    (void)hipFree(P.mout); (void)hipFree(P.miout); (void)hipFree(P.mscr); (void)hipFree(P.rb1); (void)hipFree(P.rb2);
    if (h->fe) fe_state_destroy(h->fe);
    (void)hipFree(h->ransac_buf);
    (void)hipFree(h->imu_buf);
    (void)hipEventDestroy(h->ev0); (void)hipEventDestroy(h->ev1); (void)hipEventDestroy(h->ev2);
    (void)hipEventDestroy(h->ev_fork); (void)hipEventDestroy(h->ev_join);
    (void)hipStreamDestroy(h->aux);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return VG_OK;
}

extern "C" int vg_sync(vg_handle* h) {
    if (!h) return VG_ERR_BAD_ARG;
    hipError_t e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}

extern "C" const char* vg_last_error(vg_handle* h) { return h ? h->err.c_str() : "null handle"; }
extern "C" int vg_host_register(vg_handle* h, void* p, size_t bytes) {
    if (!h || !p || !bytes) return VG_ERR_BAD_ARG;
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { h->err = std::string("vg_host_register: ") + hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}
extern "C" int vg_host_unregister(vg_handle* h, void* p) {
    if (!h || !p) return VG_ERR_BAD_ARG;
    const hipError_t e = hipHostUnregister(p);
    if (e != hipSuccess) { h->err = std::string("vg_host_unregister: ") + hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}
extern "C" void* vg_stream(vg_handle* h) { return h ? (void*)h->stream : nullptr; }

extern "C" int vg_timer_start(vg_handle* h) {
    if (!h) return VG_ERR_BAD_ARG;
    hipError_t e = hipEventRecord(h->ev0, h->stream);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}

extern "C" int vg_timer_stop(vg_handle* h, float* ms) {
    if (!h || !ms) return VG_ERR_BAD_ARG;
    hipError_t e = hipEventRecord(h->ev1, h->stream);
    if (e == hipSuccess) e = hipEventSynchronize(h->ev1);
    if (e == hipSuccess) e = hipEventElapsedTime(ms, h->ev0, h->ev1);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return VG_ERR_HIP; }
    return VG_OK;
}

// ---- what kind of box is this?  (boxes of the pool run the latency-bound BA kernels up to 1.4 x apart although they
//      report the same clocks.)  A dependent chain of FP64 FMAs has a fixed latency in CORE cycles, so its wall time (the constant
//      100 MHz counter) measures the core clock the box really runs at -- once with one lone wavefront, once with every CU loaded
//      (one workgroup of 4 wavefronts per CU, i.e. still one wavefront per SIMD: a power / clock cap shows as a longer chain) -- and the ratio of the shader-clock counter
//      to the wall counter says whether clock64() follows that clock.
__global__ void vg_probe_kernel(double* out, int n) {
    double x = 1.0 + (double)threadIdx.x * 1e-12;
    const double a = 1.0000000001, b = 1e-12;
    const long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < n; i += 8) {
        x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b);
        x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b); x = fma(x, a, b);
    }
    const long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) {
        out[4 * blockIdx.x + 0] = (double)(w1 - w0);
        out[4 * blockIdx.x + 1] = (double)(c1 - c0);
        out[4 * blockIdx.x + 2] = x;
    }
}
extern "C" int vg_probe_clocks(vg_handle* h, double* out4) {
    if (!h || !out4) return VG_ERR_BAD_ARG;
    const int n = 1 << 18, nblk = 256;
    double* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(double) * 4 * nblk) != hipSuccess) { h->err = "vg_probe_clocks: hipMalloc"; return VG_ERR_HIP; }
    std::vector<double> host(4 * nblk);
    int rc = VG_OK;
    for (int pass = 0; pass < 2 && rc == VG_OK; ++pass) {
        const int g = pass ? nblk : 1, t = pass ? 256 : 64;
        hipLaunchKernelGGL(vg_probe_kernel, dim3(g), dim3(t), 0, h->stream, d, 1024);          // warm-up (code fetch)
        hipLaunchKernelGGL(vg_probe_kernel, dim3(g), dim3(t), 0, h->stream, d, n);
        if (hipMemcpyAsync(host.data(), d, sizeof(double) * 4 * g, hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { h->err = "vg_probe_clocks: launch / copy failed"; rc = VG_ERR_HIP; break; }
        double wall = 0.0, clk = 0.0;
        for (int b = 0; b < g; ++b) { wall += host[4 * b]; clk += host[4 * b + 1]; }
        wall /= g; clk /= g;
        out4[2 * pass + 0] = wall / BA_WALL_HZ * 1e9 / n;          // ns per dependent FP64 FMA
        out4[2 * pass + 1] = wall > 0 ? clk / (wall / BA_WALL_HZ * 1e6) : 0.0;      // clock64() ticks per microsecond of wall time
    }
    (void)hipFree(d);
    return rc;
}
