// What a DEPENDENT global-memory round trip costs on this box -- alone and when 256 workgroups (one per CU) ask at once -- and what
// a launch boundary costs after a kernel that left dirty lines behind.  Round 4: the boxes of the pool run every single-window BA
// launch and the streaming front-end kernels at the same speed, but launches that fill all 256 CUs with the BA kernels'
// latency-bound workgroups are 1.2 - 1.4 x slower on some of them; this program is the probe for that difference.
//   hipcc --offload-arch=gfx950 -O3 mem_latency.hip -o mem_latency && ./mem_latency        (prints one JSON line)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// one chain per workgroup: `hops` dependent 8-byte loads through the workgroup's own region (region_words entries, visited in a random
// cyclic order with a 256-byte granularity so that every hop is a new cache line and, with regions of MBs, mostly a new page)
__global__ __launch_bounds__(64) void chase_kernel(const unsigned long long* __restrict__ buf, size_t region_words, int hops, unsigned long long* out,
                                                   long long* cycles) {
    extern __shared__ char pad[];                       // (dynamic LDS only to keep one workgroup per CU)
    if (threadIdx.x != 0) return;
    const unsigned long long* p = buf + (size_t)blockIdx.x * region_words;
    unsigned long long idx = out[blockIdx.x];           // the chain continues where the previous launch stopped: every line is new
    const long long t0 = wall_clock64();
    for (int i = 0; i < hops; ++i) idx = p[idx];
    const long long t1 = wall_clock64();
    out[blockIdx.x] = idx + (pad[0] & 0);
    cycles[blockIdx.x] = t1 - t0;
}
__global__ void dirty_kernel(double* a, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i;
}
__global__ void tiny_kernel(double* a) { if (threadIdx.x == 0) a[blockIdx.x * 512] += 1.0; }

// Instruction fetch: the BA kernels are 20 - 100 KB of code that a workgroup walks ONCE (phase after phase, no long loops), the front-end
// kernels are a few KB of loops.  big_code_kernel executes BIG_N dependent FMAs as straight-line code (~10 bytes each: larger than the
// 64 KB instruction cache two CUs share), loop_kernel the same chain from a 16-instruction loop.
#define BIG_N 12288
#define F1 x = __builtin_fma(x, a, b); b = __builtin_fma(b, a, x);
#define F8 F1 F1 F1 F1 F1 F1 F1 F1
#define F64 F8 F8 F8 F8 F8 F8 F8 F8
#define F512 F64 F64 F64 F64 F64 F64 F64 F64
__global__ __launch_bounds__(64) void big_code_kernel(double* out, double a) {
    double x = a + threadIdx.x, b = a * 0.5;
    F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512 F512                      // 12 x 512 x 2 = BIG_N dependent FMAs, ~96 KB of code
    out[blockIdx.x * 64 + threadIdx.x] = x + b;
}
__global__ __launch_bounds__(64) void loop_kernel(double* out, double a, int n) {
    double x = a + threadIdx.x, b = a * 0.5;
#pragma unroll 1
    for (int i = 0; i < n; i += 16) { F8 }
    out[blockIdx.x * 64 + threadIdx.x] = x + b;
}

int main() {
    const int nwg = 256, hops = 2500;            // 3 launches x 2500 hops < 8192 lines of a region: no line is visited twice
    const size_t region_bytes = 2u << 20, region_words = region_bytes / 8, slots = region_bytes / 256;      // 2 MB per workgroup, 8192 lines
    std::vector<unsigned long long> h((size_t)nwg * region_words, 0);
    std::mt19937_64 rng(7);
    for (int w = 0; w < nwg; ++w) {
        std::vector<unsigned> order(slots);
        std::iota(order.begin(), order.end(), 0u);
        std::shuffle(order.begin() + 1, order.end(), rng);
        for (size_t k = 0; k < slots; ++k) h[(size_t)w * region_words + (size_t)order[k] * 32] = (unsigned long long)order[(k + 1) % slots] * 32;
    }
    unsigned long long *d_buf, *d_out;
    long long* d_cyc;
    double* d_big;
    const size_t big_n = (256u << 20) / 8;
    CHK(hipMalloc(&d_buf, h.size() * 8)); CHK(hipMalloc(&d_out, nwg * 8)); CHK(hipMalloc(&d_cyc, nwg * 8)); CHK(hipMalloc(&d_big, big_n * 8));
    CHK(hipMemcpy(d_buf, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CHK(hipFuncSetAttribute((const void*)chase_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    int clock_khz = 0;
    CHK(hipDeviceGetAttribute(&clock_khz, hipDeviceAttributeWallClockRate, 0));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    auto run_chase = [&](int n, double& ns_med, double& ns_max) -> int {
        std::vector<long long> cyc(n);
        CHK(hipMemset(d_out, 0, nwg * 8));
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(chase_kernel, dim3(n), dim3(64), 150 * 1024, 0, d_buf, region_words, hops, d_out, d_cyc);
            CHK(hipDeviceSynchronize());
        }
        CHK(hipMemcpy(cyc.data(), d_cyc, n * 8, hipMemcpyDeviceToHost));
        std::sort(cyc.begin(), cyc.end());
        ns_med = (double)cyc[n / 2] / hops / (clock_khz * 1e-6);
        ns_max = (double)cyc[n - 1] / hops / (clock_khz * 1e-6);
        return 0;
    };
    double one_med, one_max, all_med, all_max;
    if (run_chase(1, one_med, one_max) || run_chase(nwg, all_med, all_max)) return 1;
    // launch boundaries: 200 dependent launches of a kernel that touches one line per workgroup, with and without 256 MB of dirty lines
    // written by the launch in front of each
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    float ms_tiny = 0, ms_pair = 0, ms_dirty = 0;
    for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, 0, d_big);
    CHK(hipEventRecord(e0)); for (int k = 0; k < 200; ++k) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, 0, d_big); CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms_tiny, e0, e1));
    CHK(hipEventRecord(e0)); for (int k = 0; k < 50; ++k) hipLaunchKernelGGL(dirty_kernel, dim3(2048), dim3(256), 0, 0, d_big, big_n); CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms_dirty, e0, e1));
    CHK(hipEventRecord(e0));
    for (int k = 0; k < 50; ++k) { hipLaunchKernelGGL(dirty_kernel, dim3(2048), dim3(256), 0, 0, d_big, big_n); hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, 0, d_big); }
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms_pair, e0, e1));
    // straight-line code against a loop, 256 workgroups of one wavefront, each launch after a launch of the other kernel (cold cache)
    float ms_big = 0, ms_loop = 0;
    for (int k = 0; k < 3; ++k) { hipLaunchKernelGGL(big_code_kernel, dim3(256), dim3(64), 0, 0, d_big, 1.0000001); hipLaunchKernelGGL(loop_kernel, dim3(256), dim3(64), 0, 0, d_big, 1.0000001, BIG_N); }
    CHK(hipDeviceSynchronize());
    for (int k = 0; k < 20; ++k) {
        float t;
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL(big_code_kernel, dim3(256), dim3(64), 0, 0, d_big, 1.0000001); CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&t, e0, e1)); ms_big += t;
        CHK(hipEventRecord(e0)); hipLaunchKernelGGL(loop_kernel, dim3(256), dim3(64), 0, 0, d_big, 1.0000001, BIG_N); CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&t, e0, e1)); ms_loop += t;
    }
    printf("{\"instruction_fetch_us\": {\"straight_line_%d_fma\": %.1f, \"same_chain_from_a_loop\": %.1f}, ", BIG_N, ms_big * 1e3 / 20, ms_loop * 1e3 / 20);
    printf("\"device\": \"%s\", \"compute_units\": %d, \"clock_MHz\": %d, \"memory_clock_MHz\": %d, \"l2_MB\": %.1f, "
           "\"dependent_load_ns\": {\"one_workgroup\": %.0f, \"256_workgroups_median\": %.0f, \"256_workgroups_slowest\": %.0f, "
           "\"what\": \"2500 dependent 8-byte loads per workgroup through its own 2 MB region, a new 256-byte line per hop, never revisited (third launch timed)\"}, "
           "\"launch_us\": {\"tiny_kernel_back_to_back\": %.2f, \"write_256MB_kernel\": %.1f, \"tiny_kernel_after_a_write_256MB_kernel\": %.2f}}\n",
           prop.name, prop.multiProcessorCount, prop.clockRate / 1000, prop.memoryClockRate / 1000, prop.l2CacheSize / 1048576.0, one_med, all_med, all_max,
           ms_tiny * 1e3 / 200, ms_dirty * 1e3 / 50, (ms_pair - ms_dirty) * 1e3 / 50);
    return 0;
}
