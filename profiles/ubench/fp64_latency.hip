#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* cyc, double a, double b) {
    double x = a + threadIdx.x * 1e-9, y = b;
    long long t0 = clock64();
#pragma unroll 1
    for (int i = 0; i < 100; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) x = fma(x, y, a);
    }
    long long t1 = clock64();
    double z = a + threadIdx.x * 1e-9 + 2.0;
#pragma unroll 1
    for (int i = 0; i < 100; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) z = __builtin_amdgcn_rsq(z) + 2.0;
    }
    long long t2 = clock64();
    float f = (float)a + threadIdx.x * 1e-6f;
#pragma unroll 1
    for (int i = 0; i < 100; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) f = fmaf(f, (float)b, (float)a);
    }
    long long t3 = clock64();
    // 4 independent f64 chains
    double p0 = x, p1 = x + 1, p2 = x + 2, p3 = x + 3;
#pragma unroll 1
    for (int i = 0; i < 100; ++i) {
#pragma unroll
        for (int j = 0; j < 10; ++j) { p0 = fma(p0, y, a); p1 = fma(p1, y, a); p2 = fma(p2, y, a); p3 = fma(p3, y, a); }
    }
    long long t4 = clock64();
    out[threadIdx.x] = x + z + f + p0 + p1 + p2 + p3;
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}
int main() {
    double* o; long long* c;
    hipMalloc(&o, 64 * 8); hipMalloc(&c, 32);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, 0.5, 0.999);
    hipDeviceSynchronize();
    long long h[4]; hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
    printf("dependent fma_f64: %.1f cyc/op; rsq_f64+add: %.1f cyc/pair; dependent fma_f32: %.1f cyc/op; 4 independent f64 chains: %.1f cyc/op\n",
           h[0] / 1000.0, h[1] / 1000.0, h[2] / 1000.0, h[3] / 4000.0);
    return 0;
}
