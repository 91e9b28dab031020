#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#define DEV __device__ __forceinline__
DEV double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
DEV double mg_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    y = y * fma(-hx * y, y, 1.5);
    return y;
}
__global__ void k(const double* din, const double* ein, double* dout, long long* cyc, int n, int reps) {
    __shared__ double dd[128], ee[128], dd2[128], ee2[128];
    __shared__ double2 rot[128];
    const int lane = threadIdx.x;
    dd[lane] = din[lane]; dd[lane + 64] = din[lane + 64]; ee[lane] = ein[lane]; ee[lane + 64] = ein[lane + 64];
    __syncthreads();
    long long tot = 0; int nrot = 0;
    for (int rep = 0; rep < reps; ++rep) {
        const double D0 = dd[lane], D1 = dd[lane + 64], E0 = ee[lane], E1 = ee[lane + 64];
        dd2[lane] = D0; dd2[lane + 64] = D1; ee2[lane] = E0; ee2[lane + 64] = E1;
        __syncthreads();
        auto rd = [&](double r0, double r1, int idx) { const int li = idx & 63; const double a0 = readlane_f64(r0, li), a1 = readlane_f64(r1, li); return idx < 64 ? a0 : a1; };
        const int l = __builtin_amdgcn_readfirstlane(rep % 7), m = __builtin_amdgcn_readfirstlane(n - 1 - (rep % 5));
        const long long t0 = clock64();
        double g = 0.3 + rd(D0, D1, m) * 1e-3, s = 1.0, cc = 1.0, p = 0.0;
        double e_i = rd(E0, E1, m - 1), d_i = rd(D0, D1, m - 1), d_ip1 = rd(D0, D1, m);
        const bool l0 = lane == 0;
        for (int i = m - 1; i >= l; --i) {
            const int ip = i > l ? i - 1 : i;
            const double e_n = ee2[ip], d_n = dd2[ip];
            const double f = s * e_i, b = cc * e_i, b2 = b + b;
            const double h2 = fma(f, f, g * g);
            if (__ballot(h2 < 1e-290) != 0ull) break;
            double ir = __builtin_amdgcn_rsq(h2);
            { const double t = h2 * ir, e1 = fma(-t, ir, 1.0), q = fma(0.375, e1, 0.5); ir = fma(ir * e1, q, ir); }
            s = f * ir; cc = g * ir;
            const double gp = d_ip1 - p;
            const double r = fma(cc, b2, (d_i - gp) * s);
            p = s * r;
            g = fma(cc, r, -b);
            if (l0) { ee[i + 1] = h2 * ir; dd[i + 1] = gp + p; rot[i] = make_double2(cc, s); }
            d_ip1 = d_i; d_i = d_n; e_i = e_n;
        }
        tot += clock64() - t0; nrot += m - l;
        __syncthreads();
    }
    if (lane == 0) { cyc[0] = tot; cyc[1] = nrot; dout[0] = dd[3] + ee[5] + rot[7].x; }
}
int main() {
    const int n = 75;
    double hd[128], he[128];
    for (int i = 0; i < 128; ++i) { hd[i] = 1.0 + 0.37 * i; he[i] = 0.5 + 0.01 * i; }
    double *d, *e, *o; long long* c;
    hipMalloc(&d, 1024); hipMalloc(&e, 1024); hipMalloc(&o, 64); hipMalloc(&c, 64);
    hipMemcpy(d, hd, 1024, hipMemcpyHostToDevice); hipMemcpy(e, he, 1024, hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e, o, c, n, 40);
    hipDeviceSynchronize();
    long long h[2]; hipMemcpy(h, c, 16, hipMemcpyDeviceToHost);
    printf("chase: %lld cycles for %lld rotations = %.1f cycles/rotation\n", h[0], h[1], (double)h[0] / h[1]);
    return 0;
}
