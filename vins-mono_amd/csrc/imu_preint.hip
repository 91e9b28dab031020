// imu_preint.hip — batched IMU pre-integration on gfx950: IntegrationBase::push_back / propagate /
// midPointIntegration / repropagate (vins_estimator/src/factor/integration_base.h:30-158), i.e. the step in
// Estimator::processIMU (estimator.cpp:93-101) that produces the constants IMUFactor::Evaluate consumes
// (the caller-side neighbour of the BA hot path).
//
// One wavefront per frame interval (the intervals of all windows of a batch are independent).  The 15x15 jacobian
// and covariance live in LDS; per IMU sample every lane evaluates the (tiny, wave-uniform) mid-point update and the
// 3x3 blocks of F (15x15) and V (15x18) redundantly, then the lanes share the three dense products
//   jacobian <- F jacobian,   covariance <- F covariance F^T + V diag(noise) V^T
// entry-wise (225 entries over 64 lanes, k ascending like a plain triple loop).  repropagate() is the same
// computation started from the stored first sample with new linearisation biases, so one entry point serves both.
#include "vg_range.h"
#include <hip/hip_runtime.h>
#include <vector>
#include "ba_math.h"
#include "vg_handle.h"
#include "../../include/vinsgpu.h"

#define IMU_OUT 467     // sum_dt | dp 3 | dq 4 | dv 3 | ba 3 | bg 3 | jacobian 225 | covariance 225

struct ImuPreLds {
    double J[225], P[225], T[225], F[225], V[15 * 18], nz[18];
};

extern "C" __global__ __launch_bounds__(64) void imu_preint_kernel(int n, const int* __restrict__ off, const double* __restrict__ samples,
                                                                   const double* __restrict__ first, const double* __restrict__ bias,
                                                                   double acc_n, double gyr_n, double acc_w, double gyr_w,
                                                                   double* __restrict__ out) {
    __shared__ ImuPreLds s;
    const int k = blockIdx.x, lane = threadIdx.x;
    if (k >= n) return;
    for (int e = lane; e < 225; e += 64) { s.J[e] = (e / 15 == e % 15) ? 1.0 : 0.0; s.P[e] = 0.0; s.F[e] = 0.0; }
    for (int e = lane; e < 15 * 18; e += 64) s.V[e] = 0.0;
    if (lane < 18) {
        // noise = diag(ACC_N^2 I, GYR_N^2 I, ACC_N^2 I, GYR_N^2 I, ACC_W^2 I, GYR_W^2 I)  (integration_base.h:18-26)
        const int b = lane / 3;
        s.nz[lane] = (b == 0 || b == 2) ? acc_n * acc_n : ((b == 1 || b == 3) ? gyr_n * gyr_n : (b == 4 ? acc_w * acc_w : gyr_w * gyr_w));
    }
    double acc0[3] = {first[6 * k], first[6 * k + 1], first[6 * k + 2]};
    double gyr0[3] = {first[6 * k + 3], first[6 * k + 4], first[6 * k + 5]};
    const double ba[3] = {bias[6 * k], bias[6 * k + 1], bias[6 * k + 2]};
    const double bg[3] = {bias[6 * k + 3], bias[6 * k + 4], bias[6 * k + 5]};
    double dp[3] = {0, 0, 0}, dv[3] = {0, 0, 0}, dq[4] = {0, 0, 0, 1}, sum_dt = 0.0;
    __syncthreads();
    for (int si = off[k]; si < off[k + 1]; ++si) {
        const double* sm = samples + (size_t)si * 7;
        const double dt = sm[0];
        const double acc1[3] = {sm[1], sm[2], sm[3]}, gyr1[3] = {sm[4], sm[5], sm[6]};
        // ---- midPointIntegration (integration_base.h:62-72), wave-uniform
        double Rq[9], Rr[9], rq[4];
        q_to_R(dq, Rq);
        const double w[3] = {0.5 * (gyr0[0] + gyr1[0]) - bg[0], 0.5 * (gyr0[1] + gyr1[1]) - bg[1], 0.5 * (gyr0[2] + gyr1[2]) - bg[2]};
        const double hq[4] = {w[0] * dt / 2, w[1] * dt / 2, w[2] * dt / 2, 1.0};
        q_mul(dq, hq, rq);
        q_to_R(rq, Rr);                                   // un-normalised result_delta_q, as the reference uses it
        const double a0[3] = {acc0[0] - ba[0], acc0[1] - ba[1], acc0[2] - ba[2]};
        const double a1[3] = {acc1[0] - ba[0], acc1[1] - ba[1], acc1[2] - ba[2]};
        double u0[3], u1[3];
        m3_vec(Rq, a0, u0);
        m3_vec(Rr, a1, u1);
        const double ua[3] = {0.5 * (u0[0] + u1[0]), 0.5 * (u0[1] + u1[1]), 0.5 * (u0[2] + u1[2])};
        double np_[3], nv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { np_[i] = dp[i] + dv[i] * dt + 0.5 * ua[i] * dt * dt; nv[i] = dv[i] + ua[i] * dt; }
        // ---- F, V blocks (integration_base.h:76-129)
        double Rw[9], Ra0[9], Ra1[9], IRw[9], RqA0[9], RrA1[9], RrA1I[9];
        skew3(w, Rw); skew3(a0, Ra0); skew3(a1, Ra1);
#pragma unroll
        for (int i = 0; i < 9; ++i) IRw[i] = ((i % 4 == 0) ? 1.0 : 0.0) - Rw[i] * dt;
        m3_mul(Rq, Ra0, RqA0);
        m3_mul(Rr, Ra1, RrA1);
        m3_mul(RrA1, IRw, RrA1I);
        if (lane < 9) {
            const int r = lane / 3, c = lane % 3, i = lane;
            const double id = (r == c) ? 1.0 : 0.0;
            double* F = s.F;
            double* V = s.V;
            F[(0 + r) * 15 + 0 + c] = id;
            F[(0 + r) * 15 + 3 + c] = -0.25 * RqA0[i] * dt * dt + -0.25 * RrA1I[i] * dt * dt;
            F[(0 + r) * 15 + 6 + c] = id * dt;
            F[(0 + r) * 15 + 9 + c] = -0.25 * (Rq[i] + Rr[i]) * dt * dt;
            F[(0 + r) * 15 + 12 + c] = -0.25 * RrA1[i] * dt * dt * -dt;
            F[(3 + r) * 15 + 3 + c] = IRw[i];
            F[(3 + r) * 15 + 12 + c] = -1.0 * id * dt;
            F[(6 + r) * 15 + 3 + c] = -0.5 * RqA0[i] * dt + -0.5 * RrA1I[i] * dt;
            F[(6 + r) * 15 + 6 + c] = id;
            F[(6 + r) * 15 + 9 + c] = -0.5 * (Rq[i] + Rr[i]) * dt;
            F[(6 + r) * 15 + 12 + c] = -0.5 * RrA1[i] * dt * -dt;
            F[(9 + r) * 15 + 9 + c] = id;
            F[(12 + r) * 15 + 12 + c] = id;
            V[(0 + r) * 18 + 0 + c] = 0.25 * Rq[i] * dt * dt;
            V[(0 + r) * 18 + 3 + c] = 0.25 * -RrA1[i] * dt * dt * 0.5 * dt;
            V[(0 + r) * 18 + 6 + c] = 0.25 * Rr[i] * dt * dt;
            V[(0 + r) * 18 + 9 + c] = 0.25 * -RrA1[i] * dt * dt * 0.5 * dt;
            V[(3 + r) * 18 + 3 + c] = 0.5 * id * dt;
            V[(3 + r) * 18 + 9 + c] = 0.5 * id * dt;
            V[(6 + r) * 18 + 0 + c] = 0.5 * Rq[i] * dt;
            V[(6 + r) * 18 + 3 + c] = 0.5 * -RrA1[i] * dt * 0.5 * dt;
            V[(6 + r) * 18 + 6 + c] = 0.5 * Rr[i] * dt;
            V[(6 + r) * 18 + 9 + c] = 0.5 * -RrA1[i] * dt * 0.5 * dt;
            V[(9 + r) * 18 + 12 + c] = id * dt;
            V[(12 + r) * 18 + 15 + c] = id * dt;
        }
        __syncthreads();
        // ---- jacobian = F * jacobian ; T = F * covariance
        double jn[4], tn[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lane + 64 * q, ec = e < 225 ? e : 0;
            const int i = ec / 15, j = ec - 15 * i;
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int t = 0; t < 15; ++t) { const double f = s.F[i * 15 + t]; a += f * s.J[t * 15 + j]; b += f * s.P[t * 15 + j]; }
            jn[q] = a; tn[q] = b;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = lane + 64 * q; if (e < 225) { s.J[e] = jn[q]; s.T[e] = tn[q]; } }
        __syncthreads();
        // ---- covariance = T * F^T + V * noise * V^T
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = lane + 64 * q, ec = e < 225 ? e : 0;
            const int i = ec / 15, j = ec - 15 * i;
            double a = 0.0, b = 0.0;
#pragma unroll
            for (int t = 0; t < 15; ++t) a += s.T[i * 15 + t] * s.F[j * 15 + t];
#pragma unroll
            for (int t = 0; t < 18; ++t) b += s.V[i * 18 + t] * s.nz[t] * s.V[j * 18 + t];
            jn[q] = a + b;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int e = lane + 64 * q; if (e < 225) s.P[e] = jn[q]; }
        // ---- propagate() tail (integration_base.h:147-155)
#pragma unroll
        for (int i = 0; i < 3; ++i) { dp[i] = np_[i]; dv[i] = nv[i]; acc0[i] = acc1[i]; gyr0[i] = gyr1[i]; }
        dq[0] = rq[0]; dq[1] = rq[1]; dq[2] = rq[2]; dq[3] = rq[3];
        {   // Eigen normalize(): divide by the norm
            const double nrm = sqrt(dq[0] * dq[0] + dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
            dq[0] /= nrm; dq[1] /= nrm; dq[2] /= nrm; dq[3] /= nrm;
        }
        sum_dt += dt;
        __syncthreads();
    }
    double* o = out + (size_t)k * IMU_OUT;
    if (lane == 0) {
        o[0] = sum_dt;
        for (int i = 0; i < 3; ++i) { o[1 + i] = dp[i]; o[8 + i] = dv[i]; o[11 + i] = ba[i]; o[14 + i] = bg[i]; }
        for (int i = 0; i < 4; ++i) o[4 + i] = dq[i];
    }
    for (int e = lane; e < 225; e += 64) { o[17 + e] = s.J[e]; o[242 + e] = s.P[e]; }
}

// C-ABI: see include/vinsgpu.h
extern "C" int vg_imu_preintegrate(vg_handle* h, int n_intervals, const int* sample_off, const double* samples, const double* first,
                                   const double* bias, const double* noise, vg_imu_preint* out) {
    VG_RANGE("vg_imu_preintegrate");
    if (!h || n_intervals <= 0 || !sample_off || !samples || !first || !bias || !noise || !out) return VG_ERR_BAD_ARG;
    if (sample_off[0] != 0) { h->err = "vg_imu_preintegrate: sample_off[0] must be 0"; return VG_ERR_BAD_ARG; }
    for (int k = 0; k < n_intervals; ++k)
        if (sample_off[k + 1] < sample_off[k]) { h->err = "vg_imu_preintegrate: sample_off must be non-decreasing"; return VG_ERR_BAD_ARG; }
    const size_t S = (size_t)sample_off[n_intervals];
    hipError_t e = hipSetDevice(h->device);
    std::vector<double> host((size_t)n_intervals * IMU_OUT);
    auto fail = [&](hipError_t err) { h->err = std::string("vg_imu_preintegrate: ") + hipGetErrorString(err); return VG_ERR_HIP; };
    if (e != hipSuccess) return fail(e);
    // one scratch allocation per handle, grown on demand (this call sits on the per-frame path of a sequence: a hipMalloc /
    // hipFree pair per call would synchronise the device every frame): [samples | first | bias | out | offsets]
    const size_t nd = 7 * (S ? S : 1) + 12 * (size_t)n_intervals + (size_t)IMU_OUT * n_intervals;
    const size_t need = nd * sizeof(double) + sizeof(int) * ((size_t)n_intervals + 1);
    if (need > h->imu_cap) {
        if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return fail(e);
        (void)hipFree(h->imu_buf);
        h->imu_buf = nullptr; h->imu_cap = 0;
        if ((e = hipMalloc(&h->imu_buf, 2 * need)) != hipSuccess) return fail(e);
        h->imu_cap = 2 * need;
    }
    double* d_s = (double*)h->imu_buf;
    double* d_f = d_s + 7 * (S ? S : 1);
    double* d_b = d_f + 6 * (size_t)n_intervals;
    double* d_o = d_b + 6 * (size_t)n_intervals;
    int* d_off = (int*)(d_o + (size_t)IMU_OUT * n_intervals);
    if ((e = hipMemcpyAsync(d_off, sample_off, sizeof(int) * (n_intervals + 1), hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
    if (S && (e = hipMemcpyAsync(d_s, samples, sizeof(double) * 7 * S, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_f, first, sizeof(double) * 6 * n_intervals, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(d_b, bias, sizeof(double) * 6 * n_intervals, hipMemcpyHostToDevice, h->stream)) != hipSuccess) return fail(e);
    hipLaunchKernelGGL(imu_preint_kernel, dim3(n_intervals), dim3(64), 0, h->stream, n_intervals, d_off, d_s, d_f, d_b,
                       noise[0], noise[1], noise[2], noise[3], d_o);
    if ((e = hipGetLastError()) != hipSuccess) return fail(e);
    if ((e = hipMemcpyAsync(host.data(), d_o, sizeof(double) * IMU_OUT * n_intervals, hipMemcpyDeviceToHost, h->stream)) != hipSuccess) return fail(e);
    if ((e = hipStreamSynchronize(h->stream)) != hipSuccess) return fail(e);
    for (int k = 0; k < n_intervals; ++k) {
        const double* o = host.data() + (size_t)k * IMU_OUT;
        vg_imu_preint& q = out[k];
        q.sum_dt = o[0];
        for (int i = 0; i < 3; ++i) { q.delta_p[i] = o[1 + i]; q.delta_v[i] = o[8 + i]; q.linearized_ba[i] = o[11 + i]; q.linearized_bg[i] = o[14 + i]; }
        for (int i = 0; i < 4; ++i) q.delta_q[i] = o[4 + i];
        for (int i = 0; i < 225; ++i) { q.jacobian[i] = o[17 + i]; q.covariance[i] = o[242 + i]; }
        q.valid = 1;
        q._pad = 0;
    }
    return VG_OK;
}
