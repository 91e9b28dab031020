// ba_math.h — small fixed-size double math for the BA kernels (device only, gfx950).
// Conventions follow vins_estimator/src/utility/utility.h of the reference: quaternions are
// stored [x y z w]; deltaQ is first-order and NOT normalised (utility.h:16-28); rotation
// perturbations are right-multiplied.
#pragma once
#include <hip/hip_runtime.h>
#include "vg_target.h"

#define DEV __device__ __forceinline__

// packed-lower-triangle index w -> (a, b), b <= a, without loops (float sqrt + one fix-up; exact for w < 2^22)
DEV void tri_decode(int w, int& a, int& b) {
    int t = (int)((sqrtf(8.0f * (float)w + 1.0f) - 1.0f) * 0.5f);
    t = (t * (t + 1) / 2 > w) ? t - 1 : t;
    t = ((t + 1) * (t + 2) / 2 <= w) ? t + 1 : t;
    a = t; b = w - t * (t + 1) / 2;
}

// ---- wavefront reductions on the DPP network (VALU only: a 64-bit __shfl_down costs two ds_bpermute round trips per
// step, ~700 cycles for a wave sum; this is ~100).  Quad swaps -> half-row mirror -> row mirror leave the 16-lane row
// total in every lane of the row; the four row totals are combined through SGPRs, so the result is in ALL lanes.
template <int CTRL> DEV double dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
DEV double readlane_f64(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// (uni() = a wave-uniform value kept in SGPRs, const_load() = a read through the constant address space: vg_target.h)
DEV double group8_sum(double v) {            // sum over aligned groups of 8 lanes, result in all 8
    v += dpp_mov_f64<0xB1>(v);               // quad_perm [1,0,3,2]
    v += dpp_mov_f64<0x4E>(v);               // quad_perm [2,3,0,1]
    v += dpp_mov_f64<0x141>(v);              // row_half_mirror
    return v;
}
DEV double wave_sum_all(double v) {
    v = group8_sum(v);
    v += dpp_mov_f64<0x140>(v);              // row_mirror
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
DEV double wave_max_all(double v) {
    v = fmax(v, dpp_mov_f64<0xB1>(v)); v = fmax(v, dpp_mov_f64<0x4E>(v));
    v = fmax(v, dpp_mov_f64<0x141>(v)); v = fmax(v, dpp_mov_f64<0x140>(v));
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

DEV void q_mul(const double* a, const double* b, double* o) {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by + ay * bw + az * bx - ax * bz;
    o[2] = aw * bz + az * bw + ax * by - ay * bx;
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
}
DEV void q_inv(const double* q, double* o) {
    // one reciprocal instead of four divisions (an f64 divide is ~40 instructions on gfx950)
    const double inv = 1.0 / (q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    o[0] = -q[0] * inv; o[1] = -q[1] * inv; o[2] = -q[2] * inv; o[3] = q[3] * inv;
}
DEV void q_normalize(double* q) {
    const double inv = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] *= inv; q[1] *= inv; q[2] *= inv; q[3] *= inv;
}
// Eigen::Quaternion::toRotationMatrix (no normalisation), row-major 3x3
DEV void q_to_R(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
// Eigen quaternion-from-rotation-matrix
DEV void R_to_q(const double* m, double* q) {
    double t = m[0] + m[4] + m[8];
    if (t > 0) {
        t = sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m[7] - m[5]) * t;
        q[1] = (m[2] - m[6]) * t;
        q[2] = (m[3] - m[1]) * t;
    } else {
        int i = 0;
        if (m[4] > m[0]) i = 1;
        if (m[8] > m[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = sqrt(m[i * 4] - m[j * 4] - m[k * 4] + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
        q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
        q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
    }
}
DEV void m3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
DEV void m3t_mul(const double* A, const double* B, double* C) {  // A^T B
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
DEV void m3_mul_t(const double* A, const double* B, double* C) {  // A B^T
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = A[i * 3] * B[j * 3] + A[i * 3 + 1] * B[j * 3 + 1] + A[i * 3 + 2] * B[j * 3 + 2];
}
DEV void m3_vec(const double* A, const double* v, double* o) {
    o[0] = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
    o[1] = A[3] * v[0] + A[4] * v[1] + A[5] * v[2];
    o[2] = A[6] * v[0] + A[7] * v[1] + A[8] * v[2];
}
DEV void m3t_vec(const double* A, const double* v, double* o) {
    o[0] = A[0] * v[0] + A[3] * v[1] + A[6] * v[2];
    o[1] = A[1] * v[0] + A[4] * v[1] + A[7] * v[2];
    o[2] = A[2] * v[0] + A[5] * v[1] + A[8] * v[2];
}
DEV void skew3(const double* v, double* S) {
    S[0] = 0;     S[1] = -v[2]; S[2] = v[1];
    S[3] = v[2];  S[4] = 0;     S[5] = -v[0];
    S[6] = -v[1]; S[7] = v[0];  S[8] = 0;
}
// bottom-right 3x3 of Qleft(q) (utility.h:51-59): w I + [v]x
DEV void qleft3(const double* q, double* M) {
    M[0] = q[3];  M[1] = -q[2]; M[2] = q[1];
    M[3] = q[2];  M[4] = q[3];  M[5] = -q[0];
    M[6] = -q[1]; M[7] = q[0];  M[8] = q[3];
}
// bottom-right 3x3 of Qleft(a) * Qright(b) (utility.h:51-68)
DEV void qleft_qright3(const double* a, const double* b, double* M) {
    double L[9], Rr[9];
    qleft3(a, L);
    Rr[0] = b[3];  Rr[1] = b[2];  Rr[2] = -b[1];
    Rr[3] = -b[2]; Rr[4] = b[3];  Rr[5] = b[0];
    Rr[6] = b[1];  Rr[7] = -b[0]; Rr[8] = b[3];
    m3_mul(L, Rr, M);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i * 3 + j] -= a[i] * b[j];
}
// PoseLocalParameterization::Plus (pose_local_parameterization.cpp:3-18)
DEV void pose_plus(const double* x, const double* d, double* o) {
    o[0] = x[0] + d[0]; o[1] = x[1] + d[1]; o[2] = x[2] + d[2];
    const double dq[4] = {d[3] / 2.0, d[4] / 2.0, d[5] / 2.0, 1.0};
    double q[4];
    q_mul(x + 3, dq, q);
    q_normalize(q);
    o[3] = q[0]; o[4] = q[1]; o[5] = q[2]; o[6] = q[3];
}
