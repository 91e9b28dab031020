// ba_marg.hip — marginalization step of Estimator::optimization() (estimator.cpp:825-1000) on gfx950,
// one workgroup per window, launched right after ba_solve_kernel on the same stream.
//
// Restates MarginalizationInfo::{addResidualBlockInfo, preMarginalize, marginalize, getParameterBlocks}
// (factor/marginalization_factor.cpp:89-319) and ResidualBlockInfo::Evaluate's loss correction (:37-68):
//   M1  re-evaluate the factors to be marginalised at the post-solve (gauge-fixed) state
//   M2  index map: dropped blocks first (pose, speed-bias, landmarks ascending), kept after
//       (poses ascending, speed-biases, extrinsic, td) — the reference's order is the iteration order of an
//       unordered_map keyed by addresses, i.e. arbitrary; this canonical order is documented in DESIGN.md
//   M3  A = sum J^T J, b = sum J^T r  (owner threads, no atomics -> bit-reproducible)
//   M4  Amm^+ by symmetric eigen-decomposition with the eps = 1e-8 cut, Schur complement, second
//       eigen-decomposition -> linearized_jacobians / linearized_residuals
//   M5  kept-block list re-labelled for the slid window (addr_shift, estimator.cpp:913-930 / :969-996)
// The eigen-solver is a parallel two-sided Jacobi (round-robin pairing, 2x2 block owners) held in LDS.
#include <hip/hip_runtime.h>
#include <mutex>
#include "ba_layout.h"
#include "ba_factors.h"
#include "../../include/vinsgpu.h"

#ifndef MG_NT
#define MG_NT 512                  // round 6: 8 wavefronts x 256 VGPRs (no spills in the factor evaluations; the long phases are one-wavefront chains or MFMA tiles now: 190 -> 166 us; 1024 threads until then, -DMG_NT=1024 still builds)
#endif
#define MG_NW (MG_NT / 64)
static_assert(MG_NT / 64 > 4, "ba_marg_kernel (c): four wavefronts share the all-factor product, the others take the target frames");
#define MG_T0W 4                   // wavefronts that share the all-factor product of the projection part (ba_marg_kernel (c))
#define MG_EPS 1e-8
#define MG_MAXSWEEP 30

extern __shared__ __attribute__((aligned(16))) char mg_smem[];
#define MG_LDS ((double*)mg_smem)

struct MCtx {
    const BaLayout* Lp;
    const int* ia; const int* hdr; const double* di; const double* pri; double* sc; double* ms; double* lds;
    int tid, lane, wave;
    double focal, tr, row, gnorm;
};

// 1/sqrt(x): hardware seed + two Newton steps
DEV double mg_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * fma(-hx * y, y, 1.5);
    y = y * fma(-hx * y, y, 1.5);
    return y;
}
DEV double mg_wave_sum(double v) { return wave_sum_all(v); }
DEV double mg_block_sum(const MCtx& c, double* red, double v) {
    v = mg_wave_sum(v);
    __syncthreads();
    if (c.lane == 0) red[c.wave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < MG_NW; ++w) s += red[w];
    return s;
}

// ------------------------------------------------------------------------------------------------
// Parallel two-sided Jacobi eigen-decomposition of the symmetric n x n matrix M (leading dimension ld).
// On exit diag(M) = eigenvalues, V (ld x ld, row-major) holds the eigenvectors as COLUMNS.
// `cs` is an LDS scratch of 2*ld doubles, `red` of MG_NW doubles.
// INLDS = true: M, V are offsets (in doubles) into the dynamic LDS block (address space known to the compiler ->
// ds_read / ds_write); INLDS = false: generic pointers to the global-memory fallback buffers.
template <bool INLDS>
DEV int jacobi_eig(const MCtx& c, double* Mg, double* Vg, int offM, int offV, int n, int ld, int offcs, int offred, bool relative) {
    double* M = INLDS ? (MG_LDS + offM) : Mg;
    double* V = INLDS ? (MG_LDS + offV) : Vg;
    double* cs = MG_LDS + offcs;
    double* red = MG_LDS + offred;
    const int N = (n + 1) & ~1;           // even player count (a padding player has zero row/col)
    const int half = N / 2;
    for (int k = c.tid; k < ld * ld; k += MG_NT) {
        const int i = k / ld, j = k % ld;
        V[k] = (i == j) ? 1.0 : 0.0;
        if (i >= n || j >= n) M[k] = 0.0;
    }
    __syncthreads();
    double fro = 0.0;
    for (int k = c.tid; k < n * n; k += MG_NT) { const double v = M[(k / n) * ld + k % n]; fro += v * v; }
    fro = mg_block_sum(c, red, fro);
    // stopping rule: a pair is rotated unless |a_pq| <= 1e-16 sqrt(|a_pp a_qq|) (relative accuracy for the small
    // eigenvalues, which the eps = 1e-8 cut and the 1/lambda of the pseudo-inverse are sensitive to) or
    // |a_pq| <= 1e-17 ||A||_F (absolute floor, 20x below the rounding noise of the entries, for null directions); stop when a sweep rotates nothing.
    // relative = true : also rotate while |a_pq| > 1e-16 sqrt(|a_pp a_qq|) (high RELATIVE accuracy of the small
    //                   eigenvalues: needed for Amm, whose spectrum spans 1e2 .. 1e12 and gets inverted);
    // relative = false: absolute criterion only, |a_pq| <= 1e-16 ||A||_F — what Eigen's tridiagonal-QL solver of the
    //                   reference guarantees — which saves ~1/3 of the sweeps on the kept block.
    const double floor_abs = sqrt(fro) * (relative ? 1e-17 : 1e-16);
    const double relf2 = relative ? 1e-32 : 0.0;
    if (c.tid == 0) { red[20] = red[21] = red[22] = red[23] = 0.0; }
    int sweep = 0;
    for (; sweep < MG_MAXSWEEP; ++sweep) {
        double nrot = 0.0;
        for (int r = 0; r < N - 1; ++r) {
#ifdef BA_PROFILE
            const long long _ta = clock64();
#endif
            // pair k of this round: (p,q), p < q; table in LDS so that the owners below need no div / mod
            int* pq = (int*)(cs + 2 * half);          // [2*half] ints
            for (int k = c.tid; k < half; k += MG_NT) {
                int p, q;
                if (k == 0) { p = N - 1; q = r; }
                else { p = r + k; if (p >= N - 1) p -= N - 1; q = r - k; if (q < 0) q += N - 1; }
                if (p > q) { const int t = p; p = q; q = t; }
                double cc = 1.0, ss = 0.0;
                if (q < n) {
                    const double apq = M[q * ld + p], app = M[p * ld + p], aqq = M[q * ld + q];
                    if (fabs(apq) > floor_abs && apq * apq > relf2 * fabs(app * aqq)) {
                        // inner rotation (|angle| <= pi/4) from cos 2phi = |d| / r, sin 2phi = |apq| / r: two rsqrt
                        // chains and no division (an f64 divide is ~40 instructions on this pipe)
                        const double d = 0.5 * (aqq - app);
                        const double ir = mg_rsqrt(fma(d, d, apq * apq));
                        const double hc = fma(0.5 * fabs(d), ir, 0.5);            // cos^2 phi in [0.5, 1]
                        const double ic = mg_rsqrt(hc);
                        cc = hc * ic;
                        ss = (d >= 0 ? 0.5 : -0.5) * apq * ir * ic;
                        nrot += 1.0;
                    }
                }
                cs[2 * k] = cc; cs[2 * k + 1] = ss;
                pq[2 * k] = p; pq[2 * k + 1] = q;
            }
            __syncthreads();
#ifdef BA_PROFILE
            const long long _tb = clock64();
#endif
            if (INLDS) {
                // SYMMETRIC storage: only M[max(i,j)][min(i,j)] is kept current.  One thread per 2x2 block (ka >= kb)
                // of the pair grid: half(half+1)/2 blocks (741 for n = 75: a single pass of the 1024 threads), all
                // loads first, then the arithmetic, then the stores; blocks whose two rotations are both the identity
                // are skipped (most of the late sweeps).
                const int nblk2 = half * (half + 1) / 2;
                const unsigned inv_half = 0xFFFFFFFFu / (unsigned)half + 1u;
                const int2* pq2 = (const int2*)pq;
                const double2* cs2 = (const double2*)cs;
                if (nblk2 <= MG_NT && n * half <= 3 * MG_NT) {
                    // one M block and up to three V items per thread; three LDS round trips per round in total:
                    // (1) pair tables, (2) all matrix entries (unconditional, clamped), (3) the stores
                    const bool actM = c.tid < nblk2;
                    int ka, kb;
                    tri_decode(actM ? c.tid : 0, ka, kb);
                    const double2 ra = cs2[ka], rb = cs2[kb];
                    const int2 pa = pq2[ka], pb = pq2[kb];
                    int vi[3], vk[3];
                    double2 rv[3];
                    int2 pv[3];
                    bool actV[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const int w = c.tid + t * MG_NT;
                        actV[t] = w < n * half;
                        const int wc = actV[t] ? w : 0;
                        vi[t] = (int)__umulhi((unsigned)wc, inv_half);
                        vk[t] = wc - vi[t] * half;
                        rv[t] = cs2[vk[t]];
                        pv[t] = pq2[vk[t]];
                    }
                    const int p = pa.x, q = pa.y, rr = pb.x, s = pb.y;
                    const bool qv = q < n, sv = s < n;
                    const int qc = qv ? q : p, sc_ = sv ? s : rr;
                    const int a00 = p > rr ? p * ld + rr : rr * ld + p;
                    const int a01 = p > sc_ ? p * ld + sc_ : sc_ * ld + p;
                    const int a10 = qc > rr ? qc * ld + rr : rr * ld + qc;
                    const int a11 = qc > sc_ ? qc * ld + sc_ : sc_ * ld + qc;
                    const double m00 = M[a00], l01 = M[a01], l10 = M[a10], l11 = M[a11];
                    double v0[3], v1[3];
                    int av0[3], av1[3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const bool ok = pv[t].y < n;
                        av0[t] = vi[t] * ld + pv[t].x;
                        av1[t] = vi[t] * ld + (ok ? pv[t].y : pv[t].x);
                        actV[t] = actV[t] && ok && rv[t].y != 0.0;
                        v0[t] = V[av0[t]]; v1[t] = V[av1[t]];
                    }
                    {
                        const double ca = ra.x, sa = ra.y, cb = rb.x, sb = rb.y;
                        const double x01 = sv ? l01 : 0.0, x10 = qv ? l10 : 0.0, x11 = (qv && sv) ? l11 : 0.0;
                        const double t00 = ca * m00 - sa * x10, t01 = ca * x01 - sa * x11;
                        const double t10 = sa * m00 + ca * x10, t11 = sa * x01 + ca * x11;
                        const double n00 = cb * t00 - sb * t01;
                        double n01 = sb * t00 + cb * t01, n10 = cb * t10 - sb * t11;
                        const double n11 = sb * t10 + cb * t11;
                        if (ka == kb) { n01 = 0.0; n10 = 0.0; }
                        if (actM && (sa != 0.0 || sb != 0.0)) {
                            M[a00] = n00;
                            if (sv) M[a01] = n01;
                            if (qv) M[a10] = n10;
                            if (qv && sv) M[a11] = n11;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        if (actV[t]) {
                            V[av0[t]] = rv[t].x * v0[t] - rv[t].y * v1[t];
                            V[av1[t]] = rv[t].y * v0[t] + rv[t].x * v1[t];
                        }
                    }
                } else {
                for (int w = c.tid; w < nblk2; w += MG_NT) {
                    int ka, kb;
                    tri_decode(w, ka, kb);
                    const double ca = cs[2 * ka], sa = cs[2 * ka + 1], cb = cs[2 * kb], sb = cs[2 * kb + 1];
                    if (sa == 0.0 && sb == 0.0) continue;
                    const int p = pq[2 * ka], q = pq[2 * ka + 1], rr = pq[2 * kb], s = pq[2 * kb + 1];
                    const bool qv = q < n, sv = s < n;
                    const int qc = qv ? q : p, sc_ = sv ? s : rr;
                    const int a00 = p > rr ? p * ld + rr : rr * ld + p;
                    const int a01 = p > sc_ ? p * ld + sc_ : sc_ * ld + p;
                    const int a10 = qc > rr ? qc * ld + rr : rr * ld + qc;
                    const int a11 = qc > sc_ ? qc * ld + sc_ : sc_ * ld + qc;
                    const double m00 = M[a00], l01 = M[a01], l10 = M[a10], l11 = M[a11];
                    const double x01 = sv ? l01 : 0.0, x10 = qv ? l10 : 0.0, x11 = (qv && sv) ? l11 : 0.0;
                    const double t00 = ca * m00 - sa * x10, t01 = ca * x01 - sa * x11;
                    const double t10 = sa * m00 + ca * x10, t11 = sa * x01 + ca * x11;
                    const double n00 = cb * t00 - sb * t01;
                    double n01 = sb * t00 + cb * t01, n10 = cb * t10 - sb * t11;
                    const double n11 = sb * t10 + cb * t11;
                    if (ka == kb) { n01 = 0.0; n10 = 0.0; }
                    M[a00] = n00;
                    if (sv) M[a01] = n01;
                    if (qv) M[a10] = n10;
                    if (qv && sv) M[a11] = n11;
                }
                // V <- V J_b : one thread per (row, pair)
                for (int w = c.tid; w < n * half; w += MG_NT) {
                    const int i = (int)__umulhi((unsigned)w, inv_half);
                    const int kb = w - i * half;
                    const double cb = cs[2 * kb], sb = cs[2 * kb + 1];
                    const int rr = pq[2 * kb], s = pq[2 * kb + 1];
                    if (sb == 0.0 || s >= n) continue;
                    const double v0 = V[i * ld + rr], v1 = V[i * ld + s];
                    V[i * ld + rr] = cb * v0 - sb * v1;
                    V[i * ld + s] = sb * v0 + cb * v1;
                }
                }
            } else {
                for (int ka = c.wave; ka < half; ka += MG_NW) {
                    const int p = pq[2 * ka], q = pq[2 * ka + 1];
                    const double ca = cs[2 * ka], sa = cs[2 * ka + 1];
                    const bool qv = q < n;
                    for (int kb = c.lane; kb < half; kb += 64) {
                        const int rr = pq[2 * kb], s = pq[2 * kb + 1];
                        const double cb = cs[2 * kb], sb = cs[2 * kb + 1];
                        const bool sv = s < n;
                        const int qc = qv ? q : p, sc_ = sv ? s : rr;
                        double m00 = M[p * ld + rr], m01 = M[p * ld + sc_], m10 = M[qc * ld + rr], m11 = M[qc * ld + sc_];
                        m01 = sv ? m01 : 0.0; m10 = qv ? m10 : 0.0; m11 = (qv && sv) ? m11 : 0.0;
                        const double t00 = ca * m00 - sa * m10, t01 = ca * m01 - sa * m11;
                        const double t10 = sa * m00 + ca * m10, t11 = sa * m01 + ca * m11;
                        double n00 = cb * t00 - sb * t01, n01 = sb * t00 + cb * t01;
                        double n10 = cb * t10 - sb * t11, n11 = sb * t10 + cb * t11;
                        if (ka == kb) { n01 = 0.0; n10 = 0.0; }
                        M[p * ld + rr] = n00;
                        if (sv) M[p * ld + s] = n01;
                        if (qv) M[q * ld + rr] = n10;
                        if (qv && sv) M[q * ld + s] = n11;
                    }
                }
                for (int i = c.wave; i < n; i += MG_NW) {
                    for (int kb = c.lane; kb < half; kb += 64) {
                        const int rr = pq[2 * kb], s = pq[2 * kb + 1];
                        if (s >= n) continue;
                        const double cb = cs[2 * kb], sb = cs[2 * kb + 1];
                        const double v0 = V[i * ld + rr], v1 = V[i * ld + s];
                        V[i * ld + rr] = cb * v0 - sb * v1;
                        V[i * ld + s] = sb * v0 + cb * v1;
                    }
                }
            }
#ifdef BA_PROFILE
            const long long _td = clock64();
#endif
            __syncthreads();
#ifdef BA_PROFILE
            if (c.tid == 0) { red[20] += (double)(_tb - _ta); red[21] += (double)(_td - _tb); red[23] += (double)(clock64() - _td); }
#endif
        }
        if (mg_block_sum(c, red, nrot) == 0.0) break;
    }
    __syncthreads();
    return sweep;
}

// ------------------------------------------------------------------------------------------------
// Fast path of the above for matrices resident in LDS with half(half+1)/2 <= MG_NT and n*half <= 4*(MG_NT-64)
// (n <= 87): same rotations, same round-robin order, same stopping rule, but software-pipelined:
//     [all waves]  M <- J_a^T M J_b   for round r          (one 2x2 block per thread, symmetric storage)
//     barrier
//     [wave 0]     rotations of round r+1 from the updated M   ||   [waves 1..15]  V <- V J_b of round r
//     barrier
// so the serial rotation chain (3 LDS reads -> 2 rsqrt chains -> store) hides behind the V update.  Rotation /
// pair tables are double-buffered in `cs` (needs 2 * 3 * half doubles).
DEV int jacobi_eig_fast(const MCtx& c, int offM, int offV, int n, int ld, int offcs, int offred, bool relative) {
    double* M = MG_LDS + offM;
    double* V = MG_LDS + offV;
    double* red = MG_LDS + offred;
    const int N = (n + 1) & ~1;
    const int half = N / 2;
    const int tstride = (3 * half + 2) & ~1;            // doubles per table buffer: [cs 2*half][pq half][flag], even
    for (int k = c.tid; k < ld * ld; k += MG_NT) {
        const int i = k / ld, j = k % ld;
        V[k] = (i == j) ? 1.0 : 0.0;
        if (i >= n || j >= n) M[k] = 0.0;
    }
    __syncthreads();
    double fro = 0.0;
    for (int k = c.tid; k < n * n; k += MG_NT) { const double v = M[(k / n) * ld + k % n]; fro += v * v; }
    fro = mg_block_sum(c, red, fro);
    const double floor_abs = sqrt(fro) * (relative ? 1e-17 : 1e-16);
    const double relf2 = relative ? 1e-32 : 0.0;
    const int nblk2 = half * (half + 1) / 2;
    const int nround = N - 1;
    double nrot = 0.0;

    // ---- round-invariant work items: the VALU issue slots of the 16 waves are what bounds a round, so everything
    //      that does not depend on the round's pairing is decoded once here
    const bool actM = c.tid < nblk2;
    int ka, kb;
    tri_decode(actM ? c.tid : 0, ka, kb);
    const bool diag = ka == kb;
    int vrow[4], vkb[4];
    bool vact[4];
    {
        const unsigned inv_half = 0xFFFFFFFFu / (unsigned)half + 1u;
        const int t0 = c.tid - 64;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int w = t0 + t * (MG_NT - 64);
            vact[t] = t0 >= 0 && w < n * half;
            const int wc = vact[t] ? w : 0;
            const int i = (int)__umulhi((unsigned)wc, inv_half);
            vkb[t] = wc - i * half;
            vrow[t] = i * ld;
        }
    }

    // rotations + pair table of round r into buffer b (wave 0 only).  An invalid partner (the padding player of an
    // odd n) is stored as q = p with the identity rotation: loads stay in range, stores are predicated on q != p.
    auto rotations = [&](int r, int b) {
        double* csb = MG_LDS + offcs + b * tstride;
        int* pqb = (int*)(csb + 2 * half);
        const int k = c.lane;
        double ss = 0.0;
        if (k < half) {
            int p, q;
            if (k == 0) { p = N - 1; q = r; }
            else { p = r + k; if (p >= N - 1) p -= N - 1; q = r - k; if (q < 0) q += N - 1; }
            if (p > q) { const int t = p; p = q; q = t; }
            const int qc = q < n ? q : p;
            const double apq = M[qc * ld + p], app = M[p * ld + p], aqq = M[qc * ld + qc];
            double cc = 1.0;
            if (q < n && fabs(apq) > floor_abs && apq * apq > relf2 * fabs(app * aqq)) {
                const double d = 0.5 * (aqq - app);
                const double ir = mg_rsqrt(fma(d, d, apq * apq));
                const double hc = fma(0.5 * fabs(d), ir, 0.5);
                const double ic = mg_rsqrt(hc);
                cc = hc * ic;
                ss = (d >= 0 ? 0.5 : -0.5) * apq * ir * ic;
                nrot += 1.0;
            }
            ((double2*)csb)[k] = make_double2(cc, ss);
            ((int2*)pqb)[k] = make_int2(p, qc);
        }
        const unsigned long long any = __ballot(ss != 0.0);
        if (k == 0) pqb[2 * half] = any != 0ull;
    };
    if (c.wave == 0) rotations(0, 0);
    __syncthreads();
    int sweep = 0;
#ifdef BA_PROFILE
    if (c.tid == 0) for (int k = 20; k < 28; ++k) red[k] = 0.0;
    long long _p0, _p1, _p2, _p3, _p4;
#define JP(v) v = clock64()
#else
#define JP(v)
#endif
    for (int it = 0;; ++it) {
        JP(_p0);
        const int b = it & 1;
        const double2* cs2 = (const double2*)(MG_LDS + offcs + b * tstride);
        const int2* pq2 = (const int2*)(MG_LDS + offcs + b * tstride + 2 * half);
        const int any = __builtin_amdgcn_readfirstlane(((const int*)pq2)[2 * half]);   // a round without rotations is skipped
        if (any && actM) {
            // ---- M <- J_a^T M J_b on the block (ka >= kb), symmetric storage
            const double2 ra = cs2[ka], rb = cs2[kb];
            const int2 pa = pq2[ka], pb = pq2[kb];
            const int a00 = max(pa.x, pb.x) * ld + min(pa.x, pb.x);
            const int a01 = max(pa.x, pb.y) * ld + min(pa.x, pb.y);
            const int a10 = max(pa.y, pb.x) * ld + min(pa.y, pb.x);
            const int a11 = max(pa.y, pb.y) * ld + min(pa.y, pb.y);
            const double m00 = M[a00], m01 = M[a01], m10 = M[a10], m11 = M[a11];
            const double ca = ra.x, sa = ra.y, cb = rb.x, sb = rb.y;
            const double t00 = ca * m00 - sa * m10, t01 = ca * m01 - sa * m11;
            const double t10 = sa * m00 + ca * m10, t11 = sa * m01 + ca * m11;
            const double n00 = cb * t00 - sb * t01, n11 = sb * t10 + cb * t11;
            const double n01 = diag ? 0.0 : sb * t00 + cb * t01;
            const double n10 = diag ? 0.0 : cb * t10 - sb * t11;
            const bool qv = pa.y != pa.x, sv = pb.y != pb.x;
            if (sa != 0.0 || sb != 0.0) {
                M[a00] = n00;
                if (sv) M[a01] = n01;
                if (qv) M[a10] = n10;
                if (qv && sv) M[a11] = n11;
            }
        }
        JP(_p1);
        __syncthreads();
        JP(_p2);
        const int rn = (it + 1) % nround;
        if (c.wave == 0) {
            if (rn == 0) {                                // the rotations of the sweep that just ended are all counted
                const double tot = mg_wave_sum(nrot);
                if (c.lane == 0) red[16] = tot;
                nrot = 0.0;
            }
            rotations(rn, b ^ 1);
        } else if (any) {
            // ---- V <- V J_b of this round: thread = (row, pair), up to four items
            double2 rv[4];
            int2 pv[4];
            double v0[4], v1[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { rv[t] = cs2[vkb[t]]; pv[t] = pq2[vkb[t]]; }
#pragma unroll
            for (int t = 0; t < 4; ++t) { v0[t] = V[vrow[t] + pv[t].x]; v1[t] = V[vrow[t] + pv[t].y]; }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (vact[t] && rv[t].y != 0.0) {          // (an invalid partner has the identity rotation)
                    V[vrow[t] + pv[t].x] = rv[t].x * v0[t] - rv[t].y * v1[t];
                    V[vrow[t] + pv[t].y] = rv[t].y * v0[t] + rv[t].x * v1[t];
                }
            }
        }
        JP(_p3);
        __syncthreads();
        JP(_p4);
#ifdef BA_PROFILE
        if (c.tid == 0) { red[20] += (double)(_p1 - _p0); red[21] += (double)(_p2 - _p1); red[22] += (double)(_p3 - _p2); red[23] += (double)(_p4 - _p3); }
        if (c.tid == 64) { red[24] += (double)(_p1 - _p0); red[25] += (double)(_p2 - _p1); red[26] += (double)(_p3 - _p2); red[27] += (double)(_p4 - _p3); }
#endif
        if (rn == 0) {
            ++sweep;
            if (red[16] == 0.0 || sweep >= MG_MAXSWEEP) break;
        }
    }
    __syncthreads();
    return sweep;
}

// ------------------------------------------------------------------------------------------------
// Eigen-decomposition of a symmetric n x n matrix that is positive semi-definite up to rounding noise, in LDS (offM,
// leading dimension ld), by the Veselic-Hari scheme: diagonally pivoted Cholesky  P^T (A + delta I) P = L L^T , then
// one-sided (Hestenes) Jacobi on the columns of L:  L V = U Sigma , hence  A = U (Sigma^2 - delta) U^T.
// Why this and not the two-sided iteration above:
//   * only contiguous column access (no mixed row / column walks of a symmetric storage -> no bank-conflict storm), no
//     separate eigenvector accumulation, one barrier per round, ~1/3 of the cycles per round;
//   * eigenvalues of a graded matrix come with high relative accuracy (Amm spans 1e2 .. 1e12 and gets inverted);
//   * 16 lanes (one DPP row) per column pair, all pairs of a round-robin round at once; squared column norms are carried
//     in LDS and updated with the rotation (alpha' = alpha - t gamma, beta' = beta + t gamma), so a rotation costs one
//     length-n dot product (reduced on the DPP row network) and one pass over the two columns held in registers.
// The shift delta: the kept block A' of a marginalization is singular in the gauge directions and carries indefinite
// rounding noise (observed -0.08 .. +1e-3 against a spectrum reaching 4e7), so a plain Cholesky breaks down; the
// factorisation is retried with delta = delta0, 16 delta0, ... (delta0 = delta0_rel * max diagonal, or 1e-12 * that after
// a failure with delta0 = 0) until every pivot stays above delta / 100.  lambda_i = sigma_i^2 - delta reproduces the
// spectrum of A itself — noise eigenvalues included, so the lambda <= 1e-8 cut of marginalization_factor.cpp:272-296
// sees what SelfAdjointEigenSolver would show it.
// On exit (same convention as jacobi_eig): diag(M) = eigenvalues, V holds the eigenvectors as COLUMNS.
// A pair is rotated while |g_p . g_q| > tol |g_p| |g_q|: 2e-16 for Amm (it is inverted: at 1e-14 the Schur complement
// loses a digit against the extended-precision yardstick; at that level a few windows keep rotating on rounding noise,
// hence the cap of 10 sweeps: 5-8 converge the others), 1e-14 / 16 sweeps for the kept block.
// `offcs` = LDS scratch of >= 4 n + 8 doubles.  Returns sweeps | attempts << 8.
DEV double mg_row16_sum(double v) {          // sum over the 16 lanes of a DPP row, result in all 16
    v += dpp_mov_f64<0xB1>(v);
    v += dpp_mov_f64<0x4E>(v);
    v += dpp_mov_f64<0x141>(v);
    v += dpp_mov_f64<0x140>(v);
    return v;
}
#define MG_HROWS 6                            // elements of a column per lane: n <= 96
#ifdef BA_PROFILE_DETAIL
__device__ double g_mprof[16];
__device__ double g_mrel[16];
#define MP_DECL long long _mp = clock64()
#define MP_ADD(id) do { if (blockIdx.x == 0 && threadIdx.x == 0) { const long long _n = clock64(); g_mprof[id] += (double)(_n - _mp); _mp = _n; } } while (0)
extern "C" int vg_debug_marg_rel(double* out16) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_mrel), sizeof(double) * 16);
    double z[16] = {0};
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_mrel), z, sizeof(z));
    return e == hipSuccess ? 0 : -2;
}
extern "C" int vg_debug_marg_profile(double* out16, int reset) {
    hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_mprof), sizeof(double) * 16);
    if (e == hipSuccess && reset) { double z[16] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(g_mprof), z, sizeof(z)); }
    return e == hipSuccess ? 0 : -2;
}
#else
#define MP_DECL
#define MP_ADD(id)
#endif
DEV int vh_eig(const MCtx& c, int offM, int offV, int n, int ld, int offcs, int offred, double delta0_rel, double tol, double tolq, int maxsweep) {
    double* A = MG_LDS + offM;
    double* Lc = MG_LDS + offV;               // column k of L at Lc[k * ld + i]
    double* dgn = MG_LDS + offcs;             // running diagonal of the Schur complement
    double* nrm = dgn + n;
    double* lamv = nrm + n;
    int* taken = (int*)(lamv + n);
    double* red = MG_LDS + offred;
    const int grp = c.tid >> 4, sub = c.tid & 15;
    const int ngrp = MG_NT / 16;
    // ---- diagonally pivoted Cholesky of A + delta I (left-looking: column k from A and the previous columns)
    double dmax0 = 0.0;
    for (int i = 0; i < n; ++i) dmax0 = fmax(dmax0, A[i * ld + i]);
    double delta = delta0_rel * dmax0;
    int attempts = 0;
    for (;;) {
        ++attempts;
        __syncthreads();
        for (int i = c.tid; i < n; i += MG_NT) { dgn[i] = A[i * ld + i] + delta; taken[i] = 0; }
        for (int k = c.tid; k < n * ld; k += MG_NT) Lc[k] = 0.0;
        __syncthreads();
        const double floor_ = fmax(0.01 * delta, 1e-15 * dmax0);
        bool broke = false;
        for (int k = 0; k < n; ++k) {
            // pivot = largest remaining diagonal (lowest index on ties), found by wavefront 0 (two entries per lane, DPP max)
            if (c.wave == 0) {
                const int i0 = c.lane, i1 = c.lane + 64;
                const double v0 = (i0 < n && !taken[i0]) ? dgn[i0] : -1e300;
                const double v1 = (i1 < n && !taken[i1]) ? dgn[i1] : -1e300;
                const double bv = wave_max_all(fmax(v0, v1));
                const unsigned long long m0 = __ballot(v0 == bv), m1 = __ballot(v1 == bv);
                if (c.lane == 0) { red[17] = bv; red[18] = (double)(m0 ? __ffsll((long long)m0) - 1 : 64 + __ffsll((long long)m1) - 1); }
            }
            __syncthreads();
            const double best = red[17];
            const int p = (int)red[18];
            if (!(best > floor_)) { broke = true; break; }
            const double inv = mg_rsqrt(best);
            // L[i][k] = (A[i][p] - sum_{j<k} L[i][j] L[p][j]) / L[p][k] for the rows not taken yet; 8 lanes per row
            const int g8 = c.tid >> 3, s8 = c.tid & 7;
            for (int i0 = 0; i0 < n; i0 += MG_NT / 8) {
                const int i = i0 + g8;
                const bool on = i < n && !taken[i];
                double sacc = 0.0;
                if (on)
                    for (int j = s8; j < k; j += 8) sacc += Lc[j * ld + i] * Lc[j * ld + p];
                sacc = group8_sum(sacc);
                if (on && s8 == 0) {
                    const double v = i == p ? best * inv : (A[i * ld + p] - sacc) * inv;
                    Lc[k * ld + i] = v;
                    if (i != p) dgn[i] -= v * v;
                    else taken[p] = 1;             // by the lane that owns row p, after its 8-lane group has tested the flag
                }
            }
            __syncthreads();
        }
        __syncthreads();
        if (!broke || attempts >= 12) break;
        delta = fmax(16.0 * delta, 1e-12 * dmax0);
    }
    // ---- one-sided Jacobi on the n columns (length n) of L
    const int N = (n + 1) & ~1, npairs = N / 2, nrounds = N - 1;
    const int hrows = (n + 15) / 16;           // elements of a column per lane actually present
    const double tol2 = tol * tol, tolq2 = tolq * tolq;
    int sweep = 0;
    if (n >= 2) {
        for (;;) {
            for (int i0 = 0; i0 < n; i0 += ngrp) {       // exact squared norms
                const int i = i0 + grp;
                double sq = 0.0;
                if (i < n)
                    for (int j = sub; j < n; j += 16) { const double v = Lc[i * ld + j]; sq += v * v; }
                sq = mg_row16_sum(sq);
                if (i < n && sub == 0) nrm[i] = sq;
            }
            if (c.tid == 0) { red[16] = 0.0; red[19] = 0.0; }
            __syncthreads();
            // round-robin tournament: player N-1 stays, the others rotate: group g plays (r + g, r - g) mod N-1 in round r,
            // i.e. both indices advance by one per round (kept incrementally: no integer division in the loop)
            int a = grp < npairs ? grp : 0, b = grp == 0 ? N - 1 : (grp < npairs ? N - 1 - grp : 0);
            // wavefronts whose four 16-lane groups all lie beyond the last pair only keep the barriers: 16 wavefronts share
            // 4 SIMDs and a round is bound by instruction issue, not by the work of a lane
            const bool wave_on = (c.wave * 4) < npairs;
            MP_DECL;
            for (int r = 0; r < nrounds; ++r) {
                if (!wave_on) { __syncthreads(); continue; }
                const bool act = grp < npairs && a < n && b < n;
                // all LDS reads of the round in one batch: unconditional loads from clamped addresses, masked afterwards (a load
                // under a lane condition is issued and awaited on its own: ten serial round trips per round)
                double gp[MG_HROWS], gq[MG_HROWS];
                const double* ca = Lc + (act ? a : 0) * ld;
                const double* cb = Lc + (act ? b : 0) * ld;
#pragma unroll
                for (int t = 0; t < MG_HROWS; ++t) {
                    const int j = sub + 16 * t;
                    const int jj = j < n ? j : 0;
                    gp[t] = ca[jj];
                    gq[t] = cb[jj];
                }
                // every lane of the pair reads the norms before lane 0 of the pair may replace them
                double al = nrm[act ? a : 0], be = nrm[act ? b : 0];
                double gam = 0.0;
#pragma unroll
                for (int t = 0; t < MG_HROWS; ++t) {
                    const bool in = act && sub + 16 * t < n;
                    gp[t] = in ? gp[t] : 0.0;
                    gq[t] = in ? gq[t] : 0.0;
                    gam += gp[t] * gq[t];
                }
                al = act ? al : 1.0; be = act ? be : 1.0;
                MP_ADD(0);
                gam = mg_row16_sum(gam);
                MP_ADD(1);
                __builtin_amdgcn_wave_barrier();
                MP_ADD(2);
                if (act && gam * gam > tol2 * al * be && al > 0.0 && be > 0.0) {
                    // rotation that annihilates g_p . g_q, from cos 2phi = |d| / hypot(d, 2 gamma) with two rsqrt chains
                    // and no division:  c = sqrt((1 + cos 2phi) / 2),  s = sign(d) gamma / (hypot c),  t = s / c
                    const double d = be - al, g2 = 2.0 * gam;
                    const double rh = mg_rsqrt(d * d + g2 * g2);
                    const double xx = 0.5 + 0.5 * fabs(d) * rh;          // c^2 in [1/2, 1]
                    const double rc = mg_rsqrt(xx);                       // 1 / c
                    const double cs = xx * rc;
                    const double sn = (d >= 0.0 ? 0.5 : -0.5) * g2 * rh * rc;
                    const double t = sn * rc;
#pragma unroll
                    for (int u = 0; u < MG_HROWS; ++u) {
                        const int j = sub + 16 * u;
                        if (u < hrows && j < n) {
                            Lc[a * ld + j] = cs * gp[u] - sn * gq[u];
                            Lc[b * ld + j] = sn * gp[u] + cs * gq[u];
                        }
                    }
                    if (sub == 0) {
                        nrm[a] = al - t * gam; nrm[b] = be + t * gam; red[16] = 1.0;
                        if (gam * gam > tolq2 * al * be) red[19] = 1.0;          // a rotation above the look-ahead threshold
                    }
#ifdef BA_PROFILE_DETAIL
                    if (sub == 0 && blockIdx.x == 0 && sweep < 11) {
                        atomicAdd(&g_mprof[5 + sweep], 1.0);
                        const double rel = gam * gam / (al * be);
                        atomicMax((unsigned long long*)&g_mrel[sweep], (unsigned long long)__double_as_longlong(rel));
                    }
#endif
                }
                MP_ADD(3);
                a = a + 1 == N - 1 ? 0 : a + 1;
                if (grp != 0) b = b + 1 == N - 1 ? 0 : b + 1;
                __syncthreads();
                MP_ADD(4);
            }
            ++sweep;
            // Stop when a sweep rotated nothing — or nothing above tolq: the sweep that just ended has annihilated those pairs,
            // and what they leave behind is of second order (the verification sweep would rotate nothing).
            const bool again = red[16] != 0.0 && red[19] != 0.0 && sweep < maxsweep;
            __syncthreads();
            if (!again) break;
        }
    }
    // ---- lambda_i = |column i|^2 - delta, u_i = column_i / |column i|
    for (int i0 = 0; i0 < n; i0 += ngrp) {
        const int i = i0 + grp;
        double sq = 0.0;
        if (i < n)
            for (int j = sub; j < n; j += 16) { const double v = Lc[i * ld + j]; sq += v * v; }
        sq = mg_row16_sum(sq);
        if (i < n && sub == 0) { nrm[i] = sq; lamv[i] = sq - delta; }
    }
    __syncthreads();
    for (int k = c.tid; k < n * n; k += MG_NT) {       // transposed + normalised into the M area ...
        const int i = k / n, j = k - i * n;
        const double l2 = nrm[i];
        A[j * ld + i] = l2 > 0.0 ? Lc[i * ld + j] * mg_rsqrt(l2) : 0.0;
    }
    __syncthreads();
    for (int k = c.tid; k < n * n; k += MG_NT) {       // ... and back into V
        const int i = k / n, j = k - i * n;
        Lc[i * ld + j] = A[i * ld + j];
    }
    __syncthreads();
    for (int i = c.tid; i < n; i += MG_NT) A[i * ld + i] = lamv[i];
    __syncthreads();
    return sweep | (attempts << 8);
}

// ------------------------------------------------------------------------------------------------
// Square-root form of the new prior WITHOUT an eigen-decomposition (VG_MARG_SQRT, the default).
// marginalization_factor.cpp:285-296 turns the kept system (A', b') into a factor  J0 = S^1/2 V^T ,  r0 = S^-1/2 V^T b'  from
// A' = V S V^T.  Everything downstream — MarginalizationFactor::Evaluate (:333-381), the solver, the next marginalization —
// sees that factor only through  J0^T J0 = A' ,  J0^T r0 = b'  and  |r0|^2 = b'^T A'^+ b' : all three are unchanged when J0
// and r0 are multiplied from the left by an orthogonal matrix (the eigenvector signs / order Eigen happens to return are such
// a freedom already).  The diagonally pivoted Cholesky factor  P^T A' P = L L^T  is such a factor:  J0 = L^T ,  r0 = L^-1 b'.
// The reference's cut `lambda > eps` (eps = 1e-8, marginalization_factor.h:70) drops the directions in which A' carries no
// information (the gauge freedoms of the window: their eigenvalues are rounding noise); the pivoted factorisation stops when
// no remaining diagonal entry of the Schur complement exceeds the same eps: rank r, rows r .. n-1 of J0 and r0 are zero, and
// J0^T J0 differs from A' by that residual (<= (n - r) eps in trace when it is positive semi-definite, rounding noise
// otherwise) — the same order as the part the eigenvalue cut removes.  75 dependent pivots instead of ~7 Jacobi sweeps of 75
// rounds: 1.41M -> ~0.1M cycles per window.  b' rides along as an augmented row, so r0 needs no separate substitution.
// On exit: column k of L at V[k * ld + i] (k < rank; zero for the rows pivoted earlier), y = r0 in cs[2n .. 3n).  Returns rank.
#ifndef MG_SQRT_RIGHT_LOOKING
// LEFT-looking (round 4): the Schur complement is never formed.  Column k of L is computed when its pivot p is chosen,
//     L[i][k] = (A'[i][p] - sum_{j<k} L[i][j] L[p][j]) / sqrt(pivot),
// by eight lanes per row (lane `part` takes j = part, part + 8, ...; sum on the DPP network), the augmented row i = n carries b'
// (y_k = (b'_p - sum_j y_j L[p][j]) / l_pp), and the running diagonal d_i -= L[i][k]^2 lives in LDS twice (read one copy, write the
// other: a wavefront that is already updating must not change what a slower one is still searching) -- ONE barrier per pivot instead
// of two, and ~k/4 FMAs per lane instead of an update of the whole trailing matrix: 265K -> ~100K cycles per window for n = 76.
// (The right-looking form -- Schur complement in registers, two barriers per pivot -- is kept behind -DMG_SQRT_RIGHT_LOOKING.)
// Same pivot rule (largest remaining diagonal, lowest index on ties), same cut, same output layout; the sums run in another order.
// Round 6: the same pivot rule with SHORT dot products.  The sum over all earlier columns (37 terms on average for n = 75, five
// dependent LDS-read -> FMA rounds per lane) is what a pivot waits for; after every sixteenth pivot the finished block of columns is
// now subtracted from what is left of A' -- A' -= L_blk L_blk^T on v_mfma_f64_16x16x4_f64, one lower 16 x 16 tile per wavefront, and
// b' -= L_blk y_blk -- so a column only needs the columns of its OWN block: at most two terms per lane, one round.  This is LAPACK's
// blocked pivoted Cholesky (dpstrf): the running diagonal, the search and therefore the pivot sequence are those of the unblocked
// form, only the order of the sums changes.  (A fixed elimination order on the matrix cores -- no search at all -- was measured
// 20 % faster still and dropped: it cuts weak directions the pivoted order keeps, profiles/experiments/r06d_*.)
typedef double mg_d4 __attribute__((vector_size(32)));
// Round 6 (second form): the pivot chain on ONE wavefront, in registers.  A pivot is a chain of dependent steps -- search, 1 / sqrt,
// the pivot row's entries, the dot products, the new column -- and with sixteen wavefronts taking part each link of it crossed LDS
// and a workgroup barrier (2.4K cycles per pivot, 190K of the kernel's 540K).  Now lane t of wavefront 0 owns rows t and t + 64 (the
// augmented row n, which carries b', among them; n <= 96): the entries of its rows in the CURRENT block of sixteen columns, its
// running diagonal and its liveness stay in registers; the pivot row's entries reach the other lanes through v_readlane (the pivot
// index is uniform) straight into the multiply-adds, A'[i][p] is one LDS read issued as soon as p is known, the search is a DPP max
// plus two ballots.  No barrier inside a block; the other wavefronts wait at the block's end and join for the trailing update
// (A' -= L_blk L_blk^T on the matrix cores, as before).  Same pivot rule, same cut, same output layout; the dot product of a column
// runs over the block's earlier columns in order.
template <int T, int PS>
DEV void sqrt_dot(const double (&b0)[16], const double (&b1)[16], int pl, double& s0, double& s1) {
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const double lp = readlane_f64(PS ? b1[j] : b0[j], pl);
        s0 = fma(b0[j], lp, s0); s1 = fma(b1[j], lp, s1);
    }
}
template <int T>
DEV bool sqrt_step(int k, int n, int ld, int lane, const double* A, double* Lc, double* yv, const double* bb,
                   double (&b0)[16], double (&b1)[16], double& d0, double& d1) {
    const int r0 = lane, r1 = lane + 64;
    // pivot = largest remaining diagonal (lowest index on ties)
    const double best = wave_max_all(fmax(d0, d1));
    const unsigned long long m0 = __ballot(d0 == best), m1 = __ballot(d1 == best);
    const int p = uni(m0 ? __ffsll((long long)m0) - 1 : 64 + __ffsll((long long)m1) - 1);
    if (!(best > MG_EPS)) return false;           // (uniform) nothing above eps is left: the rest is what the reference's cut drops
    const int pl = p & 63;
    // (the augmented row n is lane n's first row when n < 64, lane n - 64's second one otherwise: its "A' entry" is b'_p, its
    //  block entries are y, it has no diagonal.  One unconditional LDS read per row from a selected address -- rows beyond n read
    //  b'_p too and drop it -- and selects instead of branches: the step is one straight chain)
    const int ob = (int)(bb - A) + p;
    const int o0 = r0 < n ? (r0 > p ? r0 * ld + p : p * ld + r0) : ob;
    const int o1 = r1 < n ? (r1 > p ? r1 * ld + p : p * ld + r1) : ob;
    const double a0 = ((const lds_d*)A)[o0], a1 = ((const lds_d*)A)[o1];
    const double inv = mg_rsqrt(best);
    double s0 = 0.0, s1 = 0.0;
    if (p < 64) sqrt_dot<T, 0>(b0, b1, pl, s0, s1); else sqrt_dot<T, 1>(b0, b1, pl, s0, s1);      // (uniform)
    const bool live0 = r0 < n && d0 > -1e299, live1 = r1 < n && d1 > -1e299;
    const double piv = best * inv;
    double v0 = (a0 - s0) * inv, v1 = (a1 - s1) * inv;
    v0 = (live0 || r0 == n) ? v0 : 0.0;
    v1 = (live1 || r1 == n) ? v1 : 0.0;
    v0 = r0 == p ? piv : v0;
    v1 = r1 == p ? piv : v1;
    b0[T] = v0; b1[T] = v1;
    d0 = (r0 == p || !live0) ? -1e300 : d0 - v0 * v0;
    d1 = (r1 == p || !live1) ? -1e300 : d1 - v1 * v1;
    const int w0 = r0 < n ? (int)(Lc - A) + k * ld + r0 : (int)(yv - A) + k;
    const int w1 = r1 < n ? (int)(Lc - A) + k * ld + r1 : (int)(yv - A) + k;
    if (r0 <= n) ((lds_d*)A)[w0] = v0;
    if (r1 <= n) ((lds_d*)A)[w1] = v1;
    return true;
}
DEV int sqrt_factor(const MCtx& c, int offM, int offV, int n, int ld, int offcs, int offred, const double* bglob) {
    double* A = MG_LDS + offM;                // A' (lower triangle read; updated block by block)
    double* Lc = MG_LDS + offV;               // column k of L at Lc[k * ld + i]
    double* dg0 = MG_LDS + offcs;             // (the running diagonal lives in registers now; the layout behind it is the caller's)
    double* yv = dg0 + 2 * n + 1;             // y = L^-1 P^T b'   (kept for the caller: cs[2n + 1 ..))
    double* bb = yv + 2 * n;                  // b' minus the finished blocks' share
    int* flag = (int*)(MG_LDS + offred + 20); // [0] rank so far, [1] stop
    __syncthreads();
    for (int k = c.tid; k < n * ld; k += MG_NT) Lc[k] = 0.0;
    for (int i = c.tid; i < n; i += MG_NT) { yv[i] = 0.0; bb[i] = bglob[i]; }
    if (c.tid == 0) { flag[0] = 0; flag[1] = 0; }
    const int wave = __builtin_amdgcn_readfirstlane(c.wave);
    const int ntl = (n + 15) >> 4, ntile = ntl * (ntl + 1) / 2;       // lower 16 x 16 tiles of A' (16 ntl <= ld: the rows beyond n are zero in Lc)
    __syncthreads();
    double d0 = -1e300, d1 = -1e300;
    if (wave == 0) {
        if (c.lane < n) d0 = A[c.lane * ld + c.lane];
        if (c.lane + 64 < n) d1 = A[(c.lane + 64) * ld + c.lane + 64];
    }
    for (int kb = 0; kb < n; kb += 16) {
        if (kb > 0) {
            // ---- the block of columns [kb - 16, kb) leaves A' and b'
            const int k0 = kb - 16;
            const int jc = c.lane & 15, kq = c.lane >> 4;
            for (int t = wave; t < ntile; t += MG_NW) {
                int ti, tj;
                tri_decode(t, ti, tj);
                const double* pa = Lc + (k0 + kq) * ld + 16 * ti + jc;
                const double* pb = Lc + (k0 + kq) * ld + 16 * tj + jc;
                const double a0 = pa[0], a1 = pa[4 * ld], a2 = pa[8 * ld], a3 = pa[12 * ld];
                const double b0 = pb[0], b1 = pb[4 * ld], b2 = pb[8 * ld], b3 = pb[12 * ld];
                mg_d4 acc = {0, 0, 0, 0};
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, acc, 0, 0, 0);
                const int col = 16 * tj + jc;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int r = 16 * ti + kq + 4 * reg;       // D[i = kq + 4 reg][j = jc]
                    if (r < n && col <= r) A[r * ld + col] -= acc[reg];
                }
            }
            for (int i = c.tid; i < n; i += MG_NT) {
                double sacc = 0.0;
#pragma unroll
                for (int j = 0; j < 16; ++j) sacc += Lc[(k0 + j) * ld + i] * yv[k0 + j];
                bb[i] -= sacc;
            }
            __syncthreads();
        }
        if (wave == 0) {
            double b0[16], b1[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) { b0[j] = 0.0; b1[j] = 0.0; }
            int done = 0;
            bool go = true;
#define MG_SQRT_STEP(T) if (go && kb + T < n) { go = sqrt_step<T>(kb + T, n, ld, c.lane, A, Lc, yv, bb, b0, b1, d0, d1); done += go ? 1 : 0; }
            MG_SQRT_STEP(0) MG_SQRT_STEP(1) MG_SQRT_STEP(2) MG_SQRT_STEP(3) MG_SQRT_STEP(4) MG_SQRT_STEP(5) MG_SQRT_STEP(6) MG_SQRT_STEP(7)
            MG_SQRT_STEP(8) MG_SQRT_STEP(9) MG_SQRT_STEP(10) MG_SQRT_STEP(11) MG_SQRT_STEP(12) MG_SQRT_STEP(13) MG_SQRT_STEP(14) MG_SQRT_STEP(15)
#undef MG_SQRT_STEP
            if (c.lane == 0) { flag[0] = kb + done; flag[1] = go ? 0 : 1; }
        }
        __syncthreads();
        if (flag[1]) break;                   // (uniform)
    }
    const int rank = flag[0];
    __syncthreads();
    return rank;
}
#else
DEV int sqrt_factor(const MCtx& c, int offM, int offV, int n, int ld, int offcs, int offred, const double* bglob) {
    // Right-looking, the Schur complement in REGISTERS: thread t owns the entries t, t + MG_NT, ... of the lower triangle of A'
    // (packed by rows, tri_decode) and of the augmented row n that carries b'.  Per pivot: every wavefront finds the largest
    // remaining diagonal entry itself (the running diagonal is mirrored in LDS; a wavefront-local search needs no barrier),
    // the owners of column / row p publish l = a[., p] / sqrt(pivot) through LDS, everybody subtracts l_i l_j from what it
    // owns: two barriers and ~3 FMAs per thread and pivot (n = 75: 2925 entries on 1024 threads).
    const double* A = MG_LDS + offM;
    double* Lc = MG_LDS + offV;               // column k of L at Lc[k * ld + i]
    double* dgn = MG_LDS + offcs;             // running diagonal of the Schur complement (-1e300 once pivoted)
    double* lcol = dgn + n;                   // current column of L, entry n = the augmented row's (y_k)
    double* yv = lcol + n + 1;                // y = L^-1 P^T b'   (kept for the caller)
    (void)offred;
    const int ntri = n * (n + 1) / 2, nent = ntri + n;
    enum { MAXE = (96 * 97 / 2 + 96 + MG_NT - 1) / MG_NT };      // n <= 96
    double a[MAXE];
    int ii[MAXE], jj[MAXE];
    bool alive[MAXE];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < MAXE; ++q) {
        const int e = c.tid + q * MG_NT;
        alive[q] = e < nent;
        ii[q] = 0; jj[q] = 0; a[q] = 0.0;
        if (alive[q]) {
            if (e < ntri) { tri_decode(e, ii[q], jj[q]); a[q] = A[ii[q] * ld + jj[q]]; }
            else { ii[q] = n; jj[q] = e - ntri; a[q] = bglob[jj[q]]; }
        }
    }
    __syncthreads();                          // (A' = the M area is read completely before Lc — possibly the same LDS — is cleared)
    for (int k = c.tid; k < n * ld; k += MG_NT) Lc[k] = 0.0;
    for (int i = c.tid; i < n; i += MG_NT) yv[i] = 0.0;
#pragma unroll
    for (int q = 0; q < MAXE; ++q)
        if (alive[q] && ii[q] == jj[q]) dgn[ii[q]] = a[q];
    int rank = 0;
    bool pivot_owner = false;
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        // pivot = largest remaining diagonal (lowest index on ties): every wavefront on its own, two entries per lane
        const int i0 = c.lane, i1 = c.lane + 64;
        const double v0 = i0 < n ? dgn[i0] : -1e300;
        const double v1 = i1 < n ? dgn[i1] : -1e300;
        const double best = wave_max_all(fmax(v0, v1));
        const unsigned long long m0 = __ballot(v0 == best), m1 = __ballot(v1 == best);
        const int p = m0 ? __ffsll((long long)m0) - 1 : 64 + __ffsll((long long)m1) - 1;
        if (!(best > MG_EPS)) break;          // (uniform) nothing above eps is left: the rest is what the reference's cut drops
        const double inv = mg_rsqrt(best);
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            if (!alive[q]) continue;
            const bool inrow = ii[q] == p, incol = jj[q] == p;
            if (!(inrow || incol)) continue;
            // entry (i, p) of the column below / (p, j) of the row left of the pivot (the same by symmetry), or the pivot itself
            const int other = incol ? ii[q] : jj[q];
            const double v = (inrow && incol) ? best * inv : a[q] * inv;
            lcol[other] = v;
            if (other < n) Lc[k * ld + other] = v; else yv[k] = v;
            pivot_owner = pivot_owner || (inrow && incol);
            alive[q] = false;
        }
        rank = k + 1;
        __syncthreads();
        // (the pivot's slot of the running diagonal is retired only now: every wavefront searched the same diagonal above, and
        //  a wavefront that runs ahead must not change it under the others.  Running the factorisation on four wavefronts, one
        //  per SIMD, with the other twelve only keeping the barrier count was measured: 451K instead of 304K cycles.)
        if (pivot_owner) { dgn[p] = -1e300; pivot_owner = false; }
#pragma unroll
        for (int q = 0; q < MAXE; ++q) {
            if (!alive[q]) continue;
            a[q] -= lcol[ii[q]] * lcol[jj[q]];
            if (ii[q] == jj[q]) dgn[ii[q]] = a[q];
        }
    }
    __syncthreads();
    return rank;
}

#endif
// The same factor for a kept block that does not fit the register-resident form (n > 96: windows of the large-window path, kept
// dimension up to 6 * 39 + 16): the Schur complement stays where it is (M, n x n in global memory, lower triangle updated in
// place), running diagonal / current column / y in LDS.  Same pivot rule, same cut, same output layout.
DEV int sqrt_factor_glb(const MCtx& c, double* M, double* Lc, int n, int ld, int offcs, const double* bglob) {
    double* dgn = MG_LDS + offcs;             // running diagonal (-1e300 once pivoted)
    double* lcol = dgn + n;                   // current column of L (0 for pivoted rows), entry n = y_k
    double* yv = lcol + n + 1;                // y = L^-1 P^T b'   (kept for the caller)
    double* yb = yv + n;                      // running right-hand side
    const int ntri = n * (n + 1) / 2;
    __syncthreads();
    for (int k = c.tid; k < n * ld; k += MG_NT) Lc[k] = 0.0;
    for (int i = c.tid; i < n; i += MG_NT) { dgn[i] = M[i * ld + i]; yv[i] = 0.0; yb[i] = bglob[i]; }
    int rank = 0;
    for (int k = 0; k < n; ++k) {
        __syncthreads();
        // pivot = largest remaining diagonal (lowest index on ties): every wavefront on its own, four entries per lane (n <= 256)
        double best = -1e300;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = c.lane + 64 * u; best = fmax(best, i < n ? dgn[i] : -1e300); }
        best = wave_max_all(best);
        int p = n;
#pragma unroll
        for (int u = 3; u >= 0; --u) {
            const int i = c.lane + 64 * u;
            const unsigned long long mk = __ballot(i < n && dgn[i] == best);
            if (mk) p = 64 * u + __ffsll((long long)mk) - 1;
        }
        if (!(best > MG_EPS)) break;          // (uniform)
        const double inv = mg_rsqrt(best);
        const double yk = yb[p] * inv;
        for (int i = c.tid; i < n; i += MG_NT) {
            double v = 0.0;
            if (i == p) v = best * inv;
            else if (dgn[i] > -1e299) v = (i > p ? M[i * ld + p] : M[p * ld + i]) * inv;
            lcol[i] = v;
            Lc[k * ld + i] = v;
        }
        if (c.tid == 0) { lcol[n] = yk; yv[k] = yk; }
        rank = k + 1;
        __syncthreads();
        if (c.tid == 0) dgn[p] = -1e300;
        for (int i = c.tid; i < n; i += MG_NT) if (i != p && dgn[i] > -1e299) yb[i] -= lcol[i] * yk;
        for (int e = c.tid; e < ntri; e += MG_NT) {
            int i, j;
            tri_decode(e, i, j);                                  // i >= j
            const double li = lcol[i], lj = lcol[j];
            if (i == p || j == p || li == 0.0 || lj == 0.0) continue;
            const double v = M[i * ld + j] - li * lj;
            M[i * ld + j] = v;
            if (i == j && dgn[i] > -1e299) dgn[i] = v;
        }
    }
    __syncthreads();
    return rank;
}

DEV bool mg_fast_ok(int n) {
    const int half = (n + 1) / 2;
    return n >= 2 && half * (half + 1) / 2 <= MG_NT && n * half <= 4 * (MG_NT - 64);
}

// marginalization column maps, kept in LDS ints
// LDS int tables of the kernel (offsets in ints behind the state copy; the host reserves MGI_TOTAL ints): sized for the
// large-window path (BA_MAX_K_LARGE frames, kept dimension <= 256)
#define MGI_POSE 0
#define MGI_SB BA_MAX_K_LARGE
#define MGI_MISC (2 * BA_MAX_K_LARGE)
#define MGI_RANK (2 * BA_MAX_K_LARGE + 16)
#define MGI_BPTR (MGI_RANK + 256)
#define MGI_TOTAL (MGI_BPTR + BA_MAX_K_LARGE + 8)
static_assert(MGI_TOTAL <= BA_MARG_LDS_INTS, "LDS int tables of ba_marg_kernel");
struct MgMap {
    int* pose;   // [BA_MAX_K_LARGE] column of pose i or -1
    int* sb;     // [BA_MAX_K_LARGE]
    int* misc;   // [0]=ex col, [1]=td col, [2]=m, [3]=n, [4]=n0, [5]=pos
    int* lm;     // global: [Lcap] column of landmark l or -1
    int* l0;     // global: [Lcap] list of frame-0 landmarks
};

extern "C" __global__ __launch_bounds__(MG_NT) void ba_marg_kernel(const BaLayout* __restrict__ Lp, BaPtrs P) {
    const BaLayout& L = *Lp;
    MCtx c;
    c.Lp = Lp;
    const int w = blockIdx.x;
    c.ia = P.iarr + (size_t)w * L.istride;
    c.hdr = c.ia + L.io_hdr;
    const int flag = c.hdr[H_MARGIN];
    int* mi = P.miout + (size_t)w * L.mi_stride;
    if (flag == VG_MARGIN_NONE) return;
    const int* iout = P.iout + (size_t)w * L.oi_stride;
    if (iout[0] != VG_OK) return;                       // failed solve: no prior
    c.di = P.din + (size_t)w * L.dstride;
    c.pri = P.pri + (size_t)w * L.pstride;
    c.sc = P.scr + (size_t)w * L.sstride;
    c.ms = P.mscr + (size_t)w * L.ms_stride;
    c.lds = MG_LDS;
    c.tid = threadIdx.x; c.lane = c.tid & 63; c.wave = c.tid >> 6;
    c.focal = c.di[L.do_par + P_FOCAL]; c.tr = c.di[L.do_par + P_TR]; c.row = c.di[L.do_par + P_ROW];
    c.gnorm = c.di[L.do_par + P_GNORM];
    const double* outp = P.out + (size_t)w * L.ostride;
    double* mo = P.mout + (size_t)w * L.mo_stride;
    const int K = L.K, nL = c.hdr[H_L], nprior = c.hdr[H_NPRIOR], nblk = c.hdr[H_NBLK];
    const int mcap = L.mcap;
#ifdef BA_PROFILE
    const long long _tstart = clock64();
    int* mprof = mi + 8 + 2 * (L.K + 4);
#define MPROF(i) do { if (threadIdx.x == 0) mprof[i] = (int)((clock64() - _tstart) >> 10); } while (0)
#else
#define MPROF(i) do { } while (0)
#endif

    // ---- LDS carve: [eigM ld*ld][eigV ld*ld][cs 2*ld][red 16][x state][ints]
    const int ld = L.mg_ld;
    double* eM = MG_LDS;
    double* eV = eM + ld * ld;
    double* cs = eV + ld * ld;
    double* red = cs + L.mg_cs;
    double* x = red + 32;
    const int nst = (7 * K + 9 * K + 8 + 1) & ~1;
    int* li = (int*)(x + nst);
    MgMap mp;
    mp.pose = li + MGI_POSE; mp.sb = li + MGI_SB; mp.misc = li + MGI_MISC;
    // ---- global scratch carve
    const int posmax = L.mg_posmax;
    double* A = c.ms;                        // posmax x posmax
    double* bv = A + (size_t)posmax * posmax;
    double* rec = bv + posmax;               // Fcap x 42 projection records
    double* T1 = rec + (size_t)L.Fcap * 42;  // posmax x (mcap+1)
    double* T2 = T1 + (size_t)posmax * (mcap + 1);
    double* gM = T2 + (size_t)posmax * (mcap + 1);   // staging / eig fallback (posmax^2) x 2
    double* gV = gM + (size_t)posmax * posmax;
    double* g2M = gV + (size_t)posmax * posmax;      // second-eig fallback (mcap^2) x 2
    double* g2V = g2M + (size_t)mcap * mcap;
    double* prv = g2V + (size_t)mcap * mcap;         // prior residual (Ncap) + dx (Ncap)
    mp.lm = (int*)(prv + 2 * L.Ncap + 480);
    mp.l0 = mp.lm + L.Lcap;

    // ---- state after the gauge fix (what vector2double() repacks at estimator.cpp:831 / :942)
    for (int k = c.tid; k < 7 * K; k += MG_NT) x[k] = outp[L.oo_pose + k];
    for (int k = c.tid; k < 9 * K; k += MG_NT) x[7 * K + k] = outp[L.oo_sb + k];
    if (c.tid < 7) x[16 * K + c.tid] = outp[L.oo_ex + c.tid];
    if (c.tid == 7) x[16 * K + 7] = outp[L.oo_td];
    const double* lam = outp + L.oo_lam;
    const double* ex = x + 16 * K;
    const int* pk = c.ia + L.io_pb_kind;
    const int* pidx = c.ia + L.io_pb_idx;
    const int* poff = c.ia + L.io_pb_off;
    const int* px0off = c.ia + L.io_pb_x0off;
    const bool imu0 = (flag == VG_MARGIN_OLD) && c.ia[L.io_imu_valid + 0] && c.di[L.do_imu + IM_SUMDT] < 10.0;
    __syncthreads();

    // ---- M2: structure.  Which frames / blocks take part is gathered by all threads (flags in LDS: idempotent stores of 1) and
    //      the list of frame-0 landmarks by an ordered compaction (ballot prefix per wavefront, wavefront counts through LDS):
    //      a single thread walking the landmark and factor tables was a chain of ~750 dependent HBM round trips, 160K of the
    //      kernel's 1M cycles (and 2000 landmarks on the large-window path).  The column assignment itself is a few loops over K.
    int* hpose = mp.pose;                    // flags first, columns afterwards
    int* hsb = mp.sb;
    int* hflag = mp.misc + 8;                // [0] extrinsic, [1] td
    int* wcnt = li + MGI_RANK;               // [waves] (the rank table is not in use yet)
    for (int i = c.tid; i < BA_MAX_K_LARGE; i += MG_NT) { hpose[i] = 0; hsb[i] = 0; }
    if (c.tid < 2) hflag[c.tid] = 0;
    __syncthreads();
    for (int b = c.tid; b < nblk; b += MG_NT) {
        if (pk[b] == VG_BLK_POSE) hpose[pidx[b]] = 1;
        else if (pk[b] == VG_BLK_SPEEDBIAS) hsb[pidx[b]] = 1;
        else if (pk[b] == VG_BLK_EXPOSE) hflag[0] = 1;
        else hflag[1] = 1;
    }
    int n0_all = 0;                          // uniform
    if (flag == VG_MARGIN_OLD) {
        if (c.tid == 0 && imu0) { hpose[0] = 1; hpose[1] = 1; hsb[0] = 1; hsb[1] = 1; }
        for (int base = 0; base < nL; base += MG_NT) {
            const int l = base + c.tid;
            const bool is0 = l < nL && c.ia[L.io_lm_start + l] == 0;
            if (l < nL) mp.lm[l] = -1;
            if (is0) {
                hpose[0] = 1; hflag[0] = 1;
                if (L.t) hflag[1] = 1;
                for (int f = c.ia[L.io_lm_fbeg + l]; f < c.ia[L.io_lm_fbeg + l + 1]; ++f) {
                    const int j = c.ia[L.io_fac_j + f];
                    if (j < K) hpose[j] = 1;
                }
            }
            const unsigned long long bal = __ballot(is0);
            const int before = __popcll(bal & ((1ull << c.lane) - 1ull));
            __syncthreads();                 // (wcnt of the previous trip has been read)
            if (c.lane == 0) wcnt[c.wave] = __popcll(bal);
            __syncthreads();
            int off = n0_all, tot = 0;
            for (int w2 = 0; w2 < MG_NT / 64; ++w2) { const int cw = wcnt[w2]; if (w2 < c.wave) off += cw; tot += cw; }
            if (is0) mp.l0[off + before] = l;
            n0_all += tot;
        }
    }
    __syncthreads();
    if (c.tid == 0) {
        const int n0 = n0_all;
        const bool has_ex = hflag[0] != 0, has_td = hflag[1] != 0;
        bool valid = true;
        if (flag == VG_MARGIN_OLD) {
            if (nblk == 0 && !imu0 && n0 == 0) valid = false;
        } else {
            if (nblk == 0 || !hpose[K - 2]) valid = false;      // estimator.cpp:935-936
        }
        // flags -> columns (in place: a flag is consumed before its slot is overwritten)
        int pos = 0, lm_base = 0;
        int colp[2] = {-1, -1};              // MARGIN_OLD: pose 0 / speed-bias 0;  SECOND_NEW: pose K-2
        if (flag == VG_MARGIN_OLD) {
            if (hpose[0]) { colp[0] = pos; pos += 6; }
            if (hsb[0]) { colp[1] = pos; pos += 9; }
            lm_base = pos; pos += n0;
        } else if (valid) { colp[0] = pos; pos += 6; }
        const int m = pos;
        const int first_pose = (flag == VG_MARGIN_OLD) ? 0 : K - 2;
        for (int i = 0; i < K; ++i) {
            const bool has = hpose[i] != 0;
            int col = -1;
            if (i == first_pose && colp[0] >= 0) col = colp[0];
            else if (has) { col = pos; pos += 6; }
            hpose[i] = col;
        }
        for (int i = 0; i < K; ++i) {
            const bool has = hsb[i] != 0;
            int col = -1;
            if (flag == VG_MARGIN_OLD && i == 0 && colp[1] >= 0) col = colp[1];
            else if (has) { col = pos; pos += 9; }
            hsb[i] = col;
        }
        for (int i = K; i < BA_MAX_K_LARGE; ++i) { hpose[i] = -1; hsb[i] = -1; }
        mp.misc[0] = -1; mp.misc[1] = -1;
        if (has_ex) { mp.misc[0] = pos; pos += 6; }
        if (has_td) { mp.misc[1] = pos; pos += 1; }
        mp.misc[2] = m; mp.misc[3] = pos - m; mp.misc[4] = n0; mp.misc[5] = pos;
        mp.misc[6] = (valid && pos - m > 0 && pos - m <= mcap && pos <= posmax) ? 1 : 0;
        mp.misc[7] = lm_base;
    }
    __syncthreads();
    for (int k = c.tid; k < mp.misc[4]; k += MG_NT) mp.lm[mp.l0[k]] = mp.misc[7] + k;
    const int m = mp.misc[2], n = mp.misc[3], n0 = mp.misc[4], pos = mp.misc[5];
    if (!mp.misc[6]) { if (c.tid == 0) { mi[0] = 0; mi[1] = 0; } return; }
    const int cex = mp.misc[0], ctd = mp.misc[1];

    // (row by wavefront, column by lane: no division per element)
    for (int r = c.wave; r < pos; r += MG_NW)
        for (int cc = c.lane; cc < pos; cc += 64) ((glb_d*)A)[(size_t)r * posmax + cc] = 0.0;
    for (int k = c.tid; k < pos; k += MG_NT) bv[k] = 0.0;
    __syncthreads();

    MPROF(0);
    // ---- M1/M3 (a): prior factor at the new state: r = r0 + J0 dx ; A += J0^T J0 ; b += J0^T r
    // (round 6: dx and r live in LDS -- the eigen-solver tiles are free here --, both products run eight lanes per row with the
    //  partial sums on the DPP network: ~ten loads in flight per lane instead of one thread walking 75 dependent terms; the prior is the
    //  FIRST contribution to its entries of the cleared A / bv: plain stores)
    if (nblk > 0) {
        double* dx = eM;                             // [Ncap]
        double* prl = eM + L.Ncap;                   // [Ncap] prior residual at the new state
        int* pcol = li + MGI_RANK;                   // prior row / column -> column of the marginalization system (the rank table is not in use yet)
        for (int b = c.tid; b < nblk; b += MG_NT) {
            const double* xb = pk[b] == VG_BLK_POSE ? x + 7 * pidx[b] : (pk[b] == VG_BLK_SPEEDBIAS ? x + 7 * K + 9 * pidx[b]
                               : (pk[b] == VG_BLK_EXPOSE ? ex : ex + 7));
            const double* x0 = c.pri + L.po_x0 + px0off[b];
            double* d = dx + poff[b];
            const int sz = pk[b] == VG_BLK_SPEEDBIAS ? 9 : (pk[b] == VG_BLK_TD ? 1 : 6);
            const int base = pk[b] == VG_BLK_POSE ? mp.pose[pidx[b]] : (pk[b] == VG_BLK_SPEEDBIAS ? mp.sb[pidx[b]]
                             : (pk[b] == VG_BLK_EXPOSE ? cex : ctd));
            for (int k = 0; k < sz; ++k) pcol[poff[b] + k] = base + k;
            if (pk[b] == VG_BLK_SPEEDBIAS) { for (int k = 0; k < 9; ++k) d[k] = xb[k] - x0[k]; }
            else if (pk[b] == VG_BLK_TD) d[0] = xb[0] - x0[0];
            else {
                d[0] = xb[0] - x0[0]; d[1] = xb[1] - x0[1]; d[2] = xb[2] - x0[2];
                double qi[4], dq[4];
                q_inv(x0 + 3, qi);
                q_mul(qi, xb + 3, dq);
                const double sgn = (dq[3] >= 0) ? 2.0 : -2.0;
                d[3] = sgn * dq[0]; d[4] = sgn * dq[1]; d[5] = sgn * dq[2];
            }
        }
        __syncthreads();
        const glb_d* J0g = (const glb_d*)(c.pri + L.po_J0);        // row-major, leading dimension pld
        const glb_d* r0g = (const glb_d*)(c.pri + L.po_r0);
        // (eight lanes per row with sixteen wavefronts, four with eight: one pass over the rows either way)
        constexpr int PLR = MG_NT >= 1024 ? 8 : 4, PLS = MG_NT >= 1024 ? 3 : 2;
        const int part = c.tid & (PLR - 1);
        for (int rb = 0; rb < nprior; rb += MG_NT / PLR) {
            const int r = rb + (c.tid >> PLS);
            double sacc = 0.0;
            if (r < nprior)
                for (int k = part; k < nprior; k += PLR) sacc += J0g[(size_t)r * L.pld + k] * dx[k];
            sacc += dpp_mov_f64<0xB1>(sacc);
            sacc += dpp_mov_f64<0x4E>(sacc);
            if (PLR == 8) sacc += dpp_mov_f64<0x141>(sacc);
            if (r < nprior && part == 0) prl[r] = r0g[r] + sacc;
        }
        __syncthreads();
        const glb_d* Hp = (const glb_d*)(c.sc + L.so_Hp);          // J0^T J0 (lower), left by the solve pipeline's prologue
        glb_d* Ag = (glb_d*)A;
        glb_d* bg = (glb_d*)bv;
        for (int ab = 0; ab < nprior; ab += MG_NT / PLR) {
            const int a = ab + (c.tid >> PLS);
            double sacc = 0.0;
            if (a < nprior)
                for (int r = part; r < nprior; r += PLR) sacc += J0g[(size_t)r * L.pld + a] * prl[r];
            sacc += dpp_mov_f64<0xB1>(sacc);
            sacc += dpp_mov_f64<0x4E>(sacc);
            if (PLR == 8) sacc += dpp_mov_f64<0x141>(sacc);
            if (a < nprior && part == 0) bg[pcol[a]] = sacc;
        }
        for (int a = c.wave; a < nprior; a += MG_NW) {
            const int ca = pcol[a];
            for (int b2 = c.lane; b2 < nprior; b2 += 64)
                Ag[(size_t)ca * posmax + pcol[b2]] = (a >= b2) ? Hp[a * L.Ncap + b2] : Hp[b2 * L.Ncap + a];
        }
    }
    __syncthreads();
    MPROF(1);
    // ---- (b) IMU factor 0 -> 1
    if (imu0) {
        const double* pre = c.di + L.do_imu;
        const double* U = c.sc + L.so_imuU;
        double* imuJ = eM + 2 * L.Ncap;             // [15][30] weighted Jacobian | [15] weighted residual, in LDS (behind dx / r of the prior part)
        if (c.tid < 31) {
            ImuCtx ic;
            imu_ctx<true>(pre, x, x + 7 * K, x + 7, x + 7 * K + 9, c.gnorm, ic);
            if (c.tid == 30) {
                for (int r = 0; r < 15; ++r) {
                    double s = 0.0;
                    for (int k = r; k < 15; ++k) s += U[r * 15 + k] * ic.r[k];
                    imuJ[450 + r] = s;
                }
            } else {
                double raw[15];
                imu_raw_col(ic, pre, c.tid, raw);
                for (int r = 0; r < 15; ++r) {
                    double s = 0.0;
                    for (int k = r; k < 15; ++k) s += U[r * 15 + k] * raw[k];
                    imuJ[r * 30 + c.tid] = s;
                }
            }
        }
        __syncthreads();
        for (int wk = c.tid; wk < 930; wk += MG_NT) {
            const bool isg = wk >= 900;
            const int a = isg ? wk - 900 : wk / 30, b = isg ? 0 : wk % 30;
            auto colof = [&](int lc) {
                if (lc < 6) return mp.pose[0] + lc;
                if (lc < 15) return mp.sb[0] + lc - 6;
                if (lc < 21) return mp.pose[1] + lc - 15;
                return mp.sb[1] + lc - 21;
            };
            double s = 0.0;
            if (isg) { for (int r = 0; r < 15; ++r) s += imuJ[r * 30 + a] * imuJ[450 + r]; bv[colof(a)] += s; }
            else { for (int r = 0; r < 15; ++r) s += imuJ[r * 30 + a] * imuJ[r * 30 + b]; A[colof(a) * posmax + colof(b)] += s; }
        }
    }
    __syncthreads();
    MPROF(2);
    // ---- (c) projection factors of the landmarks anchored at frame 0, with the loss correction.
    // Records (42 doubles: r[2] | Ji[12] | Jj[12] | Jex[12] | Jl[2] | Jtd[2], rows as [row][col]) are staged in the LDS
    // area the eigen-solver uses later, in chunks of whole landmarks.  Every factor couples {pose 0, pose j, ex, td, landmark}.
    // Round 6: the CAMERA part of the sum runs on the matrix cores.  With the 2 x 16 row blocks of a factor
    //     a = [Ji (6) | Jex (6) | Jtd | r | 0 0]        b = [Jj (6) | 0 ...]
    // the sum over ALL factors of a^T a is the (pose 0, ex, td)^2 block with its gradient column, and for every target frame j
    // the sums over ITS factors of b^T a and b^T b are the (pose j) x (pose 0, ex, td | gradient) and (pose j)^2 blocks:
    // v_mfma_f64_16x16x4_f64 takes two factors (k = 4 rows) per instruction; wavefront 0 walks all factors for a^T a, the
    // others take the target frames.  The factors of a chunk are sorted by target frame first (stable counting sort by
    // ballots: the order inside a bucket is the factor order, whatever the timing -> bit-reproducible sums).  Until round 6
    // every thread owned three entries of the camera part and walked the records itself -- 2 x 200 dependent LDS reads per
    // entry of the (pose 0, ex, td) block --, one thread alone built the factor offsets (n0 dependent HBM round trips) and
    // the sort was two serial passes of one thread per bucket: 150-190K of the kernel's 590K cycles.
    if (flag == VG_MARGIN_OLD && n0 > 0) {
        const int ncam = 6 * K + 7;
        // LDS: [records cap x 42 doubles][jof cap ints][sorted list cap ints][landmark of a factor cap ints][compact factor offsets n0 + 1 ints] ... [partial tiles]
        const int cfb_ints = (n0 + 2) & ~1;
        const int cap = (4 * ld * ld - cfb_ints - 2 * MG_T0W * 256) / 87;      // (both eigen-solver tiles: 4 ld^2 ints) 84 ints of record + jof + list entry + landmark
        double* recL = eM;
        int* jofL = (int*)(recL + (size_t)cap * 42);         // [cap] target frame or -2
        int* slist = jofL + cap;                             // [cap] compact ids sorted by target frame (stable)
        int* kofL = slist + cap;                             // [cap] frame-0 landmark (index into l0) of a compact factor
        int* cfb = kofL + cap;                               // [n0 + 1] compact factor offsets
        double* t0p = recL + (size_t)(2 * ld * ld - MG_T0W * 256);      // [MG_T0W][256] partial tiles of the a^T a product (end of the area)
        int* bptr = li + MGI_BPTR;                           // [K + 2] bucket pointers
        int* wtot = li + MGI_RANK;                           // [waves] (the rank table is not in use yet)
        {
            // exclusive prefix of the landmarks' factor counts, by all threads
            int run = 0;                                     // uniform
            for (int base = 0; base < n0; base += MG_NT) {
                const int k = base + c.tid;
                int cnt = 0;
                if (k < n0) { const int l = mp.l0[k]; cnt = c.ia[L.io_lm_fbeg + l + 1] - c.ia[L.io_lm_fbeg + l]; }
                int inc = cnt;
                for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); inc += c.lane >= o ? v : 0; }
                __syncthreads();                             // (wtot of the previous trip has been read)
                if (c.lane == 63) wtot[c.wave] = inc;
                __syncthreads();
                int off = run, tot = 0;
                for (int w2 = 0; w2 < MG_NW; ++w2) { const int cw = wtot[w2]; if (w2 < c.wave) off += cw; tot += cw; }
                if (k < n0) cfb[k] = off + inc - cnt;
                run += tot;
            }
            if (c.tid == 0) cfb[n0] = run;
        }
        __syncthreads();
        MPROF(8);
        // this lane's element of a factor's row block inside a record (-1: a zero column): column = lane & 15, row = (lane >> 4) & 1;
        // lanes 0 .. 31 read the first factor of a pair, lanes 32 .. 63 the second
        const int mcol = c.lane & 15, mrow = (c.lane >> 4) & 1, msel = c.lane >> 5;
        const int offa = mcol < 6 ? 2 + 6 * mrow + mcol : (mcol < 12 ? 26 + 6 * mrow + (mcol - 6) : (mcol == 12 ? 40 + mrow : (mcol == 13 ? mrow : -1)));
        const int offb = mcol < 6 ? 14 + 6 * mrow + mcol : -1;
        // column of the system behind column t of block a (-1: not in the system, -2: the gradient column)
        auto acol = [&](int t) { return t < 6 ? mp.pose[0] + t : (t < 12 ? (cex < 0 ? -1 : cex + t - 6) : (t == 12 ? (L.t ? ctd : -1) : (t == 13 ? -2 : -1))); };
        for (int k0 = 0; k0 < n0;) {
            int k1 = n0;
            if (cfb[n0] - cfb[k0] > cap) {
                k1 = k0 + 1;
                while (k1 < n0 && cfb[k1 + 1] - cfb[k0] <= cap) ++k1;
            }
            const int cbase = cfb[k0], ncf = cfb[k1] - cbase;           // (a single landmark never exceeds cap: <= K factors)
            // which landmark a compact factor belongs to (thread per landmark of the chunk, <= K stores each), so that the evaluation
            // runs one thread per FACTOR: (landmark, slot) lanes were 60 % idle and the evaluation is bound by VALU issue
            for (int k = k0 + c.tid; k < k1; k += MG_NT)
                for (int cf = cfb[k] - cbase; cf < cfb[k + 1] - cbase; ++cf) kofL[cf] = k;
            __syncthreads();
            for (int cf = c.tid; cf < ncf; cf += MG_NT) {
                const int k = kofL[cf];
                const int l = mp.l0[k];
                const int f = c.ia[L.io_lm_fbeg + l] + cf - (cfb[k] - cbase);
                const int j = c.ia[L.io_fac_j + f];
                if (j >= K) { jofL[cf] = -2; continue; }   // relocalisation factors are not marginalised (estimator.cpp:864-903)
                jofL[cf] = j;
                const double* oi = c.di + L.do_obs + c.ia[L.io_fac_oi + f] * BA_OBS_STRIDE;
                const double* oj = c.di + L.do_obs + c.ia[L.io_fac_oj + f] * BA_OBS_STRIDE;
                double R[42];
                double Jtd[2] = {0, 0};
                if (L.t) proj_eval<true, true, true>(x, x + 7 * j, ex, lam[l], oi, oj, ex[7], c.focal, c.tr, c.row, R, R + 2, R + 14, R + 26, R + 38, Jtd);
                else proj_eval<false, true, true>(x, x + 7 * j, ex, lam[l], oi, oj, 0.0, c.focal, c.tr, c.row, R, R + 2, R + 14, R + 26, R + 38, Jtd);
                R[40] = Jtd[0]; R[41] = Jtd[1];
                const double sq = sqrt(1.0 / (1.0 + R[0] * R[0] + R[1] * R[1]));   // Cauchy: rho'' < 0 branch (:46-50)
#pragma unroll
                for (int q = 0; q < 42; ++q) recL[(size_t)cf * 42 + q] = R[q] * sq;
            }
            __syncthreads();
            MPROF(9);
            // stable counting sort of the chunk's factors by target frame: a wavefront takes the buckets j = wave, wave + MG_NW, ...
            // and finds their factors with ballots over 64-factor blocks (twice: totals, then -- behind the prefix -- positions)
            for (int j = c.wave; j < K; j += MG_NW) {
                int tot = 0;
                for (int b0 = 0; b0 < ncf; b0 += 64) {
                    const int cf = b0 + c.lane;
                    tot += __popcll(__ballot(cf < ncf && jofL[cf] == j));
                }
                if (c.lane == 0) bptr[j + 1] = tot;
            }
            __syncthreads();
            if (c.tid == 0) { bptr[0] = 0; for (int j = 0; j < K; ++j) bptr[j + 1] += bptr[j]; }
            __syncthreads();
            for (int j = c.wave; j < K; j += MG_NW) {
                int o = bptr[j];
                for (int b0 = 0; b0 < ncf; b0 += 64) {
                    const int cf = b0 + c.lane;
                    const bool mine = cf < ncf && jofL[cf] == j;
                    const unsigned long long bal = __ballot(mine);
                    if (mine) slist[o + __popcll(bal & ((1ull << c.lane) - 1ull))] = cf;
                    o += __popcll(bal);
                }
            }
            __syncthreads();
            MPROF(10);
            // ---- camera part on the matrix cores.  Wavefronts 0 and MG_NW-1 .. MG_NW-MG_T0W+1 share a^T a (the sorted list cut into
            //      MG_T0W runs of whole pairs; the partial tiles meet in LDS and wavefront 0 adds them in run order behind the chunk's last
            //      barrier), the wavefronts in between take the target frames.  Every result entry has ONE owner lane: plain
            //      read-modify-write of A / bv, the reads of a lane issued together.  Index reads are clamped instead of predicated:
            //      four independent LDS round trips per trip, not eight dependent ones.
            const int nvalid = bptr[K];
            const lds_d* recl = (const lds_d*)recL;
            const lds_i* sl_ = (const lds_i*)slist;
            glb_d* Ag = (glb_d*)A;
            glb_d* bg = (glb_d*)bv;
            const int offa_c = offa >= 0 ? offa : 0, offb_c = offb >= 0 ? offb : 0;
            const int t0seg = c.wave == 0 ? 0 : (c.wave > MG_NW - MG_T0W ? MG_NW - c.wave : -1);      // (uniform) run of the a^T a product, -1: a target-frame wavefront
            if (t0seg >= 0) {
                const int npair = (nvalid + 1) >> 1, per = (npair + MG_T0W - 1) / MG_T0W;
                const int qa = 2 * per * t0seg, qb = (2 * per * (t0seg + 1) < nvalid) ? 2 * per * (t0seg + 1) : nvalid;
                mg_d4 acc = {0, 0, 0, 0};
                for (int q = qa; q < qb; q += 8) {
                    int cf[4];
                    double v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int qq = q + 2 * u + msel; cf[u] = sl_[qq < qb ? qq : qa]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = recl[vg_mul24(cf[u], 42) + offa_c];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int qq = q + 2 * u + msel; v[u] = (qq < qb && offa >= 0) ? v[u] : 0.0; }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (q + 2 * u < qb) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v[u], v[u], acc, 0, 0, 0);      // (uniform)
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) ((lds_d*)t0p)[t0seg * 256 + reg * 64 + c.lane] = acc[reg];
            } else {
                for (int j = c.wave - 1; j < K; j += MG_NW - MG_T0W) {
                    const int q0 = bptr[j], q1 = bptr[j + 1];
                    if (q1 == q0 || mp.pose[j] < 0) continue;                          // (uniform)
                    mg_d4 aba = {0, 0, 0, 0}, abb = {0, 0, 0, 0};
                    for (int q = q0; q < q1; q += 8) {
                        int cf[4];
                        double va[4], vb[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int qq = q + 2 * u + msel; cf[u] = sl_[qq < q1 ? qq : q0]; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int ro = vg_mul24(cf[u], 42); va[u] = recl[ro + offa_c]; vb[u] = recl[ro + offb_c]; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int qq = q + 2 * u + msel;
                            va[u] = (qq < q1 && offa >= 0) ? va[u] : 0.0;
                            vb[u] = (qq < q1 && offb >= 0) ? vb[u] : 0.0;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (q + 2 * u < q1) {                                      // (uniform)
                                aba = __builtin_amdgcn_mfma_f64_16x16x4f64(vb[u], va[u], aba, 0, 0, 0);
                                abb = __builtin_amdgcn_mfma_f64_16x16x4f64(vb[u], vb[u], abb, 0, 0, 0);
                            }
                        }
                    }
                    // rows 0 .. 5 of both products = the columns of pose j: the accumulator entries reg 0 (rows 0 .. 3) and reg 1 (rows 4, 5)
                    const int cj = acol(mcol), pj = mp.pose[j];
                    const int r0i = pj + (c.lane >> 4), r1i = r0i + 4;
                    const bool has1 = (c.lane >> 4) < 2;                                // rows 4, 5
                    if (cj >= 0) {
                        const double o0 = Ag[(size_t)r0i * posmax + cj], o1 = Ag[(size_t)cj * posmax + r0i];
                        const double o2 = has1 ? Ag[(size_t)r1i * posmax + cj] : 0.0, o3 = has1 ? Ag[(size_t)cj * posmax + r1i] : 0.0;
                        Ag[(size_t)r0i * posmax + cj] = o0 + aba[0]; Ag[(size_t)cj * posmax + r0i] = o1 + aba[0];
                        if (has1) { Ag[(size_t)r1i * posmax + cj] = o2 + aba[1]; Ag[(size_t)cj * posmax + r1i] = o3 + aba[1]; }
                    } else if (cj == -2) {
                        const double o0 = bg[r0i], o2 = has1 ? bg[r1i] : 0.0;
                        bg[r0i] = o0 + aba[0];
                        if (has1) bg[r1i] = o2 + aba[1];
                    }
                    if (mcol < 6) {
                        const double o0 = Ag[(size_t)r0i * posmax + pj + mcol], o2 = has1 ? Ag[(size_t)r1i * posmax + pj + mcol] : 0.0;
                        Ag[(size_t)r0i * posmax + pj + mcol] = o0 + abb[0];
                        if (has1) Ag[(size_t)r1i * posmax + pj + mcol] = o2 + abb[1];
                    }
                }
            }
            MPROF(11);
            // landmark rows / columns (round 6, second form).  The entries a landmark shares with the pose of a TARGET frame have one
            // term -- the factor of that frame -- so they are a thread per (factor, column of its target pose), no loop; what sums over
            // the landmark's factors (pose 0, extrinsic pose, td, the landmark's own diagonal entry and its gradient entry: 15 per
            // landmark) is a thread per entry walking the factors in their order.  Entries of frames the landmark is not seen from
            // are not written at all (A was cleared).  It was a thread per (landmark, every camera column) with the factor loop in
            // each: 5400 looped tasks for 72 landmarks, most of them storing zeros.
            const int lmb = mp.misc[7];
            for (int wk = c.tid; wk < 6 * ncf; wk += MG_NT) {
                const int cf = wk / 6, kk = wk - 6 * cf;
                const int j = jofL[cf];
                if (j < 0 || mp.pose[j] < 0) continue;
                const lds_d* R = recl + vg_mul24(cf, 42);
                const double v = R[14 + kk] * R[38] + R[20 + kk] * R[39];
                const int cl = lmb + kofL[cf], ca = mp.pose[j] + kk;
                Ag[(size_t)cl * posmax + ca] = v; Ag[(size_t)ca * posmax + cl] = v;
            }
            for (int wk = c.tid; wk < 15 * (k1 - k0); wk += MG_NT) {
                const int k = k0 + wk / 15, e = wk - 15 * (wk / 15);
                int ca, o0, o1;
                if (e < 6) { ca = mp.pose[0] + e; o0 = 2 + e; o1 = 8 + e; }
                else if (e < 12) { ca = cex < 0 ? -1 : cex + e - 6; o0 = 26 + e - 6; o1 = 32 + e - 6; }
                else if (e == 12) { ca = L.t ? ctd : -1; o0 = 40; o1 = 41; }
                else if (e == 13) { ca = -2; o0 = 38; o1 = 39; }          // (l, l)
                else { ca = -3; o0 = 0; o1 = 1; }                          // gradient
                if (ca == -1) continue;
                double sacc = 0.0;
                const int fb = cfb[k] - cbase, fe = cfb[k + 1] - cbase;
                for (int cf = fb; cf < fe; ++cf) {
                    const lds_d* R = recl + vg_mul24(cf, 42);
                    const double v = R[o0] * R[38] + R[o1] * R[39];
                    sacc += jofL[cf] >= 0 ? v : 0.0;
                }
                const int cl = lmb + k;
                if (ca == -2) Ag[(size_t)cl * posmax + cl] = sacc;
                else if (ca == -3) bg[cl] = sacc;
                else { Ag[(size_t)cl * posmax + ca] = sacc; Ag[(size_t)ca * posmax + cl] = sacc; }
            }
            __syncthreads();
            if (c.wave == 0) {
                // a^T a of the chunk: the MG_T0W partial tiles in run order (the next chunk does not write them before its own three barriers)
                const int cj = acol(mcol);
                double tv[4], ov[4];
                int ci[4];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    double sacc = ((const lds_d*)t0p)[reg * 64 + c.lane];
#pragma unroll
                    for (int sg = 1; sg < MG_T0W; ++sg) sacc += ((const lds_d*)t0p)[sg * 256 + reg * 64 + c.lane];
                    tv[reg] = sacc;
                    ci[reg] = acol((c.lane >> 4) + 4 * reg);                           // D[row = (lane >> 4) + 4 reg][col = lane & 15]
                }
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    ov[reg] = ci[reg] < 0 ? 0.0 : (cj >= 0 ? Ag[(size_t)ci[reg] * posmax + cj] : (cj == -2 ? bg[ci[reg]] : 0.0));
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    if (ci[reg] < 0) continue;
                    if (cj >= 0) Ag[(size_t)ci[reg] * posmax + cj] = ov[reg] + tv[reg];
                    else if (cj == -2) bg[ci[reg]] = ov[reg] + tv[reg];
                }
            }
            k0 = k1;
        }
    }
    __syncthreads();

    MPROF(3);
    // ---- M4: Amm^+ [Amr | bmm], Schur complement
    bool direct = false;                   // (uniform) A' / b' already stand in the square root's LDS tile / gV
    {
        const bool in_lds = m <= ld;
        double* Mm = in_lds ? eM : gM;
        double* Vm = in_lds ? eV : gV;
        const int ldm = in_lds ? ld : posmax;
        const int offcs = (int)(cs - MG_LDS), offred = (int)(red - MG_LDS);
        // ---- certified block elimination.  The dropped block is [pose 0 | speed-bias 0 | frame-0 landmarks] and the landmark
        // part of Amm is DIAGONAL (a projection factor touches one landmark):  Amm = [P W; W^T D].  With the md x md Schur
        // complement P' = P - W D^-1 W^T (md <= 15)
        //     X1 = P'^-1 (R1 - W D^-1 R2),   X2 = D^-1 (R2 - W^T X1)       solves  Amm X = [Amr | bmm] = [R1; R2]
        // in O(md ml n) operations for ANY number of frame-0 landmarks, without the m x m eigen-decomposition.
        // marginalization_factor.cpp:272-283 takes the eigen pseudo-inverse with the lambda > 1e-8 cut, which IS the inverse
        // when every eigenvalue exceeds 1e-8; that is certified here by  lambda_min(Amm) >= 1 / ||Amm^-1||_F > 1e-7, the norm
        // summed block by block (Amm^-1 = [B11 B12; B12^T B22], B11 = P'^-1, B12 = -P'^-1 W D^-1, B22 = D^-1 - D^-1 W^T B12)
        // without storing B22.  Otherwise (or if P' is not positive definite) the eigen-decomposition below runs.
        const int md = m - n0, ml = n0;
        // ---- Round 6: the Schur complement of the dropped block in two stages on the matrix cores, without Amm^-1 [Amr | bmm].
        // With C = the md dropped camera columns followed by the n kept columns and the gradient (q = md + n + 1 indices), Z = the
        // frame-0 landmarks' rows of A restricted to C (ml x q) and D their diagonal:
        //     G' = G - Z^T D^-1 Z                              (the landmarks leave: rank-ml update of the q x q system, MFMA, k = ml)
        //     P' = G'[0:md, 0:md] = L L^T,  X = L^-1 G'[0:md, md:]
        //     [A' | b'] = G'[md:, md:] - X^T X                 (the dropped pose / speed-bias leave: MFMA, k = md <= 16)
        // which is the same elimination as X1 / X2 below, with every m-term and ml-term dot product on v_mfma_f64_16x16x4 and the
        // result written straight into the LDS tile the square root works on (until round 6: T2 through global memory, A' staged in
        // global memory and copied back -- 190K of the kernel's 460K cycles).  The certificate lambda_min(Amm) > 1e-7 is kept:
        // ||Amm^-1||_F^2 = ||P'^-1||^2 + 2 ||P'^-1 W D^-1||^2 + ||D^-1 + Y^T Y||^2 with Y = L^-1 W D^-1, and
        // ||D^-1 + Y^T Y||_F^2 = sum_l (d_l^-2 + 2 d_l^-1 |y_l|^2) + ||Y Y^T||_F^2 -- an md x md product instead of ml^2 entries.
        {
            const int q = md + n + 1, qp = 16 * ((q + 15) >> 4), np1 = n + 1, ldx = 16 * ((np1 + 15) >> 4);
            const bool fits = md >= 1 && md <= 16 && qp <= ld && n <= ld && n * ld + 16 * ldx <= ld * ld
                              && 2 * md * ml + ml <= n * ld && c.hdr[H_MARGMODE] == 0 && n <= 16 * MG_HROWS;
            if (fits) {
                lds_d* Gp = (lds_d*)eV;                      // [qp][ld]   G' (lower tiles)
                lds_d* Wl = (lds_d*)eM;                      // [md][ml]   W = A[dropped camera][landmark]
                lds_d* dinv = Wl + md * ml;                  // [ml]       1 / d_l
                lds_d* Ys = dinv + ml;                       // [md][ml]   Y = L^-1 W D^-1
                lds_d* Xs = (lds_d*)eM + n * ld;             // [16][ldx]  X = L^-1 U, rows md .. 15 and columns n + 1 .. zero
                lds_d* Ls = (lds_d*)cs;                      // [16][16]   L, then L^-1 (lower)
                lds_d* Pi = Ls + 256;                        // [md][md]   P'^-1
                int* okf = (int*)(red + 18);
                const glb_d* Ag = (const glb_d*)A;
                const glb_d* bg = (const glb_d*)bv;
                if (c.tid == 0) *okf = 1;
                double bad = 0.0;
                for (int l = c.tid; l < ml; l += MG_NT) {
                    const double d = Ag[(size_t)(md + l) * posmax + md + l];
                    bad += (d > 0.0 && d < 1e300) ? 0.0 : 1.0;
                    dinv[l] = 1.0 / d;
                }
                for (int k = c.tid; k < md * ml; k += MG_NT) {
                    const int p = k / ml, l = k - p * ml;
                    Wl[k] = Ag[(size_t)(md + l) * posmax + p];
                }
                __syncthreads();
                // ---- G' = G - Z^T D^-1 Z: a wavefront takes the lower 16 x 16 tiles t = wave, wave + MG_NW, ...
                {
                    const int qt = qp >> 4, ntile = qt * (qt + 1) / 2;
                    const int jc = c.lane & 15, kq = c.lane >> 4;
                    for (int t = __builtin_amdgcn_readfirstlane(c.wave); t < ntile; t += MG_NW) {
                        int ti, tj;
                        tri_decode(t, ti, tj);
                        const int ia = 16 * ti + jc, ib = 16 * tj + jc;                     // C indices of this lane's operand columns
                        const int ca = ia < md ? ia : m + ia - md, cb = ib < md ? ib : m + ib - md;      // their columns in A
                        const bool ga = ia == md + n, gb = ib == md + n;                   // the gradient "column"
                        const bool za = ia > md + n, zb = ib > md + n;
                        mg_d4 acc = {0, 0, 0, 0};
                        for (int l0 = 0; l0 < ml; l0 += 16) {
                            double av[4], bv4[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int l = l0 + 4 * u + kq;
                                const int lc = l < ml ? l : 0;
                                const double za_ = ga ? bg[md + lc] : Ag[(size_t)(md + lc) * posmax + (za ? 0 : ca)];
                                const double zb_ = gb ? bg[md + lc] : Ag[(size_t)(md + lc) * posmax + (zb ? 0 : cb)];
                                av[u] = (l < ml && !za) ? za_ * dinv[lc] : 0.0;
                                bv4[u] = (l < ml && !zb) ? zb_ : 0.0;
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (l0 + 4 * u < ml) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv4[u], acc, 0, 0, 0);      // (uniform)
                        }
                        double gv[4];
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int i = 16 * ti + kq + 4 * reg;                           // D[row = kq + 4 reg][col = jc]
                            const int ci = i < md ? i : m + i - md;
                            const bool gi = i == md + n, zi = i > md + n;
                            // G[i][ib]: A (symmetric: the lower entry), the gradient for the last index, zero on the padding
                            double g = 0.0;
                            if (!zi && !zb && !(gi && gb)) {
                                if (gi) g = bg[cb];
                                else if (gb) g = bg[ci];
                                else g = Ag[(size_t)(ci > cb ? ci : cb) * posmax + (ci > cb ? cb : ci)];
                            }
                            gv[reg] = g;
                        }
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) Gp[(16 * ti + kq + 4 * reg) * ld + ib] = gv[reg] - acc[reg];
                    }
                }
                __syncthreads();
                MPROF(12);
                // ---- P' = L L^T in registers (lane = row), then L^-1 by lane = column (wavefront 0)
                if (c.wave == 0) {
                    double a[16];
                    const int i = c.lane & 15;
#pragma unroll
                    for (int qq = 0; qq < 16; ++qq) a[qq] = (c.lane < md && qq <= i && qq < md) ? Gp[i * ld + qq] : 0.0;
                    bool good = true;
#pragma unroll
                    for (int jj = 0; jj < 16; ++jj) {
                        if (jj < md) {
                            const double piv = readlane_f64(a[jj], jj);
                            if (!(piv > 0.0) || !(piv < 1e300)) good = false;
                            const double di = mg_rsqrt(piv);
                            const double l = a[jj] * di;
                            a[jj] = l;
#pragma unroll
                            for (int qq = jj + 1; qq < 16; ++qq) { const double lq = readlane_f64(l, qq); a[qq] -= l * lq; }
                        }
                    }
                    if (c.lane < md) {
#pragma unroll
                        for (int qq = 0; qq < 16; ++qq) if (qq <= i && qq < md) Ls[i * 16 + qq] = a[qq];
                    }
                    __builtin_amdgcn_wave_barrier();
                    // column j of L^-1 (lower): forward substitution on e_j
                    double xx[16];
                    const int j = c.lane & 15;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        double sacc = (r == j) ? 1.0 : 0.0;
#pragma unroll
                        for (int qq = 0; qq < r; ++qq) sacc -= (r < md && qq >= j) ? Ls[r * 16 + qq] * xx[qq] : 0.0;
                        xx[r] = (r < md && r >= j) ? sacc / Ls[r * 16 + r] : 0.0;
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (c.lane < 16) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) Ls[r * 16 + j] = (r < md && j < md) ? xx[r] : 0.0;      // Ls <- L^-1 (lower, zeros elsewhere)
                    }
                    if (!good) *okf = 0;
                }
                __syncthreads();
                MPROF(13);
                // ---- X = L^-1 U (U[p][j] = G'[md + j][p]), Y = L^-1 W D^-1, P'^-1 = L^-T L^-1
                for (int k = c.tid; k < 16 * ldx; k += MG_NT) {
                    const int p = k / ldx, j = k - p * ldx;
                    double sacc = 0.0;
                    if (p < md && j < np1)
                        for (int qq = 0; qq <= p; ++qq) sacc += Ls[p * 16 + qq] * Gp[(md + j) * ld + qq];
                    Xs[k] = sacc;
                }
                for (int k = c.tid; k < md * ml; k += MG_NT) {
                    const int p = k / ml, l = k - p * ml;
                    double sacc = 0.0;
                    for (int qq = 0; qq <= p; ++qq) sacc += Ls[p * 16 + qq] * Wl[qq * ml + l];
                    Ys[k] = sacc * dinv[l];
                }
                for (int k = c.tid; k < md * md; k += MG_NT) {
                    const int i = k / md, j = k - i * md;
                    double sacc = 0.0;
                    for (int r = (i > j ? i : j); r < md; ++r) sacc += Ls[r * 16 + i] * Ls[r * 16 + j];
                    Pi[k] = sacc;
                }
                __syncthreads();
                MPROF(14);
                // ---- the certificate
                double fro = 0.0;
                for (int k = c.tid; k < md * md; k += MG_NT) {
                    const double v = Pi[k];
                    fro += v * v; bad += (v == v && fabs(v) < 1e300) ? 0.0 : 1.0;
                    const int i = k / md, j = k - i * md;
                    double mm = 0.0;                                                     // (Y Y^T)[i][j]
                    for (int l = 0; l < ml; ++l) mm += Ys[i * ml + l] * Ys[j * ml + l];
                    fro += mm * mm; bad += (mm == mm && fabs(mm) < 1e300) ? 0.0 : 1.0;
                }
                for (int k = c.tid; k < md * ml; k += MG_NT) {                           // B12 = -P'^-1 W D^-1, entry by entry
                    const int p = k / ml, l = k - p * ml;
                    double sacc = 0.0;
                    for (int qq = 0; qq < md; ++qq) sacc += Pi[p * md + qq] * Wl[qq * ml + l];
                    const double v = sacc * dinv[l];
                    fro += 2.0 * v * v; bad += (v == v && fabs(v) < 1e300) ? 0.0 : 1.0;
                }
                for (int l = c.tid; l < ml; l += MG_NT) {
                    double yn = 0.0;
                    for (int p = 0; p < md; ++p) yn += Ys[p * ml + l] * Ys[p * ml + l];
                    fro += dinv[l] * dinv[l] + 2.0 * dinv[l] * yn;
                }
                fro = mg_block_sum(c, red, fro);
                bad = mg_block_sum(c, red, bad);
                direct = *okf != 0 && bad == 0.0 && fro > 0.0 && fro < 1e14;              // 1 / sqrt(fro) > 1e-7
                __syncthreads();
                MPROF(15);
                if (direct) {
                    // ---- [A' | b'] = G'[md:, md:] - X^T X  ->  the square root's tile (eM, leading dimension ld) and b' (gV)
                    const int nt = ldx >> 4, ntile = nt * (nt + 1) / 2;
                    const int jc = c.lane & 15, kq = c.lane >> 4;
                    for (int t = __builtin_amdgcn_readfirstlane(c.wave); t < ntile; t += MG_NW) {
                        int ti, tj;
                        tri_decode(t, ti, tj);
                        mg_d4 acc = {0, 0, 0, 0};
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xs[(4 * u + kq) * ldx + 16 * ti + jc], Xs[(4 * u + kq) * ldx + 16 * tj + jc], acc, 0, 0, 0);
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int i = 16 * ti + kq + 4 * reg, j = 16 * tj + jc;       // D[row = kq + 4 reg][col = jc]
                            if (i > n || j > n || j > i) continue;                        // (lower triangle incl. the gradient row i = n)
                            const double v = Gp[(md + i) * ld + md + j] - acc[reg];
                            if (i < n) { eM[i * ld + j] = v; eM[j * ld + i] = v; }
                            else if (j < n) gV[j] = v;
                        }
                    }
                    if (c.tid == 0) mi[4] = 0;       // no sweeps: the dropped block left by block elimination
                    MPROF(4);
                    __syncthreads();
                }
            }
        }
        if (!direct) {
        bool inverse_ok = false;
        if (md >= 1 && md <= 16 && md * ml <= ld * ld) {
            double* dl = cs;                 // [ml <= Lcap] landmark diagonal        (mg_cs >= Lcap + 512 checked on the host)
            double* Ls = cs + L.Lcap;        // [md][md] Cholesky factor of P', then its inverse
            double* Pi = Ls + 256;           // [md][md] P'^-1
            double* Wl = eM;                 // [md][ml]
            double* B12 = eV;                // [md][ml]
            int* okf = (int*)(red + 18);
            if (c.tid == 0) *okf = 1;
            for (int l = c.tid; l < ml; l += MG_NT) dl[l] = A[(size_t)(md + l) * posmax + md + l];
            for (int k = c.tid; k < md * ml; k += MG_NT) {
                const int p = k / ml, l = k - p * ml;
                Wl[k] = 0.5 * (A[(size_t)p * posmax + md + l] + A[(size_t)(md + l) * posmax + p]);
            }
            __syncthreads();
            for (int k = c.tid; k < md * md; k += MG_NT) {
                const int i = k / md, j = k - i * md;
                double sacc = 0.5 * (A[(size_t)i * posmax + j] + A[(size_t)j * posmax + i]);
                for (int l = 0; l < ml; ++l) sacc -= Wl[i * ml + l] * Wl[j * ml + l] / dl[l];
                Ls[i * md + j] = sacc;
            }
            __syncthreads();
            if (c.wave == 0) {               // Cholesky of P' in registers (lane = row), then L^-1 by lane = column
                double a[16];
                const int i = c.lane & 15;
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = (c.lane < md && q <= i && q < md) ? Ls[i * md + q] : 0.0;
                bool good = true;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    if (jj < md) {
                        const double piv = readlane_f64(a[jj], jj);
                        if (!(piv > 0.0) || !(piv < 1e300)) good = false;
                        const double dinv = mg_rsqrt(piv);
                        const double l = a[jj] * dinv;
                        a[jj] = l;
#pragma unroll
                        for (int q = jj + 1; q < 16; ++q) { const double lq = readlane_f64(l, q); a[q] -= l * lq; }
                    }
                }
                if (c.lane < md) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) if (q <= i && q < md) Ls[i * md + q] = a[q];
                }
                __builtin_amdgcn_wave_barrier();
                // column j of L^-1 (lower): forward substitution on e_j
                double x[16];
                const int j = c.lane & 15;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    double sacc = (r == j) ? 1.0 : 0.0;
#pragma unroll
                    for (int q = 0; q < r; ++q) sacc -= (r < md && q >= j) ? Ls[r * md + q] * x[q] : 0.0;
                    x[r] = (r < md && r >= j) ? sacc / Ls[r * md + r] : 0.0;
                }
                __builtin_amdgcn_wave_barrier();
                if (c.lane < md) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) if (r < md) Ls[r * md + j] = x[r];      // Ls <- L^-1 (lower, zeros above)
                }
                for (int l = c.lane; l < ml; l += 64) if (!(dl[l] > 0.0)) good = false;
                if (!good) *okf = 0;
            }
            __syncthreads();
            for (int k = c.tid; k < md * md; k += MG_NT) {          // P'^-1 = L^-T L^-1
                const int i = k / md, j = k - i * md;
                double sacc = 0.0;
                for (int r = (i > j ? i : j); r < md; ++r) sacc += Ls[r * md + i] * Ls[r * md + j];
                Pi[k] = sacc;
            }
            __syncthreads();
            for (int k = c.tid; k < md * ml; k += MG_NT) {          // B12 = -P'^-1 W D^-1
                const int p = k / ml, l = k - p * ml;
                double sacc = 0.0;
                for (int q = 0; q < md; ++q) sacc += Pi[p * md + q] * Wl[q * ml + l];
                B12[k] = -sacc / dl[l];
            }
            __syncthreads();
            double fro = 0.0, bad = 0.0;
            for (int k = c.tid; k < md * md; k += MG_NT) { const double v = Pi[k]; fro += v * v; bad += (v == v && fabs(v) < 1e300) ? 0.0 : 1.0; }
            for (int k = c.tid; k < md * ml; k += MG_NT) { const double v = B12[k]; fro += 2.0 * v * v; bad += (v == v && fabs(v) < 1e300) ? 0.0 : 1.0; }
            for (int k = c.tid; k < ml * ml; k += MG_NT) {          // B22 = D^-1 - D^-1 W^T B12, entry by entry
                const int a_ = k / ml, b_ = k - a_ * ml;
                double sacc = 0.0;
                for (int p = 0; p < md; ++p) sacc += Wl[p * ml + a_] * B12[p * ml + b_];
                const double v = (a_ == b_ ? 1.0 / dl[a_] : 0.0) - sacc / dl[a_];
                fro += v * v;
                bad += (v == v && fabs(v) < 1e300) ? 0.0 : 1.0;
            }
            fro = mg_block_sum(c, red, fro);
            bad = mg_block_sum(c, red, bad);
            inverse_ok = *okf != 0 && bad == 0.0 && fro > 0.0 && fro < 1e14;      // 1 / sqrt(fro) > 1e-7
            __syncthreads();
            if (inverse_ok) {
                // T2 = Amm^-1 [Amr | bmm]: column j of the right-hand side is A[:, m + j] (j < n) or bv (j == n)
                const int nc = n + 1;
                // R = [Amr | bmm] staged in LDS behind W (rows 0 .. m-1, nc columns) when it fits, u / X1 behind B12: the loops
                // below are dot products over ml or md terms per entry — through global memory a chain of round trips each
                const bool stage = md * ml + m * nc <= ld * ld && md * ml + md * nc <= ld * ld;
                double* Rs = Wl + md * ml;       // [m][nc]
                double* U = stage ? B12 + md * ml : T1;      // [md][nc]  u = R1 - W D^-1 R2, then X1 = P'^-1 u
                double* X1 = stage ? U : T2;
                const int ldx = stage ? nc : mcap + 1;
                if (stage) {
                    for (int k = c.tid; k < m * nc; k += MG_NT) {
                        const int r = k / nc, j = k - r * nc;
                        Rs[k] = j < n ? A[(size_t)r * posmax + m + j] : bv[r];
                    }
                    __syncthreads();
                }
                for (int k = c.tid; k < md * nc; k += MG_NT) {
                    const int p = k / nc, j = k - p * nc;
                    double sacc;
                    if (stage) {
                        sacc = Rs[p * nc + j];
                        for (int l = 0; l < ml; ++l) sacc -= Wl[p * ml + l] * Rs[(md + l) * nc + j] / dl[l];
                    } else {
                        sacc = j < n ? A[(size_t)p * posmax + m + j] : bv[p];
                        for (int l = 0; l < ml; ++l) {
                            const double r2 = j < n ? A[(size_t)(md + l) * posmax + m + j] : bv[md + l];
                            sacc -= Wl[p * ml + l] * r2 / dl[l];
                        }
                    }
                    U[p * nc + j] = sacc;
                }
                __syncthreads();
                for (int k = c.tid; k < md * nc; k += MG_NT) {
                    const int p = k / nc, j = k - p * nc;
                    double sacc = 0.0;
                    for (int q = 0; q < md; ++q) sacc += Pi[p * md + q] * U[q * nc + j];
                    T2[p * (mcap + 1) + j] = sacc;
                }
                __syncthreads();
                if (stage) {                     // X1 into LDS (U's place) for the last loop
                    for (int k = c.tid; k < md * nc; k += MG_NT) { const int p = k / nc, j = k - p * nc; X1[p * nc + j] = T2[p * (mcap + 1) + j]; }
                    __syncthreads();
                }
                for (int k = c.tid; k < ml * nc; k += MG_NT) {
                    const int l = k / nc, j = k - l * nc;
                    double sacc = stage ? Rs[(md + l) * nc + j] : (j < n ? A[(size_t)(md + l) * posmax + m + j] : bv[md + l]);
                    for (int p = 0; p < md; ++p) sacc -= Wl[p * ml + l] * X1[p * ldx + j];
                    T2[(md + l) * (mcap + 1) + j] = sacc / dl[l];
                }
                __syncthreads();
            }
        }
        if (inverse_ok) {
            if (c.tid == 0) mi[4] = 0;       // no sweeps: Amm^-1 by certified block elimination
            MPROF(4);
        } else {
#ifdef BA_PROFILE
        const long long _t1 = clock64();
#endif
        for (int k = c.tid; k < m * m; k += MG_NT) {
            const int i = k / m, j = k % m;
            Mm[i * ldm + j] = 0.5 * (A[i * posmax + j] + A[j * posmax + i]);
        }
        __syncthreads();
        const bool fast1 = in_lds && m >= 1 && m <= 16 * MG_HROWS && (m + 1) / 2 <= MG_NT / 16;      // (vh_eig: a 16-lane group per column pair)
        const int sw1 = fast1 ? vh_eig(c, 0, ld * ld, m, ldm, offcs, offred, 0.0, 2e-16, 0.0, 10)
                      : in_lds ? jacobi_eig<true>(c, nullptr, nullptr, 0, ld * ld, m, ldm, offcs, offred, true)
                               : jacobi_eig<false>(c, Mm, Vm, 0, 0, m, ldm, offcs, offred, true);
#ifdef BA_PROFILE
        if (c.tid == 0) { mi[6] = (int)((clock64() - _t1) >> 10); mi[7] = (int)((_t1 - _tstart) >> 10); }
#endif
        if (c.tid == 0) mi[4] = sw1;        // sweeps | Cholesky attempts << 8 of the Amm eigen-decomposition
        MPROF(4);
        // T1 = Lambda^+ V^T [Amr | bmm]   (m x (n+1))
        for (int k = c.tid; k < m * (n + 1); k += MG_NT) {
            const int i = k / (n + 1), j = k % (n + 1);
            const double lamb = Mm[i * ldm + i];
            double s = 0.0;
            if (lamb > MG_EPS) {
                const double* src = j < n ? A + m + j : bv;          // column j of [Amr | bmm]
                const int sst = j < n ? posmax : 1;
#pragma unroll 8
                for (int r = 0; r < m; ++r) s += Vm[r * ldm + i] * src[(size_t)r * sst];
                s /= lamb;
            }
            T1[i * (mcap + 1) + j] = s;
        }
        __syncthreads();
        // T2 = V T1  = Amm^+ [Amr | bmm]
        for (int k = c.tid; k < m * (n + 1); k += MG_NT) {
            const int i = k / (n + 1), j = k % (n + 1);
            double s = 0.0;
#pragma unroll 8
            for (int r = 0; r < m; ++r) s += Vm[i * ldm + r] * T1[r * (mcap + 1) + j];
            T2[i * (mcap + 1) + j] = s;
        }
        __syncthreads();
        }   // eigen path
        // A' = Arr - Arm T2[:, :n] ; b' = brr - Arm T2[:, n]   -> staged in global (gM, gV)
        if (n * m <= ld * ld && m * (n + 1) <= ld * ld) {
            // both operands through LDS (the two eigen-solver tiles are free here): the m-term dot product of every entry used to
            // be a chain of global round trips (125K of the kernel's 850K cycles)
            double* Arm = eM;                    // [n][m]
            double* T2s = eV;                    // [m][n + 1]
            __syncthreads();
            for (int k = c.tid; k < n * m; k += MG_NT) { const int i = k / m, r = k - i * m; Arm[k] = A[(m + i) * posmax + r]; }
            for (int k = c.tid; k < m * (n + 1); k += MG_NT) { const int r = k / (n + 1), j = k - r * (n + 1); T2s[k] = T2[r * (mcap + 1) + j]; }
            __syncthreads();
            for (int k = c.tid; k < n * (n + 1); k += MG_NT) {
                const int i = k / (n + 1), j = k % (n + 1);
                double s = (j < n) ? A[(m + i) * posmax + m + j] : bv[m + i];
                const double* ar = Arm + i * m;
                const double* tc = T2s + j;
#pragma unroll 8
                for (int r = 0; r < m; ++r) s -= ar[r] * tc[r * (n + 1)];
                if (j < n) gM[i * posmax + j] = s; else gV[i] = s;
            }
        } else {
            for (int k = c.tid; k < n * (n + 1); k += MG_NT) {
                const int i = k / (n + 1), j = k % (n + 1);
                double s = (j < n) ? A[(m + i) * posmax + m + j] : bv[m + i];
#pragma unroll 8
                for (int r = 0; r < m; ++r) s -= A[(m + i) * posmax + r] * T2[r * (mcap + 1) + j];
                if (j < n) gM[i * posmax + j] = s; else gV[i] = s;      // stage in global (eM may still be Mm)
            }
        }
        __syncthreads();
        }   // !direct
    }
    MPROF(5);
    double* bp = gV;                       // b' (n)
    const bool n_lds = n <= ld;
    double* M2 = n_lds ? eM : g2M;
    double* V2 = n_lds ? eV : g2V;
    const int ld2 = n_lds ? ld : n;
    if (!direct)
        for (int k = c.tid; k < n * n; k += MG_NT) M2[(k / n) * ld2 + k % n] = gM[(k / n) * posmax + k % n];
    __syncthreads();
#ifdef BA_PROFILE
    const long long _t2 = clock64();
#endif
    const int offcs2 = (int)(cs - MG_LDS), offred2 = (int)(red - MG_LDS);
    const bool fast2 = n_lds && n >= 1 && n <= 16 * MG_HROWS;
    const bool fast2_eig = fast2 && (n + 1) / 2 <= MG_NT / 16;      // (vh_eig: a 16-lane group per column pair; the square root has no such limit)
    if ((fast2 || !n_lds) && c.hdr[H_MARGMODE] == 0) {
        // square-root form by pivoted Cholesky (see sqrt_factor): J0 = L^T, r0 = L^-1 b'
        const int rk = fast2 ? sqrt_factor(c, 0, ld * ld, n, ld2, offcs2, offred2, bp)
                             : sqrt_factor_glb(c, M2, V2, n, ld2, offcs2, bp);
        const double* yv = cs + 2 * n + 1;
        if (c.tid == 0) mi[5] = rk << 16;        // rank of the factor (the eigen path reports sweeps | attempts << 8 here)
        MPROF(6);
        for (int k = c.tid; k < n * n; k += MG_NT) {
            const int i = k / n, j = k - i * n;  // row i of J0 = column i of L
            mo[L.mo_J0 + (size_t)i * mcap + j] = i < rk ? V2[i * ld2 + j] : 0.0;
        }
        for (int i = c.tid; i < n; i += MG_NT) mo[L.mo_r0 + i] = i < rk ? yv[i] : 0.0;
    } else {
    // Column orthogonality 1e-9 (relative).  The reconstruction sum_i (|g_i|^2 - delta) u_i u_i^T = A does not depend on how
    // far the sweeps went (G G^T is invariant under the rotations); what converges are the individual eigenvalues, which
    // only matter for the eps = 1e-8 cut: theta = 1e-9 moves a small eigenvalue by ~theta^2 lambda_max ~ 1e-11, two decades
    // below the ~u |A| noise of the reference's tridiagonal QR on the same matrix.  Measured on the EuRoC-shape windows: the
    // largest |g_p.g_q| / (|g_p||g_q|) per sweep runs 0.66, 0.44, 0.41, 6e-2, 7e-3, 1e-5, 9e-8, 8e-9, 1e-10 -- each further
    // decade costs a full sweep of 75 rounds.  The last of them used to be a pure verification sweep (no rotation): the loop
    // now ends after the first sweep whose largest rotated pair was below 1e-7 (what it leaves is ~theta^2).
    const int sw2 = fast2_eig ? vh_eig(c, 0, ld * ld, n, ld2, offcs2, offred2, 2e-8, 1e-9, 1e-7, 16)
                  : n_lds ? jacobi_eig<true>(c, nullptr, nullptr, 0, ld * ld, n, ld2, offcs2, offred2, false)
                          : jacobi_eig<false>(c, M2, V2, 0, 0, n, ld2, offcs2, offred2, false);
#ifdef BA_PROFILE
    if (c.tid == 0) { mi[6] = (int)red[20] >> 10; mi[7] = (int)red[21] >> 10;
                      for (int k = 0; k < 8; ++k) mprof[8 + k] = (int)(red[20 + k] / 1024.0); }
#else
    (void)sw2;
#endif
    if (c.tid == 0) mi[5] = sw2;            // sweeps | Cholesky attempts << 8 of the kept-block eigen-decomposition
    MPROF(6);
    // ascending order like SelfAdjointEigenSolver: rank of each eigenvalue
    int* rank = li + MGI_RANK;
    for (int i = c.tid; i < n; i += MG_NT) {
        const double li_ = M2[i * ld2 + i];
        int rk = 0;
        for (int j = 0; j < n; ++j) {
            const double lj = M2[j * ld2 + j];
            rk += (lj < li_ || (lj == li_ && j < i)) ? 1 : 0;
        }
        rank[i] = rk;
    }
    __syncthreads();
    // linearized_jacobians = diag(sqrt(S)) V^T ; linearized_residuals = diag(sqrt(S_inv)) V^T b'
    for (int k = c.tid; k < n * n; k += MG_NT) {
        const int i = k / n, j = k % n;           // eigenpair i, column j
        const double lamb = M2[i * ld2 + i];
        const double sv = lamb > MG_EPS ? sqrt(lamb) : 0.0;
        mo[L.mo_J0 + (size_t)rank[i] * mcap + j] = sv * V2[j * ld2 + i];
    }
    for (int i = c.tid; i < n; i += MG_NT) {
        const double lamb = M2[i * ld2 + i];
        double s = 0.0;
        if (lamb > MG_EPS) {
            for (int r = 0; r < n; ++r) s += V2[r * ld2 + i] * bp[r];
            s *= sqrt(1.0 / lamb);
        }
        mo[L.mo_r0 + rank[i]] = s;
    }
    }   // eigen form
    MPROF(7);
    // ---- M5: kept blocks, re-labelled for the slid window, with their linearisation point
    if (c.tid == 0) {
        int nb = 0, x0o = 0;
        int* kind = mi + 8;
        int* idx = mi + 8 + (K + 4);
        double* x0 = mo + L.mo_x0;
        for (int i = 0; i < K; ++i) {
            if (mp.pose[i] < m) continue;
            kind[nb] = VG_BLK_POSE;
            idx[nb] = (flag == VG_MARGIN_OLD) ? i - 1 : (i == K - 1 ? i - 1 : i);
            for (int k = 0; k < 7; ++k) x0[x0o + k] = x[7 * i + k];
            x0o += 7; ++nb;
        }
        for (int i = 0; i < K; ++i) {
            if (mp.sb[i] < m) continue;
            kind[nb] = VG_BLK_SPEEDBIAS;
            idx[nb] = (flag == VG_MARGIN_OLD) ? i - 1 : (i == K - 1 ? i - 1 : i);
            for (int k = 0; k < 9; ++k) x0[x0o + k] = x[7 * K + 9 * i + k];
            x0o += 9; ++nb;
        }
        if (cex >= 0) { kind[nb] = VG_BLK_EXPOSE; idx[nb] = 0; for (int k = 0; k < 7; ++k) x0[x0o + k] = ex[k]; x0o += 7; ++nb; }
        if (ctd >= 0) { kind[nb] = VG_BLK_TD; idx[nb] = 0; x0[x0o++] = ex[7]; ++nb; }
        mi[0] = 1; mi[1] = n; mi[2] = m; mi[3] = nb;
#ifdef BA_PROFILE
        mi[7] = (int)((clock64() - _tstart) >> 10);
#endif
    }
}

// Moves the priors a marginalization run produced (mout / miout of that run's layout) into the prior slots the next solve
// reads (BaPtrs::pri): x0, r0 and the n used rows of J0.  One workgroup per window; windows without a new prior keep their slot.
extern "C" __global__ __launch_bounds__(256) void ba_carry_prior_kernel(const double* __restrict__ mout, const int* __restrict__ miout,
        int mo_J0, int mo_r0, int mo_x0, int mo_stride, int mi_stride, int mcap, int x0cap, double* __restrict__ pri, int po_x0,
        int po_r0, int po_J0, int pld, int pstride) {
    const int w = blockIdx.x;
    const int* mi = miout + (size_t)w * mi_stride;
    if (!mi[0]) return;
    const int n = mi[1];
    const double* mo = mout + (size_t)w * mo_stride;
    double* p = pri + (size_t)w * pstride;
    for (int k = threadIdx.x; k < x0cap; k += 256) p[po_x0 + k] = mo[mo_x0 + k];
    for (int k = threadIdx.x; k < n; k += 256) p[po_r0 + k] = mo[mo_r0 + k];
    for (int k = threadIdx.x; k < n * n; k += 256) { const int r = k / n, cc = k % n; p[po_J0 + (size_t)r * pld + cc] = mo[mo_J0 + (size_t)r * mcap + cc]; }
}
extern "C" hipError_t ba_launch_carry_prior(int nwin, const double* mout, const int* miout, int mo_J0, int mo_r0, int mo_x0, int mo_stride,
                                            int mi_stride, int mcap, int x0cap, double* pri, int po_x0, int po_r0, int po_J0, int pld,
                                            int pstride, hipStream_t stream) {
    hipLaunchKernelGGL(ba_carry_prior_kernel, dim3(nwin), dim3(256), 0, stream, mout, miout, mo_J0, mo_r0, mo_x0, mo_stride, mi_stride,
                       mcap, x0cap, pri, po_x0, po_r0, po_J0, pld, pstride);
    return hipGetLastError();
}

extern "C" hipError_t ba_launch_marg(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, hipStream_t stream) {
    {   // per device, under a mutex (see set_lds_attrs in ba_pipeline.hip)
        static std::mutex mu;
        static unsigned long long done_mask = 0;
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        std::lock_guard<std::mutex> lock(mu);
        if (!(dev >= 0 && dev < 64 && ((done_mask >> dev) & 1ull))) {
            e = hipFuncSetAttribute((const void*)ba_marg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) done_mask |= 1ull << dev;
        }
    }
    hipLaunchKernelGGL(ba_marg_kernel, dim3(L.nwin), dim3(MG_NT), L.mg_lds_bytes, stream, dL, P);
    return hipGetLastError();
}
