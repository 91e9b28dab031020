// ba_seq.hip — windows that stay on the device from frame to frame (include/vinsgpu.h: vg_ba_seq_*).
//
// What the reference does on the host around Estimator::optimization(), once per frame (Estimator::processImage,
// vins_estimator/src/estimator.cpp:120-215), done here by four small kernels on the window's own tables in HBM:
//
//   ba_seq_add_kernel    FeatureManager::addFeatureCheckParallax (feature_manager.cpp:45-107): the new frame's observations join
//                        their tracks (linear search by feature id, first match, like the find_if of :58), unknown ids open new
//                        tracks at the end of the list in the order they arrive; last_track_num; the key-frame test by the
//                        mean parallax between the second and third newest frames (compensatedParallax2, :352-382) decides the
//                        marginalization flag (estimator.cpp:124-127).  Also takes in what processIMU produced: the state guess of
//                        the new frame and the pre-integration of the new (and, after a dropped non-keyframe, the merged) interval.
//   ba_seq_tri_kernel    FeatureManager::triangulate (feature_manager.cpp:202-257) for the tracks that are in the problem and have
//                        no depth yet (tri_dlt.h).
//   ba_seq_build_kernel  what optimization() walks f_manager.feature for (estimator.cpp:719-764) and vector2double() (:486-528):
//                        the landmark list (used_num >= 2 && start_frame < WINDOW_SIZE - 2, list order), inverse depths,
//                        observation rows, the factor list and its (anchor, target)-sorted slot table, the prior's block table --
//                        exactly the tables csrc/ba_host.hip pack_window builds on the host for an uploaded window.
//   [ the solve pipeline and the marginalization kernel of ba_pipeline.hip / ba_marg.hip, unchanged ]
//   ba_seq_slide_kernel  double2vector's setDepth (feature_manager.cpp:141-159), slideWindow() (estimator.cpp:1005-1126: state and
//                        pre-integration shift for both flags), FeatureManager::removeBackShiftDepth / removeFront
//                        (feature_manager.cpp:275-351), removeFailures (:161-171); the new prior's block table.
//
// One workgroup per window.  The track table is a plain ordered list (ints: id, start_frame, n_obs, solve_flag, landmark index in
// the current problem; doubles: estimated_depth and K rows of 8 per track); the slide writes the surviving tracks, compacted in
// order, into the window's second table, so no kernel moves rows in place.  Order matters: it is the order of para_Feature.
#include <hip/hip_runtime.h>
#include "ba_math.h"
#include "tri_dlt.h"
#include "ba_layout.h"
#include "vg_handle.h"
#include "../../include/vinsgpu.h"

#define SEQ_NT 256
#define SEQ_HDR 8                 // ints in front of a track table: [0] number of tracks
#define SEQ_FT_MAX 1024           // tracks per window (LDS tables of the kernels)
#define SEQ_NIN_MAX 1024          // observations per frame
#define SEQ_L_MAX 1024            // landmarks in the problem
#define SEQ_IMU 472               // doubles per pre-integration record (= BA_IMU_STRIDE)
#define SEQ_IN_POSE 0
#define SEQ_IN_SB 7
#define SEQ_IN_IMU_NEW 16
#define SEQ_IN_IMU_MERGED (16 + SEQ_IMU)
#define SEQ_IN_ROWS (16 + 2 * SEQ_IMU)

struct SeqTab {
    int* hdr; int* id; int* start; int* nobs; int* sflag; int* lm;
    double* depth; double* obs;
};
DEV SeqTab seq_table(const SeqDev& S, int which, int w) {
    SeqTab t;
    int* ti = S.ft_i[which] + (size_t)w * S.fi_stride;
    double* td = S.ft_d[which] + (size_t)w * S.fd_stride;
    t.hdr = ti; t.id = ti + SEQ_HDR; t.start = t.id + S.FT; t.nobs = t.start + S.FT; t.sflag = t.nobs + S.FT; t.lm = t.sflag + S.FT;
    t.depth = td; t.obs = td + S.FT;
    return t;
}

// Ordered compaction by the first wavefront: dst[f] = rank of f among the flagged entries (flag[f] != 0), -1 otherwise.
// Returns the count to the first wavefront's lanes (others must read it from LDS after a barrier).
DEV int seq_compact(const int* flag, int* dst, int n, int tid) {
    int base = 0;
    if (tid < 64) {
        for (int c = 0; c < n; c += 64) {
            const int f = c + tid;
            const int on = (f < n) && flag[f];
            const unsigned long long mask = __ballot(on);
            if (f < n) dst[f] = on ? base + __popcll(mask & ((1ull << tid) - 1ull)) : -1;
            base += __popcll(mask);
        }
    }
    return base;
}

// ---------------------------------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(SEQ_NT) void ba_seq_add_kernel(const BaLayout* __restrict__ Lp, BaPtrs P, SeqDev S, int cur) {
    __shared__ int s_id[SEQ_FT_MAX];
    __shared__ int s_match[SEQ_NIN_MAX];
    __shared__ int s_isnew[SEQ_NIN_MAX];
    __shared__ int s_cond[SEQ_FT_MAX];
    __shared__ double s_val[SEQ_FT_MAX];
    __shared__ int s_misc[4];
    const BaLayout& L = *Lp;
    const int w = blockIdx.x, tid = threadIdx.x, K = S.K, WS = K - 1;
    SeqTab T = seq_table(S, cur, w);
    int* ia = P.iarr + (size_t)w * L.istride;
    double* di = P.din + (size_t)w * L.dstride;
    const int* ii = S.in_i + (size_t)w * S.ii_stride;
    const double* idd = S.in_d + (size_t)w * S.id_stride;
    const int* inid = ii + 8;
    const double* rows = idd + SEQ_IN_ROWS;
    int* info = S.info + (size_t)w * VG_SEQ_INFO_INTS;
    const int n = T.hdr[0];
    int nin = ii[0];
    int status = VG_OK;
    if (nin > S.NIN) { nin = S.NIN; status = VG_ERR_UNSUPPORTED; }
    for (int f = tid; f < n; f += SEQ_NT) s_id[f] = T.id[f];
    __syncthreads();
    // ---- find_if(feature.begin(), feature.end(), id == feature_id): first match in list order
    for (int i = tid; i < nin; i += SEQ_NT) {
        const int id = inid[i];
        int m = -1;
        for (int j = 0; j < n; ++j) if (s_id[j] == id) { m = j; break; }
        s_match[i] = m;
        s_isnew[i] = m < 0;
    }
    __syncthreads();
    // ---- unknown ids: new tracks behind the list, in arrival order (feature.push_back, :62-65)
    {
        const int cnt = seq_compact(s_isnew, s_isnew, nin, tid);          // (in place: an entry is read and written by the same lane)
        if (tid == 0) s_misc[0] = cnt;
    }
    __syncthreads();
    const int n_new = s_misc[0];
    int ntot = n + n_new;
    if (ntot > S.FT) { ntot = S.FT; status = VG_ERR_UNSUPPORTED; }
    const double td_cur = di[L.do_td];
    for (int i = tid; i < nin; i += SEQ_NT) {
        int f = s_match[i], j = 0;
        if (f >= 0) {
            j = T.nobs[f];
            if (j >= K) continue;                                         // (cannot happen: a track spans consecutive frames of the window)
            T.nobs[f] = j + 1;
        } else {
            f = n + s_isnew[i];
            if (f >= S.FT) continue;
            T.id[f] = inid[i]; T.start[f] = WS; T.nobs[f] = 1; T.sflag[f] = 0; T.lm[f] = -1;
            T.depth[f] = -1.0;                                            // FeaturePerId ctor (feature_manager.h:57-61)
        }
        const double* r = rows + (size_t)i * 8;                           // [x y z u v vx vy]
        double* o = T.obs + ((size_t)f * K + j) * 8;                      // [x y u v vx vy cur_td z]
        o[0] = r[0]; o[1] = r[1]; o[2] = r[3]; o[3] = r[4]; o[4] = r[5]; o[5] = r[6]; o[6] = td_cur; o[7] = r[2];
    }
    __syncthreads();
    // ---- key-frame test (:73-106): parallax between frames WS - 2 and WS - 1 of every track that has both
    const int last_track_num = nin - n_new;
    const int fc = WS;
    for (int f = tid; f < ntot; f += SEQ_NT) {
        const int st = T.start[f], no = T.nobs[f];
        const int on = (st <= fc - 2) && (st + no - 1 >= fc - 1);
        double ans = 0.0;
        if (on) {
            const double* fi = T.obs + ((size_t)f * K + (fc - 2 - st)) * 8;
            const double* fj = T.obs + ((size_t)f * K + (fc - 1 - st)) * 8;
            const double u_j = fj[0], v_j = fj[1];
            const double dep_i = fi[7];
            const double u_i = fi[0] / dep_i, v_i = fi[1] / dep_i;
            const double du = u_i - u_j, dv = v_i - v_j;
            ans = fmax(0.0, sqrt(fmin(du * du + dv * dv, du * du + dv * dv)));       // (p_i_comp = p_i, :369)
        }
        s_cond[f] = on;
        s_val[f] = ans;
    }
    __syncthreads();
    if (tid == 0) {
        double sum = 0.0;
        int num = 0;
        for (int f = 0; f < ntot; ++f) if (s_cond[f]) { sum += s_val[f]; ++num; }       // list order, like the reference's loop
        int flag = VG_MARGIN_OLD;                                          // addFeatureCheckParallax() == true
        if (!(fc < 2 || last_track_num < 20) && num != 0) flag = (sum / num >= S.min_parallax) ? VG_MARGIN_OLD : VG_MARGIN_SECOND_NEW;
        ia[L.io_hdr + H_MARGIN] = flag;
        T.hdr[0] = ntot;
        info[VG_SEQ_FLAG] = flag; info[VG_SEQ_N_FEATURES] = ntot; info[VG_SEQ_N_TRACKED] = last_track_num; info[VG_SEQ_N_PARALLAX] = num;
        info[VG_SEQ_STATUS] = status;
    }
    // ---- what processIMU produced (estimator.cpp:83-117): state of the new frame, pre_integrations[WINDOW_SIZE]; after a dropped
    //      non-keyframe also the merged pre_integrations[WINDOW_SIZE - 1] (:1069-1085)
    if (tid < 7) di[L.do_pose + 7 * WS + tid] = idd[SEQ_IN_POSE + tid];
    if (tid >= 32 && tid < 41) di[L.do_sb + 9 * WS + (tid - 32)] = idd[SEQ_IN_SB + (tid - 32)];
    for (int e = tid; e < 467; e += SEQ_NT) di[L.do_imu + (size_t)(K - 2) * BA_IMU_STRIDE + e] = idd[SEQ_IN_IMU_NEW + e];
    if (tid == 64) ia[L.io_imu_valid + (K - 2)] = (ii[2] && idd[SEQ_IN_IMU_NEW] <= 10.0) ? 1 : 0;
    if (ii[1] && K >= 3) {
        for (int e = tid; e < 467; e += SEQ_NT) di[L.do_imu + (size_t)(K - 3) * BA_IMU_STRIDE + e] = idd[SEQ_IN_IMU_MERGED + e];
        if (tid == 65) ia[L.io_imu_valid + (K - 3)] = (ii[3] && idd[SEQ_IN_IMU_MERGED] <= 10.0) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// grid (nwin, FT / 64), one thread per track
extern "C" __global__ __launch_bounds__(64) void ba_seq_tri_kernel(const BaLayout* __restrict__ Lp, BaPtrs P, SeqDev S, int cur) {
    __shared__ double A[TRI_ROWS * 4][64];
    __shared__ double s_P[3 * TRI_MAXOBS], s_R[9 * TRI_MAXOBS];
    const BaLayout& L = *Lp;
    const int w = blockIdx.x, t = threadIdx.x, K = S.K, WS = K - 1;
    SeqTab T = seq_table(S, cur, w);
    const int n = T.hdr[0];
    if ((int)blockIdx.y * 64 >= n) return;
    const double* di = P.din + (size_t)w * L.dstride;
    if (t < K) {                                                          // Ps / Rs of the window (vector members of the Estimator)
        const double* x = di + L.do_pose + 7 * t;
        double q[4] = {x[3], x[4], x[5], x[6]};
        q_normalize(q);
        q_to_R(q, s_R + 9 * t);
        s_P[3 * t] = x[0]; s_P[3 * t + 1] = x[1]; s_P[3 * t + 2] = x[2];
    }
    __syncthreads();
    const int f = blockIdx.y * 64 + t;
    if (f >= n) return;
    const int st = T.start[f], no = T.nobs[f];
    if (!(no >= 2 && st < WS - 2)) return;                                // :206-208
    if (T.depth[f] > 0) return;                                           // :210-211
    double Ric[9], Tic[3];
    {
        const double* ex = di + L.do_ex;
        double q[4] = {ex[3], ex[4], ex[5], ex[6]};
        q_normalize(q);
        q_to_R(q, Ric);
        Tic[0] = ex[0]; Tic[1] = ex[1]; Tic[2] = ex[2];
    }
    const double* ob = T.obs + (size_t)f * K * 8;
    T.depth[f] = tri_dlt_depth(A, t, s_P, s_R, Ric, Tic, st, no,
                               [ob](int j, double* p) { p[0] = ob[8 * j]; p[1] = ob[8 * j + 1]; p[2] = ob[8 * j + 7]; }, S.init_depth);
}

// ---------------------------------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(SEQ_NT) void ba_seq_build_kernel(const BaLayout* __restrict__ Lp, BaPtrs P, SeqDev S, int cur) {
    __shared__ int s_in[SEQ_FT_MAX];           // track is in the problem
    __shared__ int s_lm[SEQ_FT_MAX];           // its landmark index
    __shared__ int s_ls[SEQ_L_MAX], s_ln[SEQ_L_MAX], s_feat[SEQ_L_MAX];
    __shared__ int s_fbeg[SEQ_L_MAX + 1], s_off[SEQ_L_MAX + 1];
    __shared__ int s_pair[BA_MAX_K * BA_MAX_K + 1];
    __shared__ int s_misc[4];
    const BaLayout& L = *Lp;
    const int w = blockIdx.x, tid = threadIdx.x, K = S.K, WS = K - 1, Kp = L.Kp;
    SeqTab T = seq_table(S, cur, w);
    int* ia = P.iarr + (size_t)w * L.istride;
    double* di = P.din + (size_t)w * L.dstride;
    int* info = S.info + (size_t)w * VG_SEQ_INFO_INTS;
    const int* sp = S.sp + (size_t)w * S.sp_stride;
    const int n = T.hdr[0];
    const int margin = ia[L.io_hdr + H_MARGIN];
    __syncthreads();
    // ---- clear the tables a host pack would have zero-filled (the IMU validity flags and the owner task list stay)
    for (int k = tid; k < BA_HDR_INTS; k += SEQ_NT) ia[L.io_hdr + k] = 0;
    for (int k = L.io_lm_start + tid; k < L.io_task_list; k += SEQ_NT) ia[k] = 0;        // lm_start .. pair_ptr
    for (int k = L.io_pb_kind + tid; k < L.istride; k += SEQ_NT) ia[k] = 0;
    // ---- the landmark list: used_num >= 2 && start_frame < WINDOW_SIZE - 2, in list order (estimator.cpp:722-725)
    for (int f = tid; f < n; f += SEQ_NT) s_in[f] = (T.nobs[f] >= 2 && T.start[f] < WS - 2) ? 1 : 0;
    __syncthreads();
    {
        const int cnt = seq_compact(s_in, s_lm, n, tid);
        if (tid == 0) s_misc[0] = cnt;
    }
    __syncthreads();
    int nL = s_misc[0];
    int status = info[VG_SEQ_STATUS];
    if (nL > L.Lcap || nL > SEQ_L_MAX) { nL = 0; status = VG_ERR_UNSUPPORTED; }
    for (int f = tid; f < n; f += SEQ_NT) {
        const int l = nL ? s_lm[f] : -1;
        T.lm[f] = l;
        if (l >= 0) { s_ls[l] = T.start[f]; s_ln[l] = T.nobs[f]; s_feat[l] = f; }
    }
    __syncthreads();
    if (tid == 0) {
        int fb = 0, ob = 0;
        for (int l = 0; l < nL; ++l) { s_fbeg[l] = fb; s_off[l] = ob; fb += s_ln[l] - 1; ob += s_ln[l]; }
        s_fbeg[nL] = fb; s_off[nL] = ob;
        s_misc[1] = fb; s_misc[2] = ob;
    }
    __syncthreads();
    int F = s_misc[1], nobs_tot = s_misc[2];
    if (F > L.Fcap || nobs_tot > L.Ocap) { nL = 0; F = 0; nobs_tot = 0; status = VG_ERR_UNSUPPORTED; }
    // ---- per landmark: start frame, first factor, inverse depth (vector2double: 1 / estimated_depth, :519-522), observation rows,
    //      factors (first observation -> every later one, :732-763)
    for (int l = tid; l < nL; l += SEQ_NT) {
        const int f = s_feat[l], s = s_ls[l], no = s_ln[l], o = s_off[l], fb = s_fbeg[l];
        ia[L.io_lm_start + l] = s;
        ia[L.io_lm_fbeg + l] = fb;
        di[L.do_lam + l] = 1.0 / T.depth[f];
        for (int k = 1; k < no; ++k) {
            const int q = fb + k - 1;
            ia[L.io_fac_i + q] = s; ia[L.io_fac_j + q] = s + k; ia[L.io_fac_lm + q] = l;
            ia[L.io_fac_oi + q] = o; ia[L.io_fac_oj + q] = o + k;
        }
    }
    if (tid == 0) ia[L.io_lm_fbeg + nL] = F;
    for (int l = nL + tid; l < L.Lcap; l += SEQ_NT) di[L.do_lam + l] = 0.0;
    for (int e = tid; e < nobs_tot * 8; e += SEQ_NT) {                    // rows [x y u v vx vy cur_td | 0]
        const int row = e >> 3, c = e & 7;
        // landmark of this row: binary search in the offsets
        int lo = 0, hi = nL - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[mid] <= row) lo = mid; else hi = mid - 1; }
        const int f = s_feat[lo], j = row - s_off[lo];
        di[L.do_obs + (size_t)row * BA_OBS_STRIDE + c] = (c < 7) ? T.obs[((size_t)f * K + j) * 8 + c] : 0.0;
    }
    for (int e = nobs_tot * 8 + tid; e < L.Ocap * BA_OBS_STRIDE; e += SEQ_NT) di[L.do_obs + e] = 0.0;
    // ---- slot table: records sorted by (anchor, target) pair, landmark order inside a pair (pack_window)
    for (int p = tid; p < Kp * Kp; p += SEQ_NT) {
        const int i = p / Kp, d = p % Kp - i;
        int c = 0;
        if (d > 0) for (int l = 0; l < nL; ++l) c += (s_ls[l] == i && s_ln[l] > d) ? 1 : 0;
        s_pair[p] = c;
    }
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int p = 0; p < Kp * Kp; ++p) { const int c = s_pair[p]; s_pair[p] = acc; acc += c; }
        s_pair[Kp * Kp] = acc;
    }
    __syncthreads();
    for (int p = tid; p <= Kp * Kp; p += SEQ_NT) ia[L.io_pair_ptr + p] = s_pair[p];
    for (int p = tid; p < Kp * Kp; p += SEQ_NT) {
        const int i = p / Kp, d = p % Kp - i;
        if (d <= 0) continue;
        int cursor = s_pair[p];
        for (int l = 0; l < nL; ++l) if (s_ls[l] == i && s_ln[l] > d) ia[L.io_fac_slot + s_fbeg[l] + d - 1] = cursor++;
    }
    if (tid == 0) {
        // wavefront owner tasks (pack_window): every block except the diagonal pose blocks and the blocks among ex / td
        const int nb = Kp + L.e + L.t;
        int q = 0;
        for (int br = 0; br < nb; ++br)
            for (int bc = 0; bc <= br; ++bc)
                if (!(br == bc && br < Kp) && !(bc >= Kp)) ia[L.io_task_list + q++] = br * (br + 1) / 2 + bc;
        // the prior's block table: where the rows / columns of J0 sit in the solver's ordering
        const int pn = sp[0], pnb = pn ? sp[1] : 0;
        int off = 0, x0off = 0;
        for (int b = 0; b < pnb; ++b) {
            const int kind = sp[2 + b], idx = sp[2 + (K + 4) + b];
            ia[L.io_pb_kind + b] = kind; ia[L.io_pb_idx + b] = idx; ia[L.io_pb_off + b] = off; ia[L.io_pb_x0off + b] = x0off;
            int col = -1;
            if (kind == VG_BLK_POSE) col = 6 * idx;
            else if (kind == VG_BLK_SPEEDBIAS) col = L.Rc + 9 * idx;
            else if (kind == VG_BLK_EXPOSE) col = L.e ? 6 * Kp : -1;
            else col = L.t ? 6 * Kp + 6 * L.e : -1;
            ia[L.io_pb_col + b] = col;
            off += kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 6);
            x0off += kind == VG_BLK_SPEEDBIAS ? 9 : (kind == VG_BLK_TD ? 1 : 7);
        }
        int* hdr = ia + L.io_hdr;
        hdr[H_L] = nL; hdr[H_F] = F; hdr[H_NPRIOR] = pn; hdr[H_NBLK] = pnb; hdr[H_MAXIT] = S.max_iters; hdr[H_NCHUNK] = 1;
        hdr[H_MARGIN] = margin; hdr[H_STATUS] = 0; hdr[H_MARGMODE] = S.marg_mode;
        info[VG_SEQ_N_LANDMARKS] = nL; info[VG_SEQ_N_FACTORS] = F; info[VG_SEQ_STATUS] = status;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
extern "C" __global__ __launch_bounds__(SEQ_NT) void ba_seq_slide_kernel(const BaLayout* __restrict__ Lp, BaPtrs P, SeqDev S, int cur) {
    __shared__ int s_alive[SEQ_FT_MAX], s_dst[SEQ_FT_MAX], s_ns[SEQ_FT_MAX], s_nn[SEQ_FT_MAX], s_src[SEQ_FT_MAX], s_skip[SEQ_FT_MAX];
    __shared__ int s_fl[SEQ_FT_MAX];
    __shared__ double s_dep[SEQ_FT_MAX];
    __shared__ double s_tf[24];                // R0 (9), P0 (3), R1 (9), P1 (3) of removeBackShiftDepth
    __shared__ int s_misc[4];
    const BaLayout& L = *Lp;
    const int w = blockIdx.x, tid = threadIdx.x, K = S.K, WS = K - 1;
    SeqTab T = seq_table(S, cur, w), N = seq_table(S, cur ^ 1, w);
    int* ia = P.iarr + (size_t)w * L.istride;
    double* di = P.din + (size_t)w * L.dstride;
    const double* out = P.out + (size_t)w * L.ostride;
    int* info = S.info + (size_t)w * VG_SEQ_INFO_INTS;
    const int n = T.hdr[0];
    const int flag = ia[L.io_hdr + H_MARGIN];
    if (tid == 0) {
        // slideWindowOld (estimator.cpp:1115-1126): R0 = back_R0 ric, P0 = back_P0 + back_R0 tic; R1 = Rs[0] ric, P1 = Ps[0] + Rs[0] tic
        // with back_* = frame 0 and Rs[0] / Ps[0] = frame 1 of the window just solved, ric / tic as solved
        double qe[4] = {out[L.oo_ex + 3], out[L.oo_ex + 4], out[L.oo_ex + 5], out[L.oo_ex + 6]}, Ric[9];
        q_normalize(qe);
        q_to_R(qe, Ric);
        const double Tic[3] = {out[L.oo_ex], out[L.oo_ex + 1], out[L.oo_ex + 2]};
        for (int k = 0; k < 2; ++k) {
            const double* x = out + L.oo_pose + 7 * k;
            double q[4] = {x[3], x[4], x[5], x[6]}, R[9], t[3];
            q_normalize(q);
            q_to_R(q, R);
            m3_mul(R, Ric, s_tf + 12 * k);
            m3_vec(R, Tic, t);
            for (int c = 0; c < 3; ++c) s_tf[12 * k + 9 + c] = x[c] + t[c];
        }
    }
    __syncthreads();
    // ---- per track: setDepth, then removeBackShiftDepth / removeFront, then removeFailures
    for (int f = tid; f < n; f += SEQ_NT) {
        const int st = T.start[f], no = T.nobs[f], l = T.lm[f];
        double dep = T.depth[f];
        int fl = T.sflag[f];
        if (l >= 0) {                                                     // FeatureManager::setDepth (feature_manager.cpp:141-159)
            dep = 1.0 / out[L.oo_lam + l];
            fl = dep < 0 ? 2 : 1;
        }
        int alive = 1, ns = st, nn = no, src = 0, skip = -1;
        if (flag == VG_MARGIN_OLD) {
            if (st != 0) ns = st - 1;
            else {
                nn = no - 1; src = 1;
                if (nn < 2) alive = 0;
                else {
                    const double* r = T.obs + (size_t)f * K * 8;
                    const double pi[3] = {r[0] * dep, r[1] * dep, r[7] * dep};           // uv_i * estimated_depth
                    double wp[3], d[3], pj[3];
                    m3_vec(s_tf, pi, wp);
                    for (int c = 0; c < 3; ++c) d[c] = wp[c] + s_tf[9 + c] - s_tf[21 + c];   // marg_R pts_i + marg_P - new_P
                    m3t_vec(s_tf + 12, d, pj);
                    dep = pj[2] > 0 ? pj[2] : S.init_depth;
                }
            }
        } else {                                                          // removeFront(frame_count = WINDOW_SIZE)
            if (st == WS) ns = st - 1;
            else {
                const int j = WS - 1 - st;
                if (no - 1 >= j) { skip = j; nn = no - 1; if (nn == 0) alive = 0; }
            }
        }
        if (fl == 2) alive = 0;                                           // removeFailures
        s_alive[f] = alive; s_ns[f] = ns; s_nn[f] = nn; s_src[f] = src; s_skip[f] = skip; s_dep[f] = dep; s_fl[f] = fl;
    }
    __syncthreads();
    {
        const int cnt = seq_compact(s_alive, s_dst, n, tid);
        if (tid == 0) { s_misc[0] = cnt; N.hdr[0] = cnt; info[VG_SEQ_N_AFTER] = cnt; }
    }
    __syncthreads();
    for (int f = tid; f < n; f += SEQ_NT) {
        const int d = s_dst[f];
        if (d < 0) continue;
        N.id[d] = T.id[f]; N.start[d] = s_ns[f]; N.nobs[d] = s_nn[f]; N.sflag[d] = s_fl[f]; N.lm[d] = -1; N.depth[d] = s_dep[f];
    }
    for (int e = tid; e < n * K * 8; e += SEQ_NT) {
        const int f = e / (K * 8), r = e - f * (K * 8), jj = r >> 3, c = r & 7;
        const int d = s_dst[f];
        if (d < 0 || jj >= s_nn[f]) continue;
        int sj = s_src[f] + jj;
        if (s_skip[f] >= 0 && sj >= s_skip[f]) ++sj;
        N.obs[((size_t)d * K + jj) * 8 + c] = T.obs[((size_t)f * K + sj) * 8 + c];
    }
    // ---- states and pre-integrations (estimator.cpp:1010-1050 / :1069-1099); the newest slot keeps a copy of the newest frame
    for (int e = tid; e < 7 * K; e += SEQ_NT) {
        const int k = e / 7, c = e - 7 * k;
        int from;
        if (flag == VG_MARGIN_OLD) from = k < K - 1 ? k + 1 : K - 1;
        else from = k <= K - 3 ? k : K - 1;
        di[L.do_pose + e] = out[L.oo_pose + 7 * from + c];
    }
    for (int e = tid; e < 9 * K; e += SEQ_NT) {
        const int k = e / 9, c = e - 9 * k;
        int from;
        if (flag == VG_MARGIN_OLD) from = k < K - 1 ? k + 1 : K - 1;
        else from = k <= K - 3 ? k : K - 1;
        di[L.do_sb + e] = out[L.oo_sb + 9 * from + c];
    }
    if (tid < 7) di[L.do_ex + tid] = out[L.oo_ex + tid];
    if (tid == 7) di[L.do_td] = out[L.oo_td];
    if (flag == VG_MARGIN_OLD) {
        for (int e = tid; e < 467; e += SEQ_NT)
            for (int k = 0; k + 1 < K - 1; ++k)
                di[L.do_imu + (size_t)k * BA_IMU_STRIDE + e] = di[L.do_imu + (size_t)(k + 1) * BA_IMU_STRIDE + e];
        if (tid == 0) for (int k = 0; k + 1 < K - 1; ++k) ia[L.io_imu_valid + k] = ia[L.io_imu_valid + k + 1];
    }
    // ---- the new prior's block table (the factor itself is moved by ba_carry_prior_kernel); none produced: the old one stays
    const int* mi = P.miout + (size_t)w * L.mi_stride;
    int* sp = S.sp + (size_t)w * S.sp_stride;
    if (mi[0]) {
        if (tid == 0) { sp[0] = mi[1]; sp[1] = mi[3]; }
        for (int b = tid; b < K + 4; b += SEQ_NT) { sp[2 + b] = mi[8 + b]; sp[2 + (K + 4) + b] = mi[8 + (K + 4) + b]; }
    } else if (flag == VG_MARGIN_OLD) {
        // A MARGIN_OLD step whose marginalization produced nothing (a non-finite solve): the frames and tracks have been shifted
        // above, the old prior still names the un-shifted blocks -- it must not survive (the drop-in drops last_marginalization_info
        // in the same situation, host/dropin/estimator_optimization.cpp).  The window goes on without a prior and the
        // step reports VG_ERR_NUMERIC in info[VG_SEQ_STATUS], so that the caller can re-seed it (vg_ba_seq_import).
        // (MARGIN_SECOND_NEW without Pose[WINDOW_SIZE - 1] in the prior legitimately produces nothing: the old prior stays, its
        //  blocks keep their slots, estimator.cpp:933-936.)
        if (tid == 0) {
            sp[0] = 0; sp[1] = 0;
            int* info = S.info + (size_t)w * VG_SEQ_INFO_INTS;
            if (info[VG_SEQ_STATUS] == VG_OK) info[VG_SEQ_STATUS] = VG_ERR_NUMERIC;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
extern "C" hipError_t ba_seq_launch_front(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, const SeqDev& S, int cur, hipStream_t stream) {
    hipLaunchKernelGGL(ba_seq_add_kernel, dim3(L.nwin), dim3(SEQ_NT), 0, stream, dL, P, S, cur);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ba_seq_tri_kernel, dim3(L.nwin, (S.FT + 63) / 64), dim3(64), 0, stream, dL, P, S, cur);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(ba_seq_build_kernel, dim3(L.nwin), dim3(SEQ_NT), 0, stream, dL, P, S, cur);
    return hipGetLastError();
}
extern "C" hipError_t ba_seq_launch_slide(const BaLayout& L, const BaLayout* dL, const BaPtrs& P, const SeqDev& S, int cur, hipStream_t stream) {
    hipLaunchKernelGGL(ba_seq_slide_kernel, dim3(L.nwin), dim3(SEQ_NT), 0, stream, dL, P, S, cur);
    return hipGetLastError();
}
extern "C" int ba_seq_limits(int* ft_max, int* nin_max, int* hdr_ints, int* in_rows_off) {
    if (ft_max) *ft_max = SEQ_FT_MAX;
    if (nin_max) *nin_max = SEQ_NIN_MAX;
    if (hdr_ints) *hdr_ints = SEQ_HDR;
    if (in_rows_off) *in_rows_off = SEQ_IN_ROWS;
    return SEQ_L_MAX;
}
