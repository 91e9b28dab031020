// sharded_estimator.cpp — see sharded_estimator.h
#include "sharded_estimator.h"
#include <algorithm>
#include <cstring>

namespace vins_gpu {

std::vector<std::pair<int, int>> landmark_shards(const int* lm_nobs, int L, int world) {
    // cumulative Schur work; cut r at the first landmark whose cumulative work reaches total * r / world
    std::vector<double> cum((size_t)L + 1, 0.0);
    for (int l = 0; l < L; ++l) { const double w = 6.0 * lm_nobs[l]; cum[l + 1] = cum[l] + w * w; }
    const double total = cum[L];
    std::vector<int> cuts(1, 0);
    for (int r = 1; r < world; ++r) {
        const double want = total * r / world;
        int c = (int)(std::lower_bound(cum.begin(), cum.end(), want) - cum.begin());      // first index with cum >= want
        c = std::min(std::max(c, cuts.back()), L);
        cuts.push_back(c);
    }
    cuts.push_back(L);
    std::vector<std::pair<int, int>> out;
    for (int r = 0; r < world; ++r) out.emplace_back(cuts[r], cuts[r + 1]);
    return out;
}

int shard_problem(const vg_ba_problem& full, int rank, int world, ProblemShard& out) {
    if (rank < 0 || world < 1 || rank >= world) return VG_ERR_BAD_ARG;
    const auto rg = landmark_shards(full.lm_nobs, full.L, world)[rank];
    const int lo = rg.first, hi = rg.second;
    for (int l = lo + 1; l < hi; ++l)
        if (full.lm_obs_off[l] != full.lm_obs_off[l - 1] + full.lm_nobs[l - 1]) return VG_ERR_BAD_ARG;
    out.lo = lo; out.hi = hi;
    out.pb = full;                                   // frames, IMU factors, prior, options: replicated
    const int o_lo = lo < hi ? full.lm_obs_off[lo] : 0;
    const int o_hi = lo < hi ? full.lm_obs_off[hi - 1] + full.lm_nobs[hi - 1] : 0;
    out.lm_obs_off.resize((size_t)std::max(hi - lo, 1));
    for (int l = lo; l < hi; ++l) out.lm_obs_off[l - lo] = full.lm_obs_off[l] - o_lo;
    out.pb.L = hi - lo;
    out.pb.n_obs = o_hi - o_lo;
    out.pb.inv_depth = full.inv_depth + lo;
    out.pb.lm_start = full.lm_start + lo;
    out.pb.lm_nobs = full.lm_nobs + lo;
    out.pb.lm_obs_off = out.lm_obs_off.data();
    out.pb.obs = full.obs + (size_t)7 * o_lo;
    // relocalisation matches of this rank's landmarks (estimator.cpp:769-801), indices re-based
    out.relo_lm.clear(); out.relo_xy.clear();
    for (int k = 0; k < full.relo_n; ++k)
        if (full.relo_lm[k] >= lo && full.relo_lm[k] < hi) {
            out.relo_lm.push_back(full.relo_lm[k] - lo);
            out.relo_xy.push_back(full.relo_xy[2 * k]); out.relo_xy.push_back(full.relo_xy[2 * k + 1]);
        }
    out.pb.relo_n = (int)out.relo_lm.size();
    out.pb.relo_lm = out.relo_lm.data();
    out.pb.relo_xy = out.relo_xy.data();
    return VG_OK;
}

ShardedWindow::ShardedWindow(const ShardTransport& t, const vg_config* cfg) : t_(t) {
    if (vg_abi_version() != VG_ABI_VERSION) { err_ = "libvinsgpu.so was built from another include/vinsgpu.h"; return; }
    const int rc1 = cfg ? vg_create_config(cfg, &solve_) : vg_create(&solve_);
    const int rc2 = rc1 == VG_OK ? (cfg ? vg_create_config(cfg, &marg_) : vg_create(&marg_)) : rc1;
    if (rc1 != VG_OK || rc2 != VG_OK) {
        err_ = "vg_create failed: no MI355X / libvinsgpu (no CPU fallback)";
        if (solve_) vg_destroy(solve_);
        if (marg_) vg_destroy(marg_);
        solve_ = marg_ = nullptr;
        return;
    }
    // landmark shards always take the large-window path (the landmark Schur complement goes through the reduce buffers)
    vg_ba_set_large_window(solve_, 1);
    vg_ba_set_large_window(marg_, 1);
}

ShardedWindow::~ShardedWindow() {
    if (solve_) vg_destroy(solve_);
    if (marg_) vg_destroy(marg_);
}

// every rank's frame-0 tracks (estimator.cpp:853-888: the projection factors MarginalizationInfo gets are those of the features with
// start_frame == 0) with their solved inverse depths, in rank = landmark order.  Wire format per rank: [count | nobs[count] |
// inv_depth[count] | rows], as doubles, padded to the longest contribution (the transport gathers fixed-size pieces).
int ShardedWindow::gather_frame0(const vg_ba_state& st, std::vector<int>& nobs, std::vector<double>& inv_depth, std::vector<double>& obs) {
    const vg_ba_problem& p = shard_.pb;
    std::vector<double> mine(1, 0.0);
    std::vector<double> rows;
    int cnt = 0;
    for (int l = 0; l < p.L; ++l) if (p.lm_start[l] == 0) ++cnt;
    mine[0] = cnt;
    for (int l = 0; l < p.L; ++l) if (p.lm_start[l] == 0) mine.push_back(p.lm_nobs[l]);
    for (int l = 0; l < p.L; ++l) if (p.lm_start[l] == 0) mine.push_back(st.inv_depth[l]);
    for (int l = 0; l < p.L; ++l)
        if (p.lm_start[l] == 0) mine.insert(mine.end(), p.obs + (size_t)7 * p.lm_obs_off[l], p.obs + (size_t)7 * (p.lm_obs_off[l] + p.lm_nobs[l]));
    std::vector<double> all;
    if (t_.world > 1) {
        if (!t_.all_gather) { err_ = "ShardTransport::all_gather is null with world > 1"; return VG_ERR_BAD_ARG; }
        double len = (double)mine.size();
        std::vector<double> lens((size_t)t_.world, 0.0);
        if (t_.all_gather(t_.user, &len, sizeof(double), lens.data())) { err_ = "all_gather (lengths) failed"; return VG_ERR_HIP; }
        size_t longest = 0;
        for (double v : lens) longest = std::max(longest, (size_t)v);
        mine.resize(longest, 0.0);
        all.resize(longest * t_.world);
        if (t_.all_gather(t_.user, mine.data(), longest * sizeof(double), all.data())) { err_ = "all_gather (frame-0 tracks) failed"; return VG_ERR_HIP; }
        nobs.clear(); inv_depth.clear(); obs.clear();
        for (int r = 0; r < t_.world; ++r) {
            const double* q = all.data() + (size_t)r * longest;
            const int c = (int)q[0];
            int nrow = 0;
            for (int k = 0; k < c; ++k) { nobs.push_back((int)q[1 + k]); nrow += (int)q[1 + k]; }
            inv_depth.insert(inv_depth.end(), q + 1 + c, q + 1 + 2 * c);
            obs.insert(obs.end(), q + 1 + 2 * c, q + 1 + 2 * c + (size_t)7 * nrow);
        }
    } else {
        nobs.clear();
        int nrow = 0;
        for (int k = 0; k < cnt; ++k) { nobs.push_back((int)mine[1 + k]); nrow += (int)mine[1 + k]; }
        inv_depth.assign(mine.begin() + 1 + cnt, mine.begin() + 1 + 2 * cnt);
        obs.assign(mine.begin() + 1 + 2 * cnt, mine.begin() + 1 + 2 * cnt + (size_t)7 * nrow);
    }
    return VG_OK;
}

// The checks of optimize() that depend on the FULL problem only (identical on every rank).
int ShardedWindow::validate_full(const vg_ba_problem& full) {
    if (full.L < 0 || (full.L > 0 && (!full.lm_nobs || !full.lm_obs_off))) { err_ = "bad landmark tables"; return VG_ERR_BAD_ARG; }
    for (int l = 1; l < full.L; ++l)
        if (full.lm_obs_off[l] != full.lm_obs_off[l - 1] + full.lm_nobs[l - 1]) {
            err_ = "observation rows of consecutive landmarks must be consecutive";
            return VG_ERR_BAD_ARG;
        }
    // The relocalisation pose (estimator.cpp:769-801) is a block of the reduced camera system only on the ranks that hold one of its
    // matched landmarks: the ranks would build reduced systems of different sizes and sum them.  Not offered on more than one rank.
    if (full.relo_n > 0 && t_.world > 1) {
        err_ = "relocalisation factors are not offered in a window sharded over several ranks";
        return VG_ERR_UNSUPPORTED;
    }
    // A resident prior lives in the slot of the handle that made it; the marginalization handle's slot holds the prior the PREVIOUS
    // sharded frame produced, the solve handle's slot nothing at all: the caller passes the prior by value.
    if (full.prior_n == VG_PRIOR_RESIDENT) {
        err_ = "VG_PRIOR_RESIDENT is not offered by ShardedWindow::optimize: pass the prior by value";
        return VG_ERR_UNSUPPORTED;
    }
    return VG_OK;
}

int ShardedWindow::optimize(const vg_ba_problem& full, int margin_flag, vg_ba_state* st, vg_ba_summary* sm, vg_ba_prior* new_prior) {
    if (!ok()) return VG_ERR_NO_DEVICE;
    if (!st || !sm || (margin_flag != VG_MARGIN_NONE && !new_prior)) return VG_ERR_BAD_ARG;
    // Everything that can be refused is refused HERE, from the full problem, so that every rank takes the same decision before the
    // first collective (a rank that returned early would leave the others blocked in the all-reduce / all-gather).
    int rc = validate_full(full);
    if (rc != VG_OK) return rc;
    rc = shard_problem(full, t_.rank, t_.world, shard_);
    if (rc != VG_OK) { err_ = "observation rows of consecutive landmarks must be consecutive"; return rc; }
    // ---- the solve: this rank's landmarks, the reduced camera system summed over the ranks by the hook of the solve handle
    rc = vg_ba_optimize(solve_, &shard_.pb, VG_MARGIN_NONE, st, sm, nullptr);
    if (rc != VG_OK) { err_ = std::string("sharded solve: ") + vg_last_error(solve_); return rc; }
    if (new_prior) { new_prior->valid = 0; new_prior->n = 0; new_prior->m = 0; new_prior->nblocks = 0; }
    if (margin_flag == VG_MARGIN_NONE || sm->status != VG_OK) return VG_OK;
    // ---- the marginalization (estimator.cpp:825-1000): the same small single-rank problem on every rank -- all frames at their
    //      solved states, the IMU factors, the old prior, the frame-0 tracks of ALL ranks in landmark order, max_iters = 0 (the
    //      factors are evaluated where the solve ended; with nothing to iterate the gauge fix is the identity).  MARGIN_SECOND_NEW
    //      involves the prior alone.
    std::vector<int> nobs;
    std::vector<double> lam, rows;
    if (margin_flag == VG_MARGIN_OLD) {
        rc = gather_frame0(*st, nobs, lam, rows);
        if (rc != VG_OK) return rc;
    }
    const int L0 = (int)nobs.size();
    std::vector<int> start((size_t)std::max(L0, 1), 0), off((size_t)std::max(L0, 1), 0);
    for (int l = 1; l < L0; ++l) off[l] = off[l - 1] + nobs[l - 1];
    if (nobs.empty()) { nobs.push_back(0); lam.push_back(1.0); rows.assign(7, 0.0); }
    vg_ba_problem red = full;
    red.pose = st->pose; red.speedbias = st->speedbias; red.ex_pose = st->ex_pose; red.td = *st->td;
    red.L = L0; red.n_obs = L0 ? off[L0 - 1] + nobs[L0 - 1] : 0;
    red.inv_depth = lam.data(); red.lm_start = start.data(); red.lm_nobs = nobs.data(); red.lm_obs_off = off.data(); red.obs = rows.data();
    red.relo_n = 0; red.relo_pose = nullptr; red.relo_lm = nullptr; red.relo_xy = nullptr;
    red.max_iters = 0;
    std::vector<double> pose((size_t)7 * full.K), sb((size_t)9 * full.K), lam_out((size_t)std::max(L0, 1));
    double ex[7], td = 0.0;
    vg_ba_state tmp;
    tmp.pose = pose.data(); tmp.speedbias = sb.data(); tmp.ex_pose = ex; tmp.td = &td; tmp.inv_depth = lam_out.data(); tmp.relo_pose = nullptr;
    vg_ba_summary sm2;
    rc = vg_ba_optimize(marg_, &red, margin_flag, &tmp, &sm2, new_prior);
    if (rc != VG_OK) { err_ = std::string("marginalization of the sharded window: ") + vg_last_error(marg_); return rc; }
    if (sm2.status != VG_OK) { err_ = "marginalization of the sharded window reported a numeric failure"; return sm2.status; }
    return VG_OK;
}

}  // namespace vins_gpu

// ---- C entry points (tests drive the class through ctypes; an application links the class itself)
extern "C" {
typedef int (*vins_sharded_gather_fn)(void* user, const void* in, size_t bytes, void* out);
void* vins_sharded_create(int rank, int world, vins_sharded_gather_fn all_gather, void* user) {
    vins_gpu::ShardTransport t;
    t.rank = rank; t.world = world; t.all_gather = all_gather; t.user = user;
    auto* w = new vins_gpu::ShardedWindow(t);
    if (!w->ok()) { delete w; return nullptr; }
    return w;
}
void vins_sharded_destroy(void* w) { delete static_cast<vins_gpu::ShardedWindow*>(w); }
vg_handle* vins_sharded_solve_handle(void* w) { return static_cast<vins_gpu::ShardedWindow*>(w)->solve_handle(); }
vg_handle* vins_sharded_marg_handle(void* w) { return static_cast<vins_gpu::ShardedWindow*>(w)->marg_handle(); }
int vins_sharded_optimize(void* w, const vg_ba_problem* full, int margin_flag, vg_ba_state* st, vg_ba_summary* sm, vg_ba_prior* prior) {
    return static_cast<vins_gpu::ShardedWindow*>(w)->optimize(*full, margin_flag, st, sm, prior);
}
void vins_sharded_range(void* w, int* lo, int* hi) {
    const auto& s = static_cast<vins_gpu::ShardedWindow*>(w)->shard();
    *lo = s.lo; *hi = s.hi;
}
const char* vins_sharded_last_error(void* w) { return static_cast<vins_gpu::ShardedWindow*>(w)->last_error().c_str(); }
int vins_sharded_landmark_shards(const int* lm_nobs, int L, int world, int* lo_hi /* 2 * world */) {
    const auto v = vins_gpu::landmark_shards(lm_nobs, L, world);
    for (int r = 0; r < world; ++r) { lo_hi[2 * r] = v[r].first; lo_hi[2 * r + 1] = v[r].second; }
    return 0;
}
}
