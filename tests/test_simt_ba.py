"""The BA kernel sources executed by the CPU fiber emulator (tests/simt) against the NumPy oracle — `not gpu` tests, so
that kernel logic (barriers, indexing, the trust-region state machine spread over launches) is checked in this GPU-less
container on every run.  The emulated library is test infrastructure; the product never loads it (tests/simt/README.md).
Sizes are small: a fiber switch per barrier makes a full-size window take ~10 s."""
import numpy as np
import pytest

from oracle import ba_numpy as B
from vins_mono_amd import ba, synth

import ba_fixtures as FX
from test_ba_gpu import _check_solve, _check_prior, resident_prior_chain, marginalize_many_frame0_landmarks, launch_modes_agree


def test_emulated_solve_matches_oracle(simt_handle):
    prob = synth.SyntheticSequence(3, L=30).window(0)
    _check_solve(simt_handle, prob)


def test_emulated_rejected_and_interpolated_steps(simt_handle):
    build, need = FX.BRANCH_FIXTURES['low_parallax']
    prob = build(L=24)
    prob['max_iters'] = 10
    _, _, summ = _check_solve(simt_handle, prob, rtol_cost=1e-4)
    assert {'rejected', 'gn'} <= FX.trace_features(summ)


def test_emulated_failed_factorisations_escalate_mu_then_fail(simt_handle):
    """The blocked Cholesky reports a bad pivot through the 1/sqrt values of a 16-column block (NaN / Inf / <= 0), the 9x9
    chain factors likewise: a window whose every linear solve fails must walk mu x 10, five invalid steps and FAILURE with the
    same per-iteration trace as the oracle."""
    build, need = FX.BRANCH_FIXTURES['overflowing_landmark']
    with np.errstate(all='ignore'):
        _, _, summ = _check_solve(simt_handle, build(L=24))
    assert need <= FX.trace_features(summ)


def test_emulated_marginalization_matches_oracle(simt_handle):
    seq = synth.SyntheticSequence(40, L=30)
    prob = seq.window(0)
    x, _ = B.solve(prob)
    at = dict(prob)
    at.update(pose=x['pose'], sb=x['sb'], ex=x['ex'], td=x['td'], inv_depth=x['inv_depth'], max_iters=0)
    _, _, pr_o = B.optimization(at, B.MARGIN_OLD)
    for mode in (ba.VG_MARG_SQRT, ba.VG_MARG_EIGEN):          # pivoted-Cholesky square root (default) / the reference's eigen form
        simt_handle.ba_set_marg_mode(mode)
        try:
            st_g, sm_g, pr_g = simt_handle.ba_optimize(at, ba.VG_MARGIN_OLD)
        finally:
            simt_handle.ba_set_marg_mode(ba.VG_MARG_SQRT)
        assert sm_g['status'] == 0
        _check_prior(pr_g, pr_o)
        rows = int((np.abs(pr_g['J0']).sum(axis=1) > 0).sum())
        assert rows < pr_g['n']                                  # a first window is gauge-deficient: the cut removes directions


def test_emulated_marginalization_many_frame0_landmarks(simt_handle):
    """m = 15 + 60 landmarks > the LDS leading dimension of a 5-frame window (57): the case that used to leave the fast path."""
    marginalize_many_frame0_landmarks(simt_handle, K=5, L=70, w0=2, n_frames=10, min_m=60)


def test_emulated_enlarged_window_marginalization(simt_handle):
    """K = 16 takes the large-window path by itself: kept block n = 106 > 96 (pivoted Cholesky in global memory), camera part of
    the projection assembly in two entry passes."""
    marginalize_many_frame0_landmarks(simt_handle, K=16, L=40, w0=1, n_frames=20, min_m=20)


def test_emulated_prior_stays_on_the_device_between_frames(simt_handle):
    resident_prior_chain(simt_handle, L=16)


def test_emulated_graph_and_direct_launches_agree(simt_handle):
    """vg_ba_set_launch_mode under emulation: the emulated runtime records the launches of a capturing stream (grid, block, LDS
    size, argument values) and replays them on hipGraphLaunch, so the capture key, the replay over re-uploaded data and the
    re-capture on a size change are exercised here."""
    launch_modes_agree(simt_handle, L=14, nwin=2)


@pytest.mark.parametrize("K", [4, 7, 12])
def test_emulated_solve_other_window_sizes(simt_handle, K):
    """The speed-bias chain is eliminated from both ends towards block K / 2: even, odd and maximal K."""
    seq = synth.SyntheticSequence(11 + K, n_frames=K + 1, K=K, L=24)
    _check_solve(simt_handle, seq.window(0))


@pytest.mark.parametrize("ex,td", [(1, 1), (0, 1), (1, 0)])
def test_emulated_solve_with_extrinsic_and_td(simt_handle, ex, td):
    """The blocks among the extrinsic pose and td have a workgroup of their own in the accumulation kernel."""
    seq = synth.SyntheticSequence(70 + 2 * ex + td, n_frames=6, K=5, L=20, estimate_extrinsic=ex, estimate_td=td)
    _check_solve(simt_handle, seq.window(0))


# ---- large-window path (ba_big_schur_kernel / ba_solve_big_kernel / ba_big_step_kernel), forced on small windows ---------
@pytest.fixture()
def simt_large(simt_handle):
    simt_handle.ba_set_large_window(True)
    yield simt_handle
    simt_handle.ba_set_large_window(False)


def test_emulated_large_window_path_matches_oracle(simt_large):
    _check_solve(simt_large, synth.SyntheticSequence(3, L=30).window(0))


@pytest.mark.parametrize("name", ["overflowing_landmark", "far_origin"])
def test_emulated_large_window_path_trust_region_branches(simt_large, name):
    """Failed factorisations (retried in the next round with mu x 10), invalid steps, FAILURE after five of them, an
    interpolated dogleg step and the parameter-tolerance exit: the state machine spread over the launches of the large path
    (rejected steps: tests/test_sharded_window_cpu.py)."""
    build, need = FX.BRANCH_FIXTURES[name]
    prob = build()
    _, _, summ = _check_solve(simt_large, prob, rtol_cost=FX.COST_RTOL.get(name, 1e-6))
    assert need <= FX.trace_features(summ)


def test_emulated_large_window_path_with_relocalisation_extrinsic_td(simt_large):
    seq = synth.SyntheticSequence(23, n_frames=13, K=12, L=24)
    _check_solve(simt_large, seq.window(0))
    seq = synth.SyntheticSequence(73, n_frames=6, K=5, L=20, estimate_extrinsic=1, estimate_td=1)
    _check_solve(simt_large, seq.window(0))


def test_emulated_enlarged_window_31_frames(simt_handle):
    """K = 31 (WINDOW_SIZE 30): Rc = 186 -> twelve 16-column tiles, S fills 140 KB of LDS, IMU factors in two passes; few
    landmarks so that the emulator finishes in seconds (the full 2000-landmark window: tests/test_ba_large_gpu.py)."""
    from oracle import ba_cpu
    seq = synth.SyntheticSequence(5, n_frames=32, K=31, L=48)
    prob = synth.SyntheticSequence.anchor_prior(seq.window(0))
    st_o, sm_o, _ = ba_cpu.optimize(prob, margin_flag=ba.VG_MARGIN_NONE)
    st, sm, _ = simt_handle.ba_optimize(prob)
    n = sm_o['num_iterations']
    assert sm['status'] == 0 and sm['num_iterations'] == n and list(sm['it_flags'][:n]) == list(sm_o['it_flags'][:n])
    np.testing.assert_allclose(sm['it_cost'][:n], sm_o['it_cost'][:n], rtol=1e-6)
    assert np.abs(st['pose'] - st_o['pose']).max() < 1e-4 * max(1.0, np.abs(st_o['pose'][:, :3]).max())
    assert np.abs(st['sb'] - st_o['sb']).max() < 1e-4 * max(1.0, np.abs(st_o['sb']).max())
    assert np.allclose(st['inv_depth'], st_o['inv_depth'], rtol=1e-4, atol=1e-6)


def test_emulated_solver_time_cap(simt_handle):
    """vg_ba_problem::max_solver_time_s under emulation (both solve kernels): an expired cap stops before the first iteration."""
    prob = synth.SyntheticSequence(12, L=20).window(0)
    for large in (False, True):
        simt_handle.ba_set_large_window(large)
        try:
            st, sm, _ = simt_handle.ba_optimize(dict(prob, max_solver_time_s=1e-9))
        finally:
            simt_handle.ba_set_large_window(False)
        assert sm['status'] == 0 and sm['num_iterations'] == 0 and sm['final_cost'] == sm['initial_cost']


@pytest.mark.parametrize("order", ["reverse", "shuffle"])
def test_emulated_ba_under_other_fiber_orders(simt_handle, monkeypatch, order):
    """The whole BA path (solve pipeline with a prior, both marginalization flags, the large-window path, IMU pre-integration,
    triangulation) with the lanes of every workgroup scheduled in reverse and shuffled: a missing barrier or a lane-exchange through
    LDS without a wave barrier shows up as an order-dependent result (how two races in the marginalization eigen-solver were found)."""
    monkeypatch.setenv("SIMT_ORDER", order)
    seq = synth.SyntheticSequence(40, L=30)
    first = seq.window(0)
    st, sm, pr = simt_handle.ba_optimize(first, ba.VG_MARGIN_OLD)
    assert sm['status'] == 0 and pr is not None
    prob = seq.next_window(st, pr, 1)
    _check_solve(simt_handle, prob)                              # solve with the prior against the oracle
    x, _ = B.solve(prob)
    at = dict(prob)
    at.update(pose=x['pose'], sb=x['sb'], ex=x['ex'], td=x['td'], inv_depth=x['inv_depth'], max_iters=0)
    for flag in (B.MARGIN_OLD, B.MARGIN_SECOND_NEW):
        _, _, pr_o = B.optimization(at, flag)
        _, sm_g, pr_g = simt_handle.ba_optimize(at, flag)
        assert sm_g['status'] == 0 and (pr_g is None) == (pr_o is None)
        if pr_o is not None:
            _check_prior(pr_g, pr_o)
    simt_handle.ba_set_large_window(True)
    try:
        _check_solve(simt_handle, synth.SyntheticSequence(3, L=30).window(0))
    finally:
        simt_handle.ba_set_large_window(False)


@pytest.mark.parametrize("seed,L", [(3, 218), (4, 215), (5, 150)])
def test_emulated_fused_projection_kernel_over_several_chunks(simt_handle, seed, L):
    """ba_linacc_proj_kernel takes the landmarks in chunks of ~440 factors (LDS).  The first two windows have 882 / 881 factors with
    the last landmark starting BELOW factor 880: the number of chunks is the last landmark's chunk, not ceil(F / 440) — the version
    that derived it from F walked into an empty chunk on the hardware (round 4).  Window three: the EuRoC-sized window (two chunks)."""
    prob = synth.SyntheticSequence(seed, L=L).window(0)
    prob['max_iters'] = 3
    F = int(sum(prob['lm_nobs']) - len(prob['lm_nobs']))
    assert F > 440
    simt_handle.ba_set_fused_min_windows(1)          # (a single window takes the spread kernels by default)
    try:
        _check_solve(simt_handle, prob)
        simt_handle.ba_upload([prob])
        prof = simt_handle.ba_run_profiled()
        assert prof["ba_linacc_proj_kernel"][1] == 3 and prof["ba_linearize_imu_kernel+ba_linearize_proj_kernel"][1] == 2     # 3 fused rounds + the cost-only pass
    finally:
        simt_handle.ba_set_fused_min_windows(32)


def test_emulated_fused_and_spread_kernels_agree(simt_handle):
    """the same window with a prior, IMU factors and two chunks of projection factors through both launch structures: identical
    decisions, states equal to rounding (the sums run in different orders)"""
    seq = synth.SyntheticSequence(21, L=90)
    st, _, pr = simt_handle.ba_optimize(seq.window(0), ba.VG_MARGIN_OLD)
    prob = seq.next_window(st, pr, 1)
    out = []
    for n in (0, 1):
        simt_handle.ba_set_fused_min_windows(n)
        try:
            out.append(simt_handle.ba_optimize(prob, ba.VG_MARGIN_OLD))
        finally:
            simt_handle.ba_set_fused_min_windows(32)
    (sa, ma, pa), (sb, mb, pb) = out
    assert ma['num_iterations'] == mb['num_iterations'] and list(ma['it_flags']) == list(mb['it_flags'])
    assert np.abs(sa['pose'] - sb['pose']).max() < 1e-8 and np.abs(sa['sb'] - sb['sb']).max() < 1e-8
    assert np.isclose(ma['final_cost'], mb['final_cost'], rtol=1e-8)
    assert pa['n'] == pb['n'] and np.abs(pa['J0'].T @ pa['J0'] - pb['J0'].T @ pb['J0']).max() < 1e-6 * np.abs(pa['J0'].T @ pa['J0']).max()


def test_solve_kernel_lds_carve_leaves_room_for_a_second_window_per_cu(simt_handle):
    """Round 5: the per-round solve kernel of a EuRoC-shape window (K = 11, prior present) must fit HALF a CU's 160 KB of LDS -- the
    coupling rows no longer live there (chain_schur streams them) -- so that two windows are resident per CU; the other half of the
    bargain (256 threads x <= 256 VGPRs) is pinned in tests/test_codegen_guard.py."""
    seq = synth.SyntheticSequence(5, L=150)
    prob = seq.window(1)
    simt_handle.ba_upload([ba.PackedProblem(prob)], [ba.VG_MARGIN_NONE])
    assert simt_handle.ba_info()['lds_bytes'] <= 80 * 1024 - 2048


def test_emulated_both_builds_of_the_solve_kernel_agree(simt_handle):
    """ba_solve_kernel (4 wavefronts per window: batches) and ba_solve_w8_kernel (8 wavefronts: fewer than 32 windows, the drop-in's
    single window) are the same source compiled for two thread counts (csrc/ba_solve_w8.hip).  A batch of 32 small windows takes the
    first, each window alone the second: same accept / reject trace, states equal to rounding (sums strided over the threads of a
    workgroup are grouped differently), both equal to the oracle.  The GPU form of this test: test_throughput_and_latency_layouts_agree."""
    probs = [synth.SyntheticSequence(300 + s, K=5, L=8).window(0) for s in range(32)]
    simt_handle.ba_upload(probs, [ba.VG_MARGIN_NONE] * len(probs))
    simt_handle.ba_run_async()
    st, sm, _ = simt_handle.ba_download()
    for i in (0, 13, 31):
        s1, m1, _ = simt_handle.ba_optimize(probs[i])
        assert sm[i]['status'] == 0 and m1['status'] == 0 and sm[i]['num_iterations'] == m1['num_iterations']
        assert np.array_equal(sm[i]['it_flags'], m1['it_flags'])
        # (1e-9 relative to the size of the states: these 5-frame / 8-landmark windows are poorly conditioned, the builds differ in the
        #  grouping of strided sums and in the order in which the two halves of the Schur complement leave S)
        assert np.abs(st[i]['pose'] - s1['pose']).max() < 1e-9 * max(1.0, np.abs(s1['pose']).max())
        assert np.abs(st[i]['sb'] - s1['sb']).max() < 1e-9 * max(1.0, np.abs(s1['sb']).max())
        assert np.isclose(sm[i]['final_cost'], m1['final_cost'], rtol=1e-9)
        x, summ = B.solve(probs[i])
        ref = B.double2vector(probs[i], x)
        assert np.abs(s1['pose'] - ref['pose']).max() < 1e-6 and np.abs(st[i]['pose'] - ref['pose']).max() < 1e-6
