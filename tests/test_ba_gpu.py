"""GPU parity tests of the BA path: HIP (through the C-ABI) vs the NumPy oracle on identical inputs.
Tolerances: factor level 1e-9 relative (IMU rows: 1e-6 of the row weight, the information matrix is
ill-conditioned and the device factorises the covariance instead of inverting it); solve level: identical
accept/reject sequence, costs 1e-6 relative, states 1e-4 relative (BASELINE.json north_star)."""
import numpy as np
import pytest

from oracle import ba_numpy as B
from vins_mono_amd import ba, synth

import ba_fixtures as FX

pytestmark = pytest.mark.gpu


def _oracle_factor_tables(prob):
    st = B.state_of(prob)
    lay = B.Layout(prob)
    facs = B.factor_list(prob)
    pr, pJ = np.zeros((len(facs), 2)), np.zeros((len(facs), 2, 20))
    for f, (l, fi, fj, oi, oj) in enumerate(facs):
        if lay.est_td:
            r, J = B.projection_td_factor(st['pose'][fi], st['pose'][fj], st['ex'], st['inv_depth'][l], st['td'], oi, oj,
                                          prob['focal'], prob['tr'], prob['row'])
            pJ[f, :, 19] = J[4][:, 0]
        else:
            r, J = B.projection_factor(st['pose'][fi], st['pose'][fj], st['ex'], st['inv_depth'][l],
                                       np.array([oi[0], oi[1], 1.0]), np.array([oj[0], oj[1], 1.0]), prob['focal'])
        pr[f] = r
        pJ[f, :, 0:6], pJ[f, :, 6:12], pJ[f, :, 12:18], pJ[f, :, 18] = J[0], J[1], J[2], J[3][:, 0]
    K = lay.K
    ir, iJ = np.zeros((K - 1, 15)), np.zeros((K - 1, 15, 30))
    for k in range(K - 1):
        r, J = B.imu_factor(prob['imu'][k], st['pose'][k], st['sb'][k], st['pose'][k + 1], st['sb'][k + 1], prob['g_norm'])
        ir[k] = r
        iJ[k] = np.hstack(J)
    qr = None
    if prob.get('prior') is not None:
        now = [B.get_block(st, k, i) for (k, i) in prob['prior']['blocks']]
        qr, _ = B.prior_factor(prob['prior'], now)
    return pr, pJ, ir, iJ, qr


def _window_with_prior(seed, L=40, **kw):
    seq = synth.SyntheticSequence(seed, L=L, **kw)
    prob = seq.window(0)
    x, _ = B.solve(prob)
    st = B.double2vector(prob, x)
    pr = B.marginalize(prob, st, B.MARGIN_OLD)
    return seq, prob, seq.next_window(st, pr, 1)


@pytest.mark.parametrize("ex,td", [(0, 0), (1, 1)])
def test_factor_parity(handle, ex, td):
    _, _, prob = _window_with_prior(21 + ex, estimate_extrinsic=ex, estimate_td=td)
    if td:
        prob['tr'] = 0.02
    pr, pJ, ir, iJ, qr = _oracle_factor_tables(prob)
    out = handle.ba_eval_factors(prob)
    assert np.allclose(out['proj_r'], pr, rtol=1e-9, atol=1e-9)
    if not td:
        pJ[:, :, 19] = 0
    assert np.allclose(out['proj_J'], pJ, rtol=1e-9, atol=1e-8)
    for k in range(prob['pose'].shape[0] - 1):
        W = B.imu_sqrt_info(prob['imu'][k]['covariance'])
        scale = np.abs(W).sum(axis=1)
        assert np.allclose(out['imu_r'][k] / scale, ir[k] / scale, atol=1e-6), k
        assert np.allclose(out['imu_J'][k] / scale[:, None], iJ[k] / scale[:, None], atol=1e-6), k
        # sqrt_info-invariant quantities agree much tighter
        assert np.isclose(out['imu_r'][k] @ out['imu_r'][k], ir[k] @ ir[k], rtol=1e-7)
    assert np.allclose(out['prior_r'], qr, rtol=1e-10, atol=1e-10 * np.abs(qr).max())


_TERM = {'NO_CONVERGENCE': 0, 'CONVERGENCE': 1, 'FAILURE': 2}


def _check_solve(handle, prob, rtol_state=1e-4, rtol_cost=1e-6):
    """GPU vs oracle, iteration by iteration: same number of iterations, same valid / accepted flags, same termination,
    and per iteration the cost, candidate cost, model cost change, radius and dogleg step norm; then the gauge-fixed
    states (positions are compared relative to frame 0 as well, so a far-away world origin cannot hide an error)."""
    x, summ = B.solve(prob)
    ref = B.double2vector(prob, x)
    st, sm, _ = handle.ba_optimize(prob)
    assert sm['status'] == 0
    its = summ['iterations']
    assert sm['num_iterations'] == summ['num_iterations']
    assert sm['termination'] == _TERM[summ['termination']]
    flags = [(1 if it.get('valid') else 0) | (2 if it.get('accepted') else 0) for it in its]
    assert list(sm['it_flags'][:len(flags)]) == flags
    assert np.isclose(sm['initial_cost'], summ['initial_cost'], rtol=1e-9)
    for k, it in enumerate(its):
        if it.get('valid'):
            assert np.isclose(sm['it_cost'][k], it['cost'], rtol=rtol_cost, atol=1e-9), k
            assert np.isclose(sm['it_cost_cand'][k], it['cost_cand'], rtol=rtol_cost, atol=1e-9), k
            assert np.isclose(sm['it_model'][k], it['model_change'], rtol=10 * rtol_cost), k
            assert np.isclose(sm['it_radius'][k], it['radius'], rtol=1e-6), k
            assert np.isclose(sm['it_step_norm'][k], it['step_norm'], rtol=10 * rtol_cost), k
    assert np.isclose(sm['final_cost'], summ['final_cost'], rtol=rtol_cost)
    rel_g = st['pose'][:, :3] - st['pose'][0, :3]
    rel_o = ref['pose'][:, :3] - ref['pose'][0, :3]
    assert np.abs(rel_g - rel_o).max() < rtol_state * max(1.0, np.abs(rel_o).max())
    scale_p = max(1.0, np.abs(ref['pose'][:, :3]).max())
    assert np.abs(st['pose'][:, :3] - ref['pose'][:, :3]).max() < rtol_state * scale_p
    assert np.abs(st['pose'][:, 3:] - ref['pose'][:, 3:]).max() < rtol_state
    assert np.abs(st['sb'] - ref['sb']).max() < rtol_state * max(1.0, np.abs(ref['sb']).max())
    assert np.allclose(st['inv_depth'], ref['inv_depth'], rtol=1e-4, atol=1e-6)
    assert np.allclose(st['ex'], ref['ex'], atol=rtol_state)
    return st, sm, summ


@pytest.mark.parametrize("name", sorted(FX.BRANCH_FIXTURES))
def test_trust_region_branches(handle, name):
    """every branch of the restated Ceres loop (Cauchy / interpolated / Gauss-Newton step, rejected
    step with radius halving and step reuse, mu escalation after a failed factorisation, invalid-step counter and
    FAILURE, function- and parameter-tolerance exits) on a fixture whose ORACLE trace is asserted to contain it."""
    build, need = FX.BRANCH_FIXTURES[name]
    prob = build()
    _, _, summ = _check_solve(handle, prob, rtol_cost=FX.COST_RTOL.get(name, 1e-6))
    assert need <= FX.trace_features(summ), (name, need - FX.trace_features(summ))


def test_trust_region_fixtures_cover_all_branches():
    """The union of the fixtures above reaches every branch (oracle side; the GPU side is the parametrised test)."""
    seen = set()
    for name, (build, need) in FX.BRANCH_FIXTURES.items():
        seen |= need
    assert seen >= {'gn', 'cauchy', 'dogleg', 'rejected', 'mu_escalation', 'invalid', 'failure', 'function_tolerance',
                    'parameter_tolerance'}


@pytest.mark.parametrize("seed", [1, 2])
def test_solve_parity_no_prior(handle, seed):
    prob = synth.SyntheticSequence(seed, L=60).window(0)
    _check_solve(handle, prob)


def test_solve_parity_with_prior(handle):
    _, _, prob2 = _window_with_prior(4, L=60)
    _check_solve(handle, prob2)


def test_solve_parity_extrinsic_td(handle):
    _, _, prob2 = _window_with_prior(5, L=50, estimate_extrinsic=1, estimate_td=1)
    _check_solve(handle, prob2)


def test_solve_full_size_two_chunks(handle):
    """EuRoC-size window (150 landmarks, F > one LDS chunk)."""
    prob = synth.SyntheticSequence(9, L=150).window(0)
    _check_solve(handle, prob)


def test_batch_is_bit_reproducible(handle):
    probs = [synth.SyntheticSequence(30 + s, L=50).window(0) for s in range(3)]
    handle.ba_upload(probs)
    handle.ba_run_async()
    st_a, sm_a, _ = handle.ba_download()
    handle.ba_upload(list(reversed(probs)))
    handle.ba_run_async()
    st_b, sm_b, _ = handle.ba_download()
    for a, b in zip(st_a, reversed(st_b)):
        assert np.array_equal(a['pose'], b['pose']) and np.array_equal(a['inv_depth'], b['inv_depth'])
    single, _, _ = handle.ba_optimize(probs[1])
    assert np.array_equal(single['pose'], st_a[1]['pose'])


def test_fused_kernel_batch_is_bit_reproducible_and_agrees_with_the_spread_kernels():
    """A batch of >= 32 windows takes ba_linacc_proj_kernel (IMU factors, prior and projection factors linearised and accumulated by
    one workgroup per window, camera blocks on MFMA, records in LDS): run to run bit-identical, independent of the window's slot in
    the batch, and equal to the spread kernels (vg_ba_set_fused_min_windows(0)) to rounding; EuRoC-sized windows (two chunks) with
    priors, one checked against the oracle."""
    handle = ba.Handle()          # (its own: the batch slots of the shared handle keep what a run leaves in them, see resident_prior_chain)
    seqs = [synth.SyntheticSequence(300 + s, L=150 if s < 4 else 40) for s in range(36)]
    firsts = [q.window(0) for q in seqs]
    handle.ba_upload(firsts, [ba.VG_MARGIN_OLD] * len(firsts))
    handle.ba_run_async()
    st1, _, pr1 = handle.ba_download()
    probs = [q.next_window(st1[k], pr1[k], 1) for k, q in enumerate(seqs)]
    flags = [ba.VG_MARGIN_OLD] * len(probs)
    handle.ba_upload(probs, flags)
    handle.ba_run_async()
    st_a, sm_a, pr_a = handle.ba_download()
    prof = handle.ba_run_profiled()
    assert prof["ba_linacc_proj_kernel"][1] == 8 and prof["ba_linearize_imu_kernel+ba_linearize_proj_kernel"][1] == 2       # fused rounds + the cost-only pass
    handle.ba_upload(list(reversed(probs)), flags)
    handle.ba_run_async()
    st_b, sm_b, _ = handle.ba_download()
    for a, b in zip(st_a, reversed(st_b)):
        assert np.array_equal(a['pose'], b['pose']) and np.array_equal(a['sb'], b['sb']) and np.array_equal(a['inv_depth'], b['inv_depth'])
    handle.ba_set_fused_min_windows(0)
    try:
        handle.ba_upload(probs, flags)
        handle.ba_run_async()
        st_c, sm_c, pr_c = handle.ba_download()
    finally:
        handle.ba_set_fused_min_windows(32)
    for a, c, ma, mc in zip(st_a, st_c, sm_a, sm_c):
        assert ma['num_iterations'] == mc['num_iterations'] and list(ma['it_flags']) == list(mc['it_flags'])
        assert np.abs(a['pose'] - c['pose']).max() < 1e-7 and np.abs(a['sb'] - c['sb']).max() < 1e-7
    x, summ = B.solve(probs[0])
    ref = B.double2vector(probs[0], x)
    assert summ['num_iterations'] == sm_a[0]['num_iterations']
    assert np.abs(st_a[0]['pose'] - ref['pose']).max() < 1e-6 and np.abs(st_a[0]['sb'] - ref['sb']).max() < 1e-6
    handle.close()


def _check_prior(gp, op, tol=1e-7, dx=1e-12):
    """Prior parity against (A, b) = the Schur complement BEFORE the second eigen-decomposition, recomputed in
    extended precision from the oracle's assembled system (the double-precision eigen pseudo-inverse of Amm, whose
    entries span 1e2..1e12, is itself only good to ~1e-8 relative in either implementation).
    marginalize() drops eigen-directions with lambda <= 1e-8; for a gauge-deficient window those eigenvalues are
    rounding noise (observed -0.09 .. +1e-3 against a spectrum reaching 4e7), so WHICH subspace is dropped — and
    therefore J0^T r0 — is only defined up to the component of b in that noise subspace.  Both sides are therefore
    held to the Schur complement: the HIP result must reproduce (A, b) at least as well as the oracle's own
    eigen-reconstruction does (x3 slack), plus `dx` * Hessian row sums when the linearisation points differ by dx."""
    assert gp is not None and op is not None
    assert gp['blocks'] == op['blocks']
    assert gp['n'] == op['n'] and gp['m'] == op['m']
    H_g, H_o = gp['J0'].T @ gp['J0'], op['J0'].T @ op['J0']
    g_g, g_o = gp['J0'].T @ gp['r0'], op['J0'].T @ op['r0']
    A, b = B.schur_extended(op['A_full'], op['b_full'], op['m'])     # extended-precision yardstick
    amax = np.abs(A).max()
    rows = np.abs(A).sum(axis=1)
    assert np.abs(H_g - A).max() < max(3 * np.abs(H_o - A).max(), tol * amax) + dx * amax
    g_scale = np.abs(op['J0']).T @ np.abs(op['r0'])
    assert np.abs(g_g - b).max() < 3 * np.abs(g_o - b).max() + tol * g_scale.max() + (dx * rows).max()
    for a, c in zip(gp['x0'], op['x0']):
        assert np.allclose(a, c, atol=max(1e-9, dx))


@pytest.mark.parametrize("ex,td", [(0, 0), (1, 1)])
def test_marginalize_old_parity(handle, ex, td):
    """M1-M5 in isolation: both sides marginalise at the SAME state (the oracle's optimum, max_iters = 0)."""
    seq = synth.SyntheticSequence(40 + ex, L=60, estimate_extrinsic=ex, estimate_td=td)
    prob = seq.window(0)
    x, _ = B.solve(prob)
    at = dict(prob)
    at.update(pose=x['pose'], sb=x['sb'], ex=x['ex'], td=x['td'], inv_depth=x['inv_depth'], max_iters=0)
    st_o, _, pr_o = B.optimization(at, B.MARGIN_OLD)
    st_g, sm_g, pr_g = handle.ba_optimize(at, ba.VG_MARGIN_OLD)
    assert sm_g['status'] == 0 and sm_g['num_iterations'] == 0
    assert np.allclose(st_g['pose'], st_o['pose'], atol=1e-13) and np.allclose(st_g['sb'], st_o['sb'], atol=1e-13)
    _check_prior(pr_g, pr_o)


def marginalize_many_frame0_landmarks(h, K=11, L=200, w0=6, n_frames=18, min_m=100):
    """A window that has been sliding for a while: most landmarks are anchored at frame 0 (long tracks pile up there as
    slideWindow() shifts start_frame down, feature_manager.cpp:297-323), so the dropped block [pose 0 | speed-bias 0 | their
    inverse depths] is wider than the kept part.  Amm^-1 [Amr | bmm] comes from the block elimination over the diagonal
    landmark part for any such width (no m x m eigen-decomposition, no LDS-size cliff).  Shared with the emulator tests."""
    seq = synth.SyntheticSequence(47, n_frames=n_frames, K=K, L=L)
    prob = seq.window(w0)
    m = 15 + int((np.asarray(prob['lm_start']) == 0).sum())
    assert m >= min_m, m
    x, _ = B.solve(prob)
    at = dict(prob)
    at.update(pose=x['pose'], sb=x['sb'], ex=x['ex'], td=x['td'], inv_depth=x['inv_depth'], max_iters=0)
    _, _, pr_o = B.optimization(at, B.MARGIN_OLD)
    _, sm_g, pr_g = h.ba_optimize(at, ba.VG_MARGIN_OLD)
    assert sm_g['status'] == 0 and pr_g['m'] == m
    _check_prior(pr_g, pr_o)


def test_marginalize_many_frame0_landmarks(handle):
    marginalize_many_frame0_landmarks(handle)


@pytest.mark.parametrize("flag", ["old", "second_new"])
def test_marginalization_eigen_form_matches_the_square_root_form(handle, flag):
    """vg_ba_set_marg_mode: the default prior factor is the pivoted-Cholesky square root of the kept system, VG_MARG_EIGEN the
    reference's S^1/2 V^T (marginalization_factor.cpp:285-296).  Both must reproduce the same (A', b') — what everything downstream
    consumes — and the same constant |r0|^2 = b'^T A'^+ b'; the next solve must not see the difference."""
    if flag == "old":
        seq = synth.SyntheticSequence(43, L=80)
        prob = seq.window(0)
        mflag = ba.VG_MARGIN_OLD
    else:
        seq, _, prob = _window_with_prior(6, L=150)
        mflag = ba.VG_MARGIN_SECOND_NEW
    out = {}
    for mode in (ba.VG_MARG_SQRT, ba.VG_MARG_EIGEN):
        handle.ba_set_marg_mode(mode)
        try:
            out[mode] = handle.ba_optimize(prob, mflag)
        finally:
            handle.ba_set_marg_mode(ba.VG_MARG_SQRT)
    (st_a, _, pa), (st_b, _, pb) = out[ba.VG_MARG_SQRT], out[ba.VG_MARG_EIGEN]
    assert np.array_equal(st_a['pose'], st_b['pose'])            # the solve is the same launch sequence
    assert pa['blocks'] == pb['blocks'] and pa['n'] == pb['n']
    Ha, Hb = pa['J0'].T @ pa['J0'], pb['J0'].T @ pb['J0']
    ga, gb = pa['J0'].T @ pa['r0'], pb['J0'].T @ pb['r0']
    assert np.abs(Ha - Hb).max() < 1e-7 * np.abs(Hb).max()
    assert np.abs(ga - gb).max() < 1e-6 * (np.abs(pb['J0']).T @ np.abs(pb['r0'])).max()
    assert np.isclose(pa['r0'] @ pa['r0'], pb['r0'] @ pb['r0'], rtol=1e-4)
    assert np.count_nonzero(np.abs(pa['J0']).sum(axis=1)) <= pa['n']
    if flag == "old":
        nxt_a, nxt_b = seq.next_window(st_a, pa, 1), None
        seq2 = synth.SyntheticSequence(43, L=80)
        seq2.window(0)
        nxt_b = seq2.next_window(st_b, pb, 1)
        s2a, sma, _ = handle.ba_optimize(nxt_a)
        s2b, smb, _ = handle.ba_optimize(nxt_b)
        assert sma['num_iterations'] == smb['num_iterations']
        assert np.abs(s2a['pose'] - s2b['pose']).max() < 1e-6 and np.abs(s2a['sb'] - s2b['sb']).max() < 1e-6


def test_optimization_chain_two_windows(handle):
    """solve + MARGIN_OLD twice; each side consumes its OWN prior and state (what a drop-in runs)."""
    seq = synth.SyntheticSequence(40, L=60)
    prob = seq.window(0)
    st_o, _, pr_o = B.optimization(prob, B.MARGIN_OLD)
    st_g, sm_g, pr_g = handle.ba_optimize(prob, ba.VG_MARGIN_OLD)
    assert sm_g['status'] == 0
    _check_prior(pr_g, pr_o, dx=1e-9)
    prob2_o = seq.next_window(st_o, pr_o, 1)
    seq_b = synth.SyntheticSequence(40, L=60)
    seq_b.window(0)
    prob2_g = seq_b.next_window(st_g, pr_g, 1)
    st2_o, s2_o, pr2_o = B.optimization(prob2_o, B.MARGIN_OLD)
    st2_g, sm2_g, pr2_g = handle.ba_optimize(prob2_g, ba.VG_MARGIN_OLD)
    # the prior's constant cost term |r0|^2 divides by eigenvalues just above the 1e-8 cut (gauge directions),
    # so the cost VALUE is only reproducible to ~1e-4; gradient/Hessian (hence the states) are not affected
    assert np.isclose(sm2_g['final_cost'], s2_o['final_cost'], rtol=1e-3)
    assert np.abs(st2_g['pose'] - st2_o['pose']).max() < 1e-4 * max(1.0, np.abs(st2_o['pose']).max())
    assert np.abs(st2_g['sb'] - st2_o['sb']).max() < 1e-4 * max(1.0, np.abs(st2_o['sb']).max())
    # the second prior is a function of the second optimum: the two chains' optima differ by ~1e-5, which moves the
    # Hessian by a few 1e-4 relative (lever arms), so the chained priors are NOT comparable element-wise.  Parity of
    # the marginalization itself is asserted at the SAME linearisation point: the oracle marginalises the HIP chain's
    # own window at the HIP chain's own optimum.
    pr2_ref = B.marginalize(prob2_g, st2_g, B.MARGIN_OLD)
    _check_prior(pr2_g, pr2_ref, dx=1e-9)


def _same_prior(a, b):
    return (a is None) == (b is None) and (a is None or (a['n'] == b['n'] and a['blocks'] == b['blocks'] and np.array_equal(a['J0'], b['J0'])
                                                      and np.array_equal(a['r0'], b['r0'])
                                                      and all(np.array_equal(u, v) for u, v in zip(a['x0'], b['x0']))))


def _same_state(a, b):
    return all(np.array_equal(a[k], b[k]) for k in ('pose', 'sb', 'ex', 'inv_depth')) and a['td'] == b['td']


def resident_prior_chain(h, L=40):
    """VG_PRIOR_RESIDENT: the prior a run's marginalization leaves on the device feeds the next frame's solve of the same batch
    slot without crossing the host boundary — bit-identical to taking it through vg_ba_prior and back (last_marginalization_info
    staying in place between two calls of optimization(), estimator.cpp:703-709).  Shared with the CPU emulator tests."""
    seqs = [synth.SyntheticSequence(90 + s, n_frames=14, L=L) for s in range(3)]
    first = [q.window(0) for q in seqs]
    old = [ba.VG_MARGIN_OLD] * 3

    def step(probs, flags):
        h.ba_upload(probs, flags)
        h.ba_run_async()
        return h.ba_download()

    st1, sm1, pr1 = step(first, old)
    assert all(s['status'] == 0 for s in sm1) and all(p is not None for p in pr1)

    def with_priors(probs, priors):
        return [dict(p, prior=q) for p, q in zip(probs, priors)]
    # (next_window draws the new frame's noise from the sequence's generator: every window is built once, the prior swapped)
    w2 = [q.next_window(st1[i], pr1[i], 1) for i, q in enumerate(seqs)]
    # frame 2: slot 1 brings its prior from the host (a mixed batch), slots 0 and 2 use what the run left on the device
    st2r, sm2r, pr2r = step(with_priors(w2, ['resident', pr1[1], 'resident']), old)
    # frame 3 on top of it: the carry a second time; slot 1 is not marginalized in this run ...
    w3 = [q.next_window(st2r[i], pr2r[i], 2) for i, q in enumerate(seqs)]
    flags3 = [ba.VG_MARGIN_OLD, ba.VG_MARGIN_NONE, ba.VG_MARGIN_OLD]
    st3r, sm3r, pr3r = step(with_priors(w3, ['resident'] * 3), flags3)
    assert pr3r[1] is None
    # ... so its slot still holds frame 2's prior while slots 0 and 2 have moved on: give those two frame 2's prior again
    st3m, sm3m, _ = step(with_priors(w3, [pr2r[0], 'resident', pr2r[2]]), [ba.VG_MARGIN_NONE] * 3)
    # the same chain with every prior taken through the host
    st1h, _, pr1h = step(first, old)
    assert all(_same_state(a, b) for a, b in zip(st1, st1h)) and all(_same_prior(a, b) for a, b in zip(pr1, pr1h))
    st2h, sm2h, pr2h = step(w2, old)
    for i in range(3):
        assert sm2r[i]['status'] == 0 and sm2r[i]['num_iterations'] == sm2h[i]['num_iterations']
        assert _same_state(st2r[i], st2h[i]) and _same_prior(pr2r[i], pr2h[i]), i
    st3h, sm3h, pr3h = step(w3, flags3)
    for i in range(3):
        assert _same_state(st3r[i], st3h[i]) and _same_prior(pr3r[i], pr3h[i]), i
        assert _same_state(st3m[i], st3h[i]), i
    # a slot that holds nothing is refused
    hh = h
    with pytest.raises(RuntimeError, match="status -1.*holds no prior"):
        hh.ba_upload(with_priors(w2, ['resident'] * 3) + [dict(first[0], prior='resident')])       # a fourth slot was never used


def test_prior_stays_on_the_device_between_frames(handle):
    resident_prior_chain(handle, L=60)


def launch_modes_agree(h, L=40, nwin=3):
    """vg_ba_set_launch_mode: the solve pipeline replayed as a hipGraph gives the bits of the direct launches — same kernels, same
    arguments.  Covers the replay of one captured graph over several runs and re-uploads (the kernel arguments do not change, the
    data does), a re-capture when the size class changes, and the way back to direct launches.  Shared with the CPU emulator
    tests (which record the launches of a capturing stream and replay them)."""
    seqs = [synth.SyntheticSequence(130 + s, n_frames=13, L=L) for s in range(nwin)]
    first = [q.window(0) for q in seqs]
    old = [ba.VG_MARGIN_OLD] * nwin

    def chain(mode):
        h.ba_set_launch_mode(mode)
        before = h.ba_launch_stats()
        out = []
        probs, flags = first, old
        for frame in range(3):
            h.ba_upload(probs, flags)
            h.ba_run_async()
            st, sm, pr = h.ba_download()
            out.append((st, sm, pr))
            if frame < 2:
                probs = [q.next_window(st[i], pr[i], frame + 1) for i, q in enumerate(seqs)]
        # a batch of another size class (fewer windows, another landmark count): new grid sizes -> another graph
        h.ba_upload([synth.SyntheticSequence(7, L=max(8, L // 2)).window(0)], [ba.VG_MARGIN_NONE])
        h.ba_run_async()
        out.append(h.ba_download())
        h.ba_run_async()                                   # the same batch again: replay (re-solves from the uploaded state)
        out.append(h.ba_download())
        after = h.ba_launch_stats()
        return out, {k: after[k] - before[k] for k in ('graph_launches', 'graph_captures')}, after['mode']

    seqs_state = [q.rng.bit_generator.state for q in seqs] if hasattr(seqs[0], 'rng') else None
    direct, d_stats, d_mode = chain(ba.VG_LAUNCH_DIRECT)
    if seqs_state is not None:
        for q, s in zip(seqs, seqs_state):
            q.rng.bit_generator.state = s
    try:
        graph, g_stats, g_mode = chain(ba.VG_LAUNCH_GRAPH)
    finally:
        h.ba_set_launch_mode(ba.VG_LAUNCH_DEFAULT)
    assert d_mode == 'direct' and d_stats == {'graph_launches': 0, 'graph_captures': 0}
    assert g_mode == 'graph' and g_stats['graph_launches'] == 5 and 2 <= g_stats['graph_captures'] <= 5, g_stats
    for (st_d, sm_d, pr_d), (st_g, sm_g, pr_g) in zip(direct, graph):
        for a, b in zip(st_d if isinstance(st_d, list) else [st_d], st_g if isinstance(st_g, list) else [st_g]):
            for k in ('pose', 'sb', 'inv_depth'):
                assert np.array_equal(a[k], b[k]), k
        for a, b in zip(sm_d if isinstance(sm_d, list) else [sm_d], sm_g if isinstance(sm_g, list) else [sm_g]):
            assert a['status'] == 0 == b['status'] and a['final_cost'] == b['final_cost'] and a['num_iterations'] == b['num_iterations']
        for a, b in zip(pr_d if isinstance(pr_d, list) else [pr_d], pr_g if isinstance(pr_g, list) else [pr_g]):
            assert (a is None) == (b is None)
            if a is not None:
                assert np.array_equal(a['J0'], b['J0']) and np.array_equal(a['r0'], b['r0'])


def test_graph_and_direct_launches_agree(handle):
    launch_modes_agree(handle, L=60)


def test_marginalize_second_new_parity(handle):
    _, _, prob2 = _window_with_prior(6, L=150)
    K = prob2['pose'].shape[0]
    assert (B.KIND_POSE, K - 2) in prob2['prior']['blocks']
    prob2 = dict(prob2)
    prob2['max_iters'] = 0
    st_o, _, pr_o = B.optimization(prob2, B.MARGIN_SECOND_NEW)
    st_g, sm, pr_g = handle.ba_optimize(prob2, ba.VG_MARGIN_SECOND_NEW)
    _check_prior(pr_g, pr_o)
    assert (B.KIND_POSE, K - 2) not in pr_g['blocks'] and pr_g['n'] == prob2['prior']['n'] - 6


def test_marginalize_second_new_keeps_old_prior_when_pose_absent(handle):
    _, _, prob2 = _window_with_prior(7, L=8)
    K = prob2['pose'].shape[0]
    if (B.KIND_POSE, K - 2) in prob2['prior']['blocks']:
        pytest.skip("prior touches pose K-2 for this seed")
    _, _, pr_g = handle.ba_optimize(prob2, ba.VG_MARGIN_SECOND_NEW)
    assert pr_g is None        # valid == 0: caller keeps last_marginalization_info (estimator.cpp:935-936)


# ---------------------------------------------------------------------------------------- edge cases / full size
def relocalisation_problem(seed=51, loop_frame=3):
    """A window with relocalisation factors: loop frame = a perturbed copy of frame `loop_frame`, matches for landmarks whose
    track starts at or before it (estimator.cpp:781)."""
    seq = synth.SyntheticSequence(seed, L=40)
    prob = seq.window(0)
    relo_pose = prob['pose'][loop_frame].copy()
    relo_pose[:3] += [0.05, -0.03, 0.02]
    match = []
    c = seq.cfg
    Rr, Pr = B.q2R(relo_pose[3:]), relo_pose[:3]
    for l in range(len(prob['inv_depth'])):
        if prob['lm_start'][l] <= loop_frame and len(match) < 15:
            s = int(prob['lm_start'][l])
            o = prob['obs'][int(prob['obs_off'][l])]
            pc = np.array([o[0], o[1], 1.0]) / prob['inv_depth'][l]
            Xw = B.q2R(prob['pose'][s][3:]) @ (c['ric'] @ pc + c['tic']) + prob['pose'][s][:3]
            p = c['ric'].T @ (Rr.T @ (Xw - Pr) - c['tic'])
            match.append((l, p[0] / p[2], p[1] / p[2]))
    prob['relo'] = dict(pose=relo_pose, match=match)
    return prob


def test_relocalisation_factors(handle):
    """estimator.cpp:769-801: extra ProjectionFactors to a relocalisation pose (an extra pose block)."""
    prob = relocalisation_problem()
    x, summ = B.solve(prob)
    ref = B.double2vector(prob, x)
    st, sm, _ = handle.ba_optimize(prob)
    assert sm['status'] == 0 and sm['num_iterations'] == summ['num_iterations']
    assert np.isclose(sm['final_cost'], summ['final_cost'], rtol=1e-6)
    assert np.abs(st['pose'] - ref['pose']).max() < 1e-6 and np.abs(st['inv_depth'] - ref['inv_depth']).max() < 1e-6


def test_no_landmarks_and_skipped_imu_factor(handle):
    """IMU + prior only (every track filtered out), and one pre-integration longer than 10 s (estimator.cpp:714)."""
    _, _, prob = _window_with_prior(52, L=30)
    prob = dict(prob)
    prob.update(inv_depth=np.zeros(0), lm_start=np.zeros(0, np.int32), lm_nobs=np.zeros(0, np.int32), obs_off=np.zeros(0, np.int32),
                obs=np.zeros((0, 7)))
    prob['imu'] = [dict(m) for m in prob['imu']]
    prob['imu'][4]['sum_dt'] = 10.5                  # -> factor skipped on both sides
    x, summ = B.solve(prob)
    ref = B.double2vector(prob, x)
    st, sm, _ = handle.ba_optimize(prob)
    assert sm['status'] == 0 and sm['num_iterations'] == summ['num_iterations']
    assert np.isclose(sm['final_cost'], summ['final_cost'], rtol=1e-6, atol=1e-9)
    assert np.abs(st['pose'] - ref['pose']).max() < 1e-6 and np.abs(st['sb'] - ref['sb']).max() < 1e-6


def test_iteration_limits(handle):
    prob = synth.SyntheticSequence(53, L=30).window(0)
    for iters in (0, 1, 3):
        p = dict(prob)
        p['max_iters'] = iters
        x, summ = B.solve(p)
        st, sm, _ = handle.ba_optimize(p)
        assert sm['num_iterations'] == summ['num_iterations'] == iters
        assert np.isclose(sm['final_cost'], summ['final_cost'], rtol=1e-7)


def test_full_size_batch_against_cpp_oracle(handle):
    """BASELINE configs[3] shape: 24 EuRoC-size windows (150 landmarks, prior) in one launch vs oracle/ba_cpu.cpp;
    acceptance = BASELINE.json north_star: states within 1e-4 relative."""
    from oracle import ba_cpu
    seqs = [synth.SyntheticSequence(600 + s) for s in range(24)]
    first = [q.window(0) for q in seqs]
    handle.ba_upload(first, [ba.VG_MARGIN_OLD] * len(first))
    handle.ba_run_async()
    st1, sm1, pr1 = handle.ba_download()
    probs = [q.next_window(st1[i], pr1[i], 1) for i, q in enumerate(seqs)]
    handle.ba_upload(probs, [ba.VG_MARGIN_OLD] * len(probs))
    handle.ba_run_async()
    st2, sm2, pr2 = handle.ba_download()
    worst = 0.0
    for i, p in enumerate(probs):
        ref, sref, pref = ba_cpu.optimize(p, ba.VG_MARGIN_OLD)
        assert sm2[i]['status'] == 0 and sm2[i]['num_iterations'] == sref['num_iterations']
        e = max(np.abs(st2[i]['pose'][:, :3] - ref['pose'][:, :3]).max() / max(1.0, np.abs(ref['pose'][:, :3]).max()),
                np.abs(st2[i]['pose'][:, 3:] - ref['pose'][:, 3:]).max(),
                np.abs(st2[i]['sb'] - ref['sb']).max() / max(1.0, np.abs(ref['sb']).max()))
        worst = max(worst, e)
        assert pr2[i]['blocks'] == pref['blocks'] and pr2[i]['n'] == pref['n']
    assert worst < 1e-4, worst
    # size-independent properties of the outputs: unit quaternions, frame-0 gauge (position + yaw) preserved
    for i, p in enumerate(probs):
        assert np.allclose(np.linalg.norm(st2[i]['pose'][:, 3:], axis=1), 1.0, atol=1e-12)
        assert np.allclose(st2[i]['pose'][0, :3], p['pose'][0, :3], atol=1e-12)
        y0, y1 = B.R2ypr(B.q2R(p['pose'][0, 3:]))[0], B.R2ypr(B.q2R(st2[i]['pose'][0, 3:]))[0]
        assert abs(y0 - y1) < 1e-9


def test_imu_preintegration_on_device(handle):
    """vg_imu_preintegrate vs the two independent CPU restatements of IntegrationBase
    (oracle/ba_numpy.Preintegration, synth.preintegrate): mid-point states, 15x15 jacobian and covariance; ragged
    intervals (1, 20 and 120 samples, the last one longer than 10 s), repropagation with other biases."""
    rng = np.random.default_rng(77)
    noise = (0.08, 0.004, 4e-5, 2e-6)
    intervals, biases = [], []
    for n, dt in ((1, 0.005), (20, 0.005), (120, 0.1), (7, 0.0049)):
        iv = [(0.0, rng.normal(0, 1, 3) + [0, 0, 9.8], rng.normal(0, 0.3, 3))]
        for _ in range(n):
            iv.append((dt, rng.normal(0, 1, 3) + [0, 0, 9.8], rng.normal(0, 0.3, 3)))
        intervals.append(iv)
        biases.append((rng.normal(0, 0.05, 3), rng.normal(0, 0.02, 3)))
    got = handle.imu_preintegrate(intervals, biases, noise)
    for iv, (ba_, bg_), g in zip(intervals, biases, got):
        ref = B.Preintegration(iv[0][1], iv[0][2], ba_, bg_, *noise)
        for dt, a, w in iv[1:]:
            ref.push_back(dt, a, w)
        r = ref.as_dict()
        r2 = synth.preintegrate(iv, ba_, bg_, *noise)
        assert np.isclose(g['sum_dt'], r['sum_dt'], rtol=1e-14)
        for key in ('delta_p', 'delta_q', 'delta_v', 'jacobian', 'covariance'):
            scale = max(1e-300, np.abs(r[key]).max())
            assert np.abs(g[key] - r[key]).max() <= 1e-11 * scale, key
            assert np.abs(g[key] - r2[key]).max() <= 1e-11 * scale, key
        assert np.array_equal(g['lin_ba'], np.asarray(ba_)) and np.array_equal(g['lin_bg'], np.asarray(bg_))
    assert got[2]['sum_dt'] > 10.0
    # the device result feeds the BA unchanged: same window solved with device-side pre-integration
    seq = synth.SyntheticSequence(61, L=30)
    prob = seq.window(0)
    ref_state, _, _ = handle.ba_optimize(prob)
    c = seq.cfg
    ivs = []
    h = seq.frame_dt / seq.imu_per_frame
    for k in range(prob['pose'].shape[0] - 1):
        t = seq.times[k]
        ivs.append([(0.0,) + seq._imu_sample(t)] + [(h,) + seq._imu_sample(t + s * h) for s in range(1, seq.imu_per_frame + 1)])
    dev = handle.imu_preintegrate(ivs, [(seq.ba_lin, seq.bg_lin)] * len(ivs), (c['acc_n'], c['gyr_n'], c['acc_w'], c['gyr_w']))
    p2 = dict(prob)
    p2['imu'] = dev
    st2, sm2, _ = handle.ba_optimize(p2)
    assert sm2['status'] == 0 and np.abs(st2['pose'] - ref_state['pose']).max() < 1e-9


def test_ragged_batch_matches_single_windows(handle):
    """One launch with windows of different sizes / structure (L = 12, 60, 150; with and without prior; with and
    without marginalization): every window must equal its own single-window launch bit for bit, and the oracle."""
    _, _, wp = _window_with_prior(71, L=60)
    probs = [synth.SyntheticSequence(72, L=12).window(0), wp, synth.SyntheticSequence(73, L=150).window(0)]
    flags = [ba.VG_MARGIN_NONE, ba.VG_MARGIN_OLD, ba.VG_MARGIN_SECOND_NEW]
    handle.ba_upload(probs, flags)
    handle.ba_run_async()
    st, sm, pr = handle.ba_download()
    for i, p in enumerate(probs):
        s1, m1, p1 = handle.ba_optimize(p, flags[i])
        assert sm[i]['status'] == 0 and sm[i]['num_iterations'] == m1['num_iterations']
        assert np.array_equal(st[i]['pose'], s1['pose']) and np.array_equal(st[i]['sb'], s1['sb'])
        assert np.array_equal(st[i]['inv_depth'], s1['inv_depth'])
        assert (pr[i] is None) == (p1 is None)
        if p1 is not None:
            assert np.array_equal(pr[i]['J0'], p1['J0']) and np.array_equal(pr[i]['r0'], p1['r0'])
        x, summ = B.solve(p)
        ref = B.double2vector(p, x)
        assert np.abs(st[i]['pose'] - ref['pose']).max() < 1e-6
    assert pr[0] is None and pr[1] is not None


def test_throughput_and_latency_layouts_agree(handle):
    """The IMU / prior linearisation runs as one workgroup per window in a full batch and as (K-1)/2 + 1 workgroups per window
    while they all fit the chip at once (BaLayout::nig / nprw): a batch large enough for the first layout must give every window
    the result it gets alone (second layout).  The cost partials are summed in a different grouping, so equality is to
    rounding, not bitwise."""
    seqs = [synth.SyntheticSequence(900 + s, L=40) for s in range(48)]           # 48 * 6 > 256 workgroups: throughput layout
    probs = [q.window(0) for q in seqs]
    handle.ba_upload(probs, [ba.VG_MARGIN_OLD] * len(probs))
    handle.ba_run_async()
    st, sm, pr = handle.ba_download()
    for i in (0, 7, 23, 47):
        s1, m1, p1 = handle.ba_optimize(probs[i], ba.VG_MARGIN_OLD)              # latency layout
        assert sm[i]['status'] == 0 and sm[i]['num_iterations'] == m1['num_iterations']
        assert np.array_equal(sm[i]['it_flags'], m1['it_flags'])
        assert np.abs(st[i]['pose'] - s1['pose']).max() < 1e-9 and np.abs(st[i]['sb'] - s1['sb']).max() < 1e-9
        assert np.isclose(sm[i]['final_cost'], m1['final_cost'], rtol=1e-10)
        Ha, Hb = pr[i]['J0'].T @ pr[i]['J0'], p1['J0'].T @ p1['J0']
        assert np.abs(Ha - Hb).max() < 1e-7 * np.abs(Hb).max()


def test_error_behaviour(handle):
    """Status codes instead of exceptions on the data path: a non-finite input poisons only
    its own window (VG_ERR_NUMERIC, failureDetection() territory); structural errors are refused up front."""
    good = synth.SyntheticSequence(74, L=30).window(0)
    bad = dict(good)
    bad['pose'] = good['pose'].copy()
    bad['pose'][3, 0] = np.nan
    handle.ba_upload([good, bad, good], [ba.VG_MARGIN_OLD] * 3)
    handle.ba_run_async()
    st, sm, pr = handle.ba_download(allow_numeric_failure=True)
    assert sm[0]['status'] == 0 and sm[2]['status'] == 0 and sm[1]['status'] == -4
    assert pr[1] is None and pr[0] is not None
    assert np.array_equal(st[0]['pose'], st[2]['pose'])
    with pytest.raises(RuntimeError, match="status -4"):
        handle.ba_upload([bad])
        handle.ba_run_async()
        handle.ba_download()
    # more frames than the single-workgroup pipeline holds in LDS: taken by the large-window path (tests/test_ba_large_gpu.py,
    # incl. its marginalization); beyond BA_MAX_K_LARGE frames: refused outright
    def stretched(extra):
        big = dict(good)
        big['pose'] = np.vstack([good['pose']] + [good['pose'][-1:]] * extra)
        big['sb'] = np.vstack([good['sb']] + [good['sb'][-1:]] * extra)
        big['imu'] = list(good['imu']) + [good['imu'][-1]] * extra
        return big
    handle.ba_upload([stretched(3)], [ba.VG_MARGIN_OLD])
    with pytest.raises(RuntimeError, match="status -3"):
        handle.ba_upload([stretched(30)])
    # inconsistent tables
    broken = dict(good)
    broken['lm_nobs'] = good['lm_nobs'].copy()
    broken['lm_nobs'][0] = 1                     # a landmark needs >= 2 observations (estimator.cpp:723)
    with pytest.raises(RuntimeError, match="status -1"):
        handle.ba_upload([broken])


def test_triangulate_on_device(handle):
    """SURVEY 8(f) row 4 (first half): FeatureManager::triangulate on the device vs the NumPy restatement (LAPACK SVD)
    on a synthetic window: true depths recovered, parity of the DLT null vector ratio, INIT_DEPTH substitution."""
    seq = synth.SyntheticSequence(81, L=150)
    prob = seq.window(0)
    c = seq.cfg
    K = prob['pose'].shape[0]
    Ps = prob['pose'][:, :3]
    Rs = np.array([B.q2R(q) for q in prob['pose'][:, 3:]])
    start, nobs, off = prob['lm_start'], prob['lm_nobs'], prob['obs_off']
    pts = np.concatenate([prob['obs'][:, :2], np.ones((len(prob['obs']), 1))], axis=1)      # (x, y, 1) normalised points
    ref = B.triangulate(Ps, Rs.reshape(K, 9), c['tic'], c['ric'], start, nobs, off, pts)
    got = handle.triangulate(Ps, Rs.reshape(K, 9), c['tic'], c['ric'], start, nobs, off, pts)
    assert got.shape == ref.shape
    assert np.allclose(got, ref, rtol=1e-8, atol=1e-10)
    # noisy poses / pixels, yet the depths are close to the window's (also noisy) inverse depths for long tracks
    lng = nobs >= 6
    assert np.median(np.abs(got[lng] * prob['inv_depth'][lng] - 1.0)) < 0.25
    # a point behind the camera / at infinity -> depth < 0.1 -> INIT_DEPTH (feature_manager.cpp:251-254)
    pts2 = pts.copy()
    pts2[off[0]:off[0] + nobs[0], :2] = pts2[off[0], :2]                  # zero parallax: rank-deficient system
    r2 = B.triangulate(Ps, Rs.reshape(K, 9), c['tic'], c['ric'], start[:1], nobs[:1], off[:1], pts2, 7.5)
    g2 = handle.triangulate(Ps, Rs.reshape(K, 9), c['tic'], c['ric'], start[:1], nobs[:1], off[:1], pts2, 7.5)
    assert (r2[0] == 7.5) == (g2[0] == 7.5)
    with pytest.raises(RuntimeError, match="status -1"):
        handle.triangulate(Ps, Rs.reshape(K, 9), c['tic'], c['ric'], [K - 1], [3], [0], pts)


def test_solver_time_cap(handle):
    """SOLVER_TIME (estimator.cpp:812-815, options.max_solver_time_in_seconds): the cap is tested on the device clock before every
    iteration.  A cap far above the solve changes nothing; an expired cap ends the solve NO_CONVERGENCE at the current point."""
    prob = synth.SyntheticSequence(12, L=40).window(0)
    st0, sm0, _ = handle.ba_optimize(prob)
    st1, sm1, _ = handle.ba_optimize(dict(prob, max_solver_time_s=10.0))
    assert sm1['num_iterations'] == sm0['num_iterations'] and np.array_equal(st0['pose'], st1['pose'])
    st2, sm2, _ = handle.ba_optimize(dict(prob, max_solver_time_s=1e-9))
    assert sm2['status'] == 0 and sm2['termination'] == sm0['termination'] == 0      # NO_CONVERGENCE, as after max_iters
    assert sm2['num_iterations'] == 0 and sm2['final_cost'] == sm2['initial_cost'] == sm0['initial_cost']
    ref = B.double2vector(prob, dict(pose=prob['pose'], sb=prob['sb'], ex=prob['ex'], td=prob['td'], inv_depth=prob['inv_depth']))
    assert np.abs(st2['pose'] - ref['pose']).max() < 1e-12


def test_state_download_overtakes_the_marginalization(handle):
    """vg_ba_batch_download_state / _prior: the states come back while the marginalization kernel is still queued or running
    (second stream behind an event recorded after the solve pipeline); both parts equal the one-call download bit for bit."""
    probs = [_window_with_prior(60 + k, L=40)[2] for k in range(3)]
    handle.ba_upload(probs, [ba.VG_MARGIN_OLD] * 3)
    handle.ba_run_async()
    st0, sm0, pr0 = handle.ba_download()
    handle.ba_run_async()
    outs, st, pri, sm, packed = handle.ba_prepare_download()
    assert handle.ba_download_state_raw() == 0
    st1 = [o.state_dict(p.has_relo) for o, p in zip(outs, packed)]
    assert handle.ba_download_prior_raw() == 0
    pr1 = [o.prior_dict() for o in outs]
    for a, b in zip(st0, st1):
        assert np.array_equal(a['pose'], b['pose']) and np.array_equal(a['sb'], b['sb']) and np.array_equal(a['inv_depth'], b['inv_depth'])
    for a, b in zip(pr0, pr1):
        assert a['n'] == b['n'] and np.array_equal(a['J0'], b['J0']) and np.array_equal(a['r0'], b['r0'])
    s, m, p, t_ms = handle.ba_optimize_split(probs[0], ba.VG_MARGIN_OLD)
    assert np.array_equal(s['pose'], st0[0]['pose']) and np.array_equal(p['J0'], pr0[0]['J0']) and t_ms > 0


def test_handles_are_independent_across_host_threads():
    """The C-ABI is re-entrant across handles (one host thread per handle is how the boundary reaches
    0.9 x the device-resident rate): four threads, each with its own handle, solve and marginalize different batches at the
    same time, a few times over — every result must be bit-identical to the same batch solved alone."""
    import threading
    nthr, reps = 4, 3
    batches = [[synth.SyntheticSequence(300 + 10 * t + k, L=40 + 5 * t).window(0) for k in range(6)] for t in range(nthr)]
    flags = [ba.VG_MARGIN_OLD] * 6
    ref = []
    h0 = ba.Handle()
    for b in batches:
        h0.ba_upload(b, flags)
        h0.ba_run_async()
        ref.append(h0.ba_download())
    h0.close()
    handles = [ba.Handle() for _ in range(nthr)]
    out, errs = [None] * nthr, []

    def work(t):
        try:
            res = []
            for _ in range(reps):
                handles[t].ba_upload(batches[t], flags)
                handles[t].ba_run_async()
                res.append(handles[t].ba_download())
            out[t] = res
        except Exception as ex:                       # noqa: BLE001
            errs.append(repr(ex))
    ts = [threading.Thread(target=work, args=(t,)) for t in range(nthr)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for hh in handles:
        hh.close()
    assert not errs, errs
    for t in range(nthr):
        st_r, sm_r, pr_r = ref[t]
        for st, sm, pr in out[t]:
            for i in range(6):
                assert sm[i]['status'] == 0 and sm[i]['num_iterations'] == sm_r[i]['num_iterations']
                assert _same_state(st[i], st_r[i]) and _same_prior(pr[i], pr_r[i]), (t, i)
