"""GPU parity tests of the BA path: HIP (through the C-ABI) vs the NumPy oracle on identical inputs.
Tolerances: factor level 1e-9 relative (IMU rows: 1e-6 of the row weight, the information matrix is
ill-conditioned and the device factorises the covariance instead of inverting it); solve level: identical
accept/reject sequence, costs 1e-6 relative, states 1e-4 relative (BASELINE.json north_star)."""
import numpy as np
import pytest

from oracle import ba_numpy as B
from vins_mono_amd import ba, synth

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


def _check_solve(handle, prob, rtol_state=1e-4):
    x, summ = B.solve(prob)
    ref = B.double2vector(prob, x)
    st, sm, _ = handle.ba_optimize(prob)
    assert sm['status'] == 0
    its = summ['iterations']
    assert sm['num_iterations'] == summ['num_iterations']
    flags = [(1 if it.get('valid') else 0) | (2 if it.get('accepted') else 0) for it in its]
    assert list(sm['it_flags']) == flags
    assert np.isclose(sm['initial_cost'], summ['initial_cost'], rtol=1e-9)
    for k, it in enumerate(its):
        if it.get('valid'):
            assert np.isclose(sm['it_cost_cand'][k], it['cost_cand'], rtol=1e-6, atol=1e-9), k
            assert np.isclose(sm['it_model'][k], it['model_change'], rtol=1e-5), k
            assert np.isclose(sm['it_radius'][k], it['radius'], rtol=1e-6), k
    assert np.isclose(sm['final_cost'], summ['final_cost'], rtol=1e-6)
    scale_p = max(1.0, np.abs(ref['pose'][:, :3]).max())
    assert np.abs(st['pose'][:, :3] - ref['pose'][:, :3]).max() < rtol_state * scale_p
    assert np.abs(st['pose'][:, 3:] - ref['pose'][:, 3:]).max() < rtol_state
    assert np.abs(st['sb'] - ref['sb']).max() < rtol_state * max(1.0, np.abs(ref['sb']).max())
    assert np.allclose(st['inv_depth'], ref['inv_depth'], rtol=1e-4, atol=1e-6)
    assert np.allclose(st['ex'], ref['ex'], atol=rtol_state)
    return st, sm


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
