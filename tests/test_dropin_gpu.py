"""The drop-in inside the REFERENCE'S OWN CLASS, driven at the level of the ROS callbacks.

oracle/_ref/libvins_ref_gpu.so is the reference's estimator.cpp / feature_manager.cpp / factor/* compiled unchanged (as in
libvins_ref.so) with ONE symbol replaced: `Estimator::optimization()` is the product's body
(vins-mono_amd/host/dropin/estimator_optimization.cpp, written against the reference's estimator.h and calling the C-ABI of
libvinsgpu.so).  Both libraries run the same synthetic sequence through `Estimator::processIMU` / `Estimator::processImage`
(estimator.cpp:81-215): feature bookkeeping and key-frame selection by parallax (feature_manager.cpp:45-107), triangulation,
optimization(), failureDetection(), slideWindow() for both marginalization flags — with the IMU buffers of a dropped
non-keyframe merged into the previous interval (:1069-1099) —, removeFailures().  Nothing around optimization() is restated:
what differs between the two runs is exactly the code this repo replaces (Ceres solve + MarginalizationInfo on the CPU side).

Acceptance = BASELINE.json north_star: identical decisions (key-frame flags, feature bookkeeping), states within 1e-4 relative."""
import numpy as np
import pytest

from oracle import ref as R
from vins_mono_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not (R.available() and R.gpu_available()), reason="oracle/_ref libraries are not built")]


def _rel(a, b):
    return np.abs(a - b).max() / max(1.0, np.abs(b).max())


def _compare(ref, got, tol=1e-4, tol_after_flip=1.5e-3):
    """Frame by frame.  As long as both runs have taken the same trust-region decisions (number of iterations, accepted /
    rejected steps) in every frame so far they must agree to `tol` (north_star: 1e-4 relative).  Those decisions are thresholds
    (function tolerance 1e-6, step quality rho > 1e-3, radius updates at 0.25 / 0.75): two implementations that agree to 1e-9
    can still land on different sides of one, and then differ by the size of a step (~1e-4 .. 1e-3) until the following frames
    pull the two runs together again.  Measured over 100 sequences x 14 solves (profiles/r04g_flip_stats.json, round 4): 40 of 1400
    frames take a different decision (2.9 %, in 23 of the 100 sequences; the same with the reference's eigen form of the prior, 39 —
    the square-root form is not the cause); error before any flip <= 7.5e-5 (median 7e-7), in the frame of a flip <= 9.2e-4 (median
    1.2e-4), in the later frames of such a sequence <= 8.8e-4.  The frame of a flip and the two after it are therefore compared at
    `tol_after_flip` = 1.5e-3 (round 3: 2e-3, unmeasured); at most a quarter of the frames may be flips (measured worst: 3 of 14)."""
    assert len(ref) == len(got)
    worst, loose_left, n_flips, rows, checks = 0.0, 0, 0, [], []
    for r, g in zip(ref, got):
        assert r['frame'] == g['frame'] and r['solver_flag'] == g['solver_flag'] == 1      # no failure-detection reboot on either side
        assert r['flag'] == g['flag'], r['frame']                                            # same key-frame decision
        assert r['n_features'] == g['n_features'] and set(r['depth']) == set(g['depth'])      # same tracks survive
        parts = (_rel(g['pose'][:, :3], r['pose'][:, :3]), np.abs(g['pose'][:, 3:] - r['pose'][:, 3:]).max(), _rel(g['sb'][:, :3], r['sb'][:, :3]),
                 np.abs(g['sb'][:, 3:] - r['sb'][:, 3:]).max(), np.abs(g['ex'] - r['ex']).max())
        e = max(parts)
        same = r['trace'].shape == g['trace'].shape and np.array_equal(r['trace'][:, :2], g['trace'][:, :2])
        if not same:                             # a different number of iterations or a different accept / reject sequence
            n_flips += 1
            loose_left = 3
        flipped = loose_left > 0
        loose_left = max(0, loose_left - 1)
        worst = max(worst, e)
        ids = [i for i in r['depth'] if r['depth'][i] > 0 and g['depth'][i] > 0]
        dr, dg = np.array([r['depth'][i] for i in ids]), np.array([g['depth'][i] for i in ids])
        dmed = float(np.median(np.abs(dg / dr - 1.0)))
        rows.append((r['frame'], r['flag'], r['iterations'], g['iterations'], bool(same), ["%.1e" % v for v in parts], "%.1e" % dmed,
                     "%.3e %.3e" % (r['trace'][-1][2], g['trace'][-1][2])))
        checks.append((e, dmed, flipped))
        assert (r['prior'] is None) == (g['prior'] is None)
        if r['prior']:                           # (False = the run was asked not to fetch priors between frames)
            assert sorted(r['prior']['blocks']) == sorted(g['prior']['blocks']) and r['prior']['n'] == g['prior']['n']
    print("frame, flag, iterations (reference, drop-in), same decisions, [position, quaternion, velocity, biases, extrinsic] differences, depth, last cost:")
    for row in rows:
        print("  ", row)
    for (e, dmed, fl), row in zip(checks, rows):
        assert e < (tol_after_flip if fl else tol), row
        assert dmed < (tol_after_flip if fl else tol), row
    assert n_flips <= len(ref) // 4, rows
    return worst


@pytest.mark.parametrize("min_parallax,expect_second_new", [(10.0 / 460.0, False), (0.1, True)])
def test_reference_loop_with_the_drop_in_optimization(min_parallax, expect_second_new):
    seq_a = synth.SyntheticSequence(11, n_frames=26, K=26, L=500)
    seq_b = synth.SyntheticSequence(11, n_frames=26, K=26, L=500)
    ref = R.run_sequence(seq_a, 24, L=R.lib(), min_parallax=min_parallax)
    got = R.run_sequence(seq_b, 24, L=R.lib_gpu(), min_parallax=min_parallax)
    flags = [r['flag'] for r in ref]
    assert (1 in flags) == expect_second_new and 0 in flags
    worst = _compare(ref, got)
    # the estimator tracks the simulated trajectory (both sides): the comparison above is not between two diverged runs
    for r in ref[3:]:
        assert np.abs(r['pose'][9][:3] - seq_a.P[r['frame']]).max() < 0.15
    print("worst relative state difference over the sequence:", worst)


def test_drop_in_with_td_estimation():
    """ESTIMATE_TD = 1: ProjectionTdFactor + the td block (estimator.cpp:694-698, :741-746), td carried through the prior.
    (Extrinsic estimation is covered at window level, tests/test_ref_parity.py: on this short synthetic sequence without a
    calibration prior the camera-IMU translation is unobservable and drifts by a metre in BOTH builds — not a comparison.)"""
    seq_a = synth.SyntheticSequence(12, n_frames=20, K=20, L=400, estimate_td=1)
    seq_b = synth.SyntheticSequence(12, n_frames=20, K=20, L=400, estimate_td=1)
    ref = R.run_sequence(seq_a, 18, L=R.lib())
    got = R.run_sequence(seq_b, 18, L=R.lib_gpu())
    _compare(ref, got)
    assert all(abs(r['td'] - g['td']) < 1e-5 for r, g in zip(ref, got))


def test_clear_state_between_two_optimizations_drops_the_pending_prior():
    """optimization() leaves its marginalization result on the device; the reference's unchanged clearState()
    (failureDetection() in processImage, estimator.cpp:190-200, or the restart callback) knows nothing of it.  The first optimization()
    of the re-initialised estimator must not install that result as last_marginalization_info — the reference has no prior there.
    Without the continuity check in the drop-in the states of frame 23 are off by metres."""
    ref = R.run_sequence(synth.SyntheticSequence(11, n_frames=28, K=28, L=300), 26, L=R.lib(), reset_at=13, collect_priors=False)
    got = R.run_sequence(synth.SyntheticSequence(11, n_frames=28, K=28, L=300), 26, L=R.lib_gpu(), reset_at=13, collect_priors=False)
    assert [r['frame'] for r in ref] == [10, 11, 12, 23, 24, 25]
    _compare(ref, got)


def test_solver_time_cap_is_a_runtime_switch_of_the_drop_in():
    from test_dropin_simt import _solver_time_cap_switch
    _solver_time_cap_switch(R.lib_gpu())
