"""Solve fixtures that force every branch of the restated Ceres trust-region loop (Ceres
semantics C1-C8 of oracle/ASSUMPTIONS.md).  Shared by the oracle-level tests (CPU) and the GPU parity tests, so that
"the GPU follows the oracle through a rejected step / a Cauchy step / ..." is asserted on fixtures whose oracle trace
is asserted to CONTAIN that branch.

Each builder returns a `prob` dict (see vins_mono_amd/synth.py).  `BRANCH_FIXTURES` maps a name to (builder, the
set of trace features its oracle run must show).  Features:
    gn / cauchy / dogleg      the three arms of DoglegStrategy::ComputeTraditionalDoglegStep
    rejected                  rho <= min_relative_decrease: radius halved, Gauss-Newton step + gradient reused
    mu_escalation             linear solve failed, mu *= 10 and retry (DoglegStrategy::ComputeGaussNewtonStep)
    invalid                   no usable step: counted, mu *= 10, re-linearised
    failure                   5 consecutive invalid steps -> termination FAILURE
    function_tolerance        |cost change| <= 1e-6 cost
    parameter_tolerance       |step| <= 1e-8 (|x| + 1e-8)
"""
import copy

import numpy as np

from oracle import ba_numpy as B
from vins_mono_amd import synth


def perturbed(prob, seed, p=1.0, theta_deg=10.0, lam_factor=5.0, v=0.0):
    """Large state perturbation: positions N(0, p) m, rotations N(0, theta) per axis, velocities N(0, v), inverse
    depths x lognormal(log lam_factor)."""
    rng = np.random.default_rng(seed)
    q = copy.deepcopy(prob)
    for i in range(q['pose'].shape[0]):
        q['pose'][i][:3] += rng.normal(0, p, 3)
        d = rng.normal(0, np.deg2rad(theta_deg), 3)
        q['pose'][i] = B.pose_plus(q['pose'][i], np.concatenate([np.zeros(3), d]))
        q['sb'][i][:3] += rng.normal(0, v, 3)
    q['inv_depth'] = q['inv_depth'] * np.exp(rng.normal(0, np.log(lam_factor), q['inv_depth'].shape[0]))
    return q


def fx_large_perturbation(L=40):
    """p 1 m, theta 10 deg, lambda x5, 32 iterations: Cauchy steps first (radius 1e4 is far inside the GN step), GN
    steps, then a run of rejected GN steps (radius halved five times with the step reused) and interpolated steps."""
    q = perturbed(synth.SyntheticSequence(2, L=L).window(0), 102)
    q['max_iters'] = 32
    return q


def fx_very_large_perturbation(L=40):
    q = perturbed(synth.SyntheticSequence(2, L=L).window(0), 102, p=3.0, theta_deg=20.0)
    q['max_iters'] = 32
    return q


def fx_vision_only(L=40):
    """No IMU factor is valid (sum_dt > 10 s rule, estimator.cpp:714) and there is no prior: speed-bias columns are
    empty, the 7-dof gauge is held by the mu D^2 damping alone; converges by the function tolerance."""
    q = synth.SyntheticSequence(1, L=L).window(0)
    q['imu'] = [None] * (q['pose'].shape[0] - 1)
    q['max_iters'] = 32
    return q


def fx_two_imu_missing(L=40):
    q = synth.SyntheticSequence(1, L=L).window(0)
    q['imu'][3] = None
    q['imu'][4] = None
    q['max_iters'] = 16
    return q


def fx_far_origin(L=40):
    """World origin 1e6 m away: |x| ~ 6e6, so the parameter tolerance 1e-8 |x| is 6 cm and fires on the third step."""
    q = synth.SyntheticSequence(1, L=L).window(0)
    q['pose'][:, :3] += 1e6
    q['max_iters'] = 32
    return q


def fx_overflowing_landmark(L=40):
    """One inverse depth of 1e-170: the residual stays finite but lambda^2 underflows, so the landmark Jacobian is
    +-inf and every linear solve fails: mu is escalated 1e-8 -> 1 inside the first step, then five invalid steps ->
    FAILURE with the state unchanged.  (Real Ceres rejects the non-finite Jacobian at evaluation time and also leaves
    the state unchanged; this fixture exists to walk the mu / invalid-step / FAILURE code.)"""
    q = synth.SyntheticSequence(1, L=L).window(0)
    q['inv_depth'][3] = 1e-170
    return q


def fx_lambda_near_zero(L=40):
    """lambda -> 0+ on one landmark and x50 on another (verdict item 1a): stays finite, first step interpolated."""
    q = synth.SyntheticSequence(1, L=L).window(0)
    q['inv_depth'][3] = 1e-6
    q['inv_depth'][7] *= 50
    q['max_iters'] = 16
    return q


def fx_single_landmark():
    """Near-degenerate window: one landmark only (IMU factors carry the solve)."""
    q = synth.SyntheticSequence(6, L=40).window(0)
    keep = int(np.argmax(q['lm_nobs']))
    o, n = int(q['obs_off'][keep]), int(q['lm_nobs'][keep])
    q['obs'] = q['obs'][o:o + n].copy()
    q['obs_off'] = np.array([0], np.int32)
    q['lm_start'] = q['lm_start'][keep:keep + 1].copy()
    q['lm_nobs'] = q['lm_nobs'][keep:keep + 1].copy()
    q['inv_depth'] = q['inv_depth'][keep:keep + 1].copy()
    q['max_iters'] = 16
    return q


def fx_low_parallax(L=40, shrink=0.01):
    """Near-zero parallax: the trajectory's translation is shrunk to 1 % about frame 0 (centimetre baselines against
    2-12 m depths, i.e. parallax at the pixel-noise level), tracks re-drawn for it, no IMU factors: the depths are
    barely observable and the first steps are large, partly rejected ones."""
    seq = synth.SyntheticSequence(5, L=L)
    seq.P = seq.P[0] + shrink * (seq.P - seq.P[0])
    seq._make_landmarks()
    q = seq.window(0)
    q['imu'] = [None] * (q['pose'].shape[0] - 1)
    q['max_iters'] = 16
    return q


BRANCH_FIXTURES = {
    'large_perturbation': (fx_large_perturbation, {'cauchy', 'gn', 'dogleg', 'rejected'}),
    'very_large_perturbation': (fx_very_large_perturbation, {'cauchy', 'dogleg', 'rejected'}),
    'vision_only': (fx_vision_only, {'function_tolerance'}),
    'two_imu_missing': (fx_two_imu_missing, set()),
    'far_origin': (fx_far_origin, {'parameter_tolerance'}),
    'overflowing_landmark': (fx_overflowing_landmark, {'mu_escalation', 'invalid', 'failure'}),
    'lambda_near_zero': (fx_lambda_near_zero, {'dogleg'}),
    'single_landmark': (fx_single_landmark, set()),
    'low_parallax': (fx_low_parallax, {'rejected', 'dogleg', 'gn'}),
}


# relative tolerance of the per-iteration cost comparison GPU vs oracle (default 1e-6): the low-parallax window is
# ill-conditioned by construction (depths barely observable), its candidate costs agree to ~1e-6 between any two
# implementations that sum in a different order
COST_RTOL = {'low_parallax': 1e-4}


def trace_features(summary):
    """The set of trust-region branches an oracle run went through (from ba_numpy.solve's per-iteration records)."""
    f = set()
    for it in summary['iterations']:
        if it.get('branch'):
            f.add(it['branch'])
        if it.get('mu_tries'):
            f.add('mu_escalation')
        if not it.get('valid'):
            f.add('invalid')
        elif 'rho' in it and not it.get('accepted'):
            f.add('rejected')
        if it.get('exit'):
            f.add(it['exit'])
    if summary['termination'] == 'FAILURE':
        f.add('failure')
    return f
