"""The stand-alone C++ classes of vins-mono_amd/host driven on the GPU: the Estimator round trip must equal a direct vg_ba_optimize
call (packing / filter / prior bookkeeping under test), the BA replay the oracle's CSV.  (The FeatureTracker class and the front-end
drop-in are held to the reference's own node in tests/test_fe_dropin.py, which replaced the Python mirror that used to live here.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from vins_mono_amd import ba, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "vins-mono_amd", "lib")


def test_estimator_shim_roundtrip_equals_direct_abi(handle):
    lib = C.CDLL(os.path.join(LIBDIR, "libvins_host.so"))
    seq = synth.SyntheticSequence(77, L=60)
    p1 = seq.window(0)
    st1, _, pr1 = handle.ba_optimize(p1, ba.VG_MARGIN_OLD)
    prob = seq.next_window(st1, pr1, 1)
    st, sm, pr = handle.ba_optimize(prob, ba.VG_MARGIN_OLD)
    pk = ba.PackedProblem(prob)
    K, L = pk.K, pk.L
    pose, sb, depth = np.zeros((K, 7)), np.zeros((K, 9)), np.zeros(L)
    flag, pn, pnb, iters = np.zeros(L, np.int32), C.c_int(), C.c_int(), C.c_int()
    kind, idx = np.zeros(32, np.int32), np.zeros(32, np.int32)
    J0, r0 = np.zeros(128 * 128), np.zeros(128)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    rc = lib.vins_host_estimator_roundtrip(C.byref(pk.struct), 0, pose.ctypes.data_as(dp), sb.ctypes.data_as(dp), depth.ctypes.data_as(dp),
                                           flag.ctypes.data_as(ip), C.byref(pn), C.byref(pnb), kind.ctypes.data_as(ip), idx.ctypes.data_as(ip),
                                           J0.ctypes.data_as(dp), r0.ctypes.data_as(dp), C.byref(iters))
    assert rc == 0 and iters.value == sm['num_iterations']
    # the shim goes para -> Eigen members (q -> R -> q) once more: agreement to rounding
    assert np.abs(pose - st["pose"]).max() < 1e-9 and np.abs(sb - st["sb"]).max() < 1e-9   # inputs differ by a q->R->q round trip
    assert np.allclose(1.0 / depth, st["inv_depth"], rtol=1e-9)
    assert np.all(flag == np.where(st['inv_depth'] < 0, 2, 1))
    assert pn.value == pr['n'] and [(int(kind[b]), int(idx[b])) for b in range(pnb.value)] == pr['blocks']
    n = pn.value
    Jg = J0[:n * n].reshape(n, n)
    assert np.abs(Jg.T @ Jg - pr['J0'].T @ pr['J0']).max() < 1e-6 * np.abs(pr['J0'].T @ pr['J0']).max()


def test_ba_replay_csv_matches_oracle(tmp_path):
    """SURVEY section 7 boundary test: an N-window replay through the drop-in Estimator (vins_replay ba: device IMU
    pre-integration, Estimator::optimization() = solve + MARGIN_OLD on the device, Estimator::slideWindow() with
    removeBackShiftDepth, prior carried through last_marginalization_info / ..._parameter_blocks) writes the result file
    of utility/visualization.cpp:157-172; the same replay through the NumPy oracle must give the same CSV to 1e-4."""
    import replay_util as R
    plan = R.make_plan(5, 5, L=50)
    seqf, outf = tmp_path / "seq.bin", tmp_path / "vins_result.csv"
    R.write_sequence(plan, str(seqf))
    r = subprocess.run([os.path.join(LIBDIR, "vins_replay"), "ba", str(seqf), str(outf)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = np.array([[float(v) for v in line.rstrip(",\n").split(",")] for line in open(outf)])
    ref = R.run_oracle(plan)
    assert got.shape == ref.shape == (5, 11)
    assert np.array_equal(got[:, 0], np.round(ref[:, 0]))                 # stamps [ns]
    assert np.abs(got[:, 1:4] - ref[:, 1:4]).max() < 1e-4 * max(1.0, np.abs(ref[:, 1:4]).max())
    assert np.abs(got[:, 4:8] - ref[:, 4:8]).max() < 1e-4
    assert np.abs(got[:, 8:11] - ref[:, 8:11]).max() < 1e-4 * max(1.0, np.abs(ref[:, 8:11]).max())
    # the windows really are chained: later rows move with the trajectory
    assert np.abs(got[-1, 1:4] - got[0, 1:4]).max() > 0.1


def test_estimator_shim_drops_the_prior_after_a_numeric_failure(handle):
    """A non-finite solve (VG_ERR_NUMERIC) must not leave the PRE-slide prior behind: after MARGIN_OLD it still names pose /
    speed-bias 0 and un-shifted frame indices and would be bound to the wrong frames by the next optimization()."""
    lib = C.CDLL(os.path.join(LIBDIR, "libvins_host.so"))
    seq = synth.SyntheticSequence(78, L=40)
    p1 = seq.window(0)
    st1, _, pr1 = handle.ba_optimize(p1, ba.VG_MARGIN_OLD)
    prob = seq.next_window(st1, pr1, 1)
    prob['sb'] = prob['sb'].copy()
    prob['sb'][4, 1] = np.nan
    pk = ba.PackedProblem(prob)
    K, L = pk.K, pk.L
    pose, sb, depth = np.zeros((K, 7)), np.zeros((K, 9)), np.zeros(L)
    flag, pn, pnb, iters = np.zeros(L, np.int32), C.c_int(-1), C.c_int(-1), C.c_int()
    kind, idx = np.zeros(32, np.int32), np.zeros(32, np.int32)
    J0, r0 = np.zeros(128 * 128), np.zeros(128)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    rc = lib.vins_host_estimator_roundtrip(C.byref(pk.struct), 0, pose.ctypes.data_as(dp), sb.ctypes.data_as(dp), depth.ctypes.data_as(dp),
                                           flag.ctypes.data_as(ip), C.byref(pn), C.byref(pnb), kind.ctypes.data_as(ip), idx.ctypes.data_as(ip),
                                           J0.ctypes.data_as(dp), r0.ctypes.data_as(dp), C.byref(iters))
    assert rc == 0
    assert pn.value == 0 and pnb.value == 0          # neither the old prior (it was set on entry) nor a new one


def test_estimator_shim_relocalisation_by_products(handle):
    """estimator.cpp:769-801 + :596-616 through the drop-in Estimator: match_points / relo_Pose as setReloFrame() leaves them ->
    the matched landmarks' factors reach the device, and double2vector() forms relo_relative_t / q / yaw and drift_correct_r / t
    from the gauge-fixed loop pose the device returns.  Expected values: the reference's expressions restated in NumPy on
    the direct-ABI result."""
    from test_ba_gpu import relocalisation_problem
    from oracle import ba_numpy as B
    lib = C.CDLL(os.path.join(LIBDIR, "libvins_host.so"))
    prob = relocalisation_problem(loop_frame=3)
    st, sm, _ = handle.ba_optimize(prob)
    assert sm['status'] == 0
    pk = ba.PackedProblem(prob)
    K = pk.K
    prev_t = np.array([0.3, -0.2, 0.1])
    yaw0 = np.deg2rad(12.0)
    prev_r = np.array([[np.cos(yaw0), -np.sin(yaw0), 0], [np.sin(yaw0), np.cos(yaw0), 0], [0, 0, 1.0]])
    pose, fixed, rt, rq, ryaw, dr, dt = np.zeros((K, 7)), np.zeros(7), np.zeros(3), np.zeros(4), C.c_double(), np.zeros(9), np.zeros(3)
    dp = C.POINTER(C.c_double)
    rc = lib.vins_host_estimator_relo_roundtrip(C.byref(pk.struct), 3, prev_t.ctypes.data_as(dp), np.ascontiguousarray(prev_r).ctypes.data_as(dp),
                                                pose.ctypes.data_as(dp), fixed.ctypes.data_as(dp), rt.ctypes.data_as(dp), rq.ctypes.data_as(dp),
                                                C.byref(ryaw), dr.ctypes.data_as(dp), dt.ctypes.data_as(dp))
    assert rc == 0
    assert np.abs(pose - st['pose']).max() < 1e-9 and np.abs(fixed - st['relo_pose']).max() < 1e-9

    def yaw_deg(R):
        return np.rad2deg(np.arctan2(R[1, 0], R[0, 0]))
    relo_r, relo_t = B.q2R(st['relo_pose'][3:]), st['relo_pose'][:3]
    R3, P3 = B.q2R(st['pose'][3][3:]), st['pose'][3][:3]
    assert np.allclose(rt, relo_r.T @ (P3 - relo_t), atol=1e-9)
    Rrel = B.q2R(rq)
    assert np.allclose(Rrel, relo_r.T @ R3, atol=1e-9)
    d = yaw_deg(R3) - yaw_deg(relo_r)
    d = d - 360.0 * np.floor((d + 180.0) / 360.0) if d > 0 else d + 360.0 * np.floor((-d + 180.0) / 360.0)
    assert abs(ryaw.value - d) < 1e-9
    dy = np.deg2rad(yaw_deg(prev_r) - yaw_deg(relo_r))
    Rd = np.array([[np.cos(dy), -np.sin(dy), 0], [np.sin(dy), np.cos(dy), 0], [0, 0, 1.0]])
    assert np.allclose(dr.reshape(3, 3), Rd, atol=1e-9) and np.allclose(dt, prev_t - Rd @ relo_t, atol=1e-9)


def test_slide_window_second_new_merges_the_imu_interval(handle):
    """estimator.cpp:1069-1099 through the drop-in Estimator::slideWindow(): after a non-keyframe the interval that ended at the
    dropped frame is folded into its predecessor.  Expected values: the REFERENCE's IntegrationBase fed the concatenated
    samples (push_back, integration_base.h:30-36), when oracle/_ref is present; the NumPy restatement otherwise."""
    from oracle import ba_numpy as B, ref as R
    lib = C.CDLL(os.path.join(LIBDIR, "libvins_host.so"))
    rng = np.random.default_rng(123)
    noise = np.array([0.08, 0.004, 4e-5, 2e-6])

    def interval(n):
        first = np.concatenate([rng.normal(0, 1, 3) + [0, 0, 9.8], rng.normal(0, 0.3, 3)])
        smp = np.array([[0.005, *(rng.normal(0, 1, 3) + [0, 0, 9.8]), *rng.normal(0, 0.3, 3)] for _ in range(n)])
        return first, smp
    fa, sa = interval(20)
    fb, sb_ = interval(13)
    bias = np.concatenate([rng.normal(0, 0.05, 3), rng.normal(0, 0.02, 3)])
    q = rng.normal(size=4)
    st_prev = np.concatenate([rng.normal(size=3), q / np.linalg.norm(q), rng.normal(size=9)])
    q = rng.normal(size=4)
    st_new = np.concatenate([rng.normal(size=3), q / np.linalg.norm(q), rng.normal(size=9)])
    merged, out, rel, ns = ba.ImuPreint(), np.zeros(16), C.c_int(), C.c_int()
    dp = C.POINTER(C.c_double)
    rc = lib.vins_host_slide_second_new(len(sa), np.ascontiguousarray(sa).ctypes.data_as(dp), fa.ctypes.data_as(dp), len(sb_), np.ascontiguousarray(sb_).ctypes.data_as(dp),
                                        fb.ctypes.data_as(dp), bias.ctypes.data_as(dp), noise.ctypes.data_as(dp), st_prev.ctypes.data_as(dp), st_new.ctypes.data_as(dp),
                                        C.byref(merged), out.ctypes.data_as(dp), C.byref(rel), C.byref(ns))
    assert rc == 0 and rel.value == 1 and ns.value == 33
    samples = [(0.0, fa[:3], fa[3:])] + [(r[0], r[1:4], r[4:7]) for r in sa] + [(r[0], r[1:4], r[4:7]) for r in sb_]
    if R.available():
        R.configure(noise[0], noise[2], noise[1], noise[3])
        ref = R.preintegrate(samples, bias[:3], bias[3:])
    else:
        ref = synth.preintegrate(samples, bias[:3], bias[3:], *noise)
    got = dict(delta_p=np.array(merged.delta_p), delta_q=np.array(merged.delta_q), delta_v=np.array(merged.delta_v),
               jacobian=np.array(merged.jacobian).reshape(15, 15), covariance=np.array(merged.covariance).reshape(15, 15))
    assert np.isclose(merged.sum_dt, ref['sum_dt'], rtol=1e-13)
    for key, val in got.items():
        assert np.abs(val - ref[key]).max() <= 1e-11 * np.abs(ref[key]).max(), key
    exp = st_new.copy()
    exp[3:7] = B.R2q(B.q2R(st_new[3:7]))
    assert np.allclose(out, exp, atol=1e-12)                      # frame WINDOW_SIZE moved into slot WINDOW_SIZE-1


def test_estimator_shim_relocalisation_without_a_matched_landmark(handle):
    """setReloFrame() happened but no feature of match_points is in the window any more: the reference still adds relo_Pose as a
    parameter block (estimator.cpp:771-772), the solve leaves it alone, and double2vector() gauge-fixes it for the by-products
    (:598-616).  Expected values: the REFERENCE's Estimator run on the same window (oracle/_ref) when it is present."""
    from test_ba_gpu import relocalisation_problem
    from oracle import ba_numpy as B, ref as R
    lib = C.CDLL(os.path.join(LIBDIR, "libvins_host.so"))
    prob = relocalisation_problem(loop_frame=3)
    pk = ba.PackedProblem(prob)
    K = pk.K
    prev_t = np.array([0.3, -0.2, 0.1])
    prev_r = B.ypr2R(np.array([12.0, 0, 0]))
    pose, fixed, rt, rq, ryaw, dr, dt = np.zeros((K, 7)), np.zeros(7), np.zeros(3), np.zeros(4), C.c_double(), np.zeros(9), np.zeros(3)
    dp = C.POINTER(C.c_double)
    rc = lib.vins_host_estimator_relo_nomatch_roundtrip(C.byref(pk.struct), 3, prev_t.ctypes.data_as(dp), np.ascontiguousarray(prev_r).ctypes.data_as(dp),
                                                        pose.ctypes.data_as(dp), fixed.ctypes.data_as(dp), rt.ctypes.data_as(dp), rq.ctypes.data_as(dp),
                                                        C.byref(ryaw), dr.ctypes.data_as(dp), dt.ctypes.data_as(dp))
    assert rc == 0
    plain = dict(prob, relo=None)                                   # what reaches the solver: the window without relo factors
    st, sm, _ = handle.ba_optimize(plain)
    assert np.abs(pose - st['pose']).max() < 1e-9
    # the transform the device reports reproduces its own gauge fix
    x, _ = B.solve(plain)
    assert np.allclose(sm['gauge_rot'] @ (x['pose'][5][:3] - sm['gauge_p0']) + prob['pose'][0][:3], st['pose'][5][:3], atol=1e-6)
    relo_r = sm['gauge_rot'] @ B.q2R(B.qnormalized(prob['relo']['pose'][3:]))
    relo_t = sm['gauge_rot'] @ (prob['relo']['pose'][:3] - sm['gauge_p0']) + st['pose'][0][:3]
    assert np.allclose(B.q2R(fixed[3:]), relo_r, atol=1e-9) and np.allclose(fixed[:3], relo_t, atol=1e-9)
    R3, P3 = B.q2R(st['pose'][3][3:]), st['pose'][3][:3]
    assert np.allclose(rt, relo_r.T @ (P3 - relo_t), atol=1e-9) and np.allclose(B.q2R(rq), relo_r.T @ R3, atol=1e-9)
    if R.available():
        ref_prob = dict(prob)
        ref_prob['relo'] = dict(prob['relo'], match=[(len(prob['inv_depth']) + 50 + k, 0.0, 0.0) for k in range(3)], local_index=3, prev_t=prev_t, prev_r=prev_r)
        st_r, _, _ = R.optimization(ref_prob, 1)
        assert np.allclose(rt, st_r['relo_relative_t'], atol=1e-6) and np.allclose(B.q2R(rq), B.q2R(st_r['relo_relative_q']), atol=1e-6)
        assert abs(ryaw.value - st_r['relo_relative_yaw']) < 1e-6
        assert np.allclose(dr.reshape(3, 3), st_r['drift_correct_r'], atol=1e-8) and np.allclose(dt, st_r['drift_correct_t'], atol=1e-6)
