"""Malformed inputs at the C-ABI: a caller's mistake must come back as a vg_status, never as an out-of-bounds access.  Random
corruptions of valid problems (sizes, offsets, indices, prior block tables, frame inputs of a sequence) are thrown at the emulated
library built with AddressSanitizer (tests/simt, `make asan`): every call must return a status and ASAN must stay silent; calls that
are accepted must solve without a sanitizer report.  Runs in a child process (a crash must not take pytest down)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIMT = os.path.join(ROOT, "tests", "simt")

_CHILD = r"""
import sys, ctypes as C
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
import numpy as np
import conftest
from vins_mono_amd import ba, synth
h = conftest._simt_handle()
rng = np.random.default_rng(7)
seq = synth.SyntheticSequence(3, L=30)
base = seq.anchor_prior(seq.window(0))
n_err = n_ok = 0
def mutate(prob, k):
    p = dict(prob)
    for key in ('lm_start', 'lm_nobs', 'obs_off', 'inv_depth', 'obs', 'pose', 'sb'):
        p[key] = np.array(p[key]).copy()
    pr = dict(p['prior']); pr['blocks'] = list(pr['blocks']); p['prior'] = pr
    L = len(p['lm_start'])
    l = int(rng.integers(0, L))
    if k == 0: p['lm_nobs'][l] = int(rng.choice([0, 1, -3, 40]))
    elif k == 1: p['lm_start'][l] = int(rng.choice([-1, 10, 11, 1000]))
    elif k == 2: p['obs_off'][l] = int(rng.choice([-5, 10**6, p['obs'].shape[0] - 1]))
    elif k == 3: pr['blocks'][int(rng.integers(0, len(pr['blocks'])))] = (int(rng.choice([-1, 4, 7])), 0)
    elif k == 4: pr['blocks'][0] = (0, int(rng.choice([-2, 11, 500])))
    elif k == 5: pr['blocks'] = pr['blocks'] + [pr['blocks'][0]]            # a block twice (n no longer matches)
    elif k == 6: p['max_iters'] = int(rng.choice([-1, 33, 10**6]))
    elif k == 7: pr['n'] = pr['n'] + int(rng.choice([-3, 5]))
    elif k == 8: p['relo'] = dict(pose=np.array([0, 0, 0, 0, 0, 0, 1.0]), match=[(int(rng.choice([-1, L, L + 7])), 0.1, 0.2)])
    elif k == 9: p['lm_nobs'][l] = 11 - int(p['lm_start'][l]) + 1           # one observation past the last frame
    return p
for it in range(60):
    p = mutate(base, it %% 10)
    try:
        st, sm, pr = h.ba_optimize(p, int(rng.integers(0, 3)))
        n_ok += 1
    except RuntimeError as e:
        assert "status -" in str(e), e
        n_err += 1
    except (AssertionError, ValueError, IndexError):      # (the Python binding itself may refuse: not the library's business)
        n_err += 1
# a valid call still works afterwards
st, sm, pr = h.ba_optimize(base, 0)
assert sm['status'] == 0
# non-finite and absurd values in a well-formed problem: a per-window status (VG_ERR_NUMERIC = -4) or a normal result, never a crash
for what in ("nan_pose", "inf_depth", "zero_quat", "nan_obs", "huge_prior"):
    p = {k: (np.array(v).copy() if isinstance(v, np.ndarray) else v) for k, v in base.items()}
    if what == "nan_pose": p['pose'][3, 0] = np.nan
    if what == "inf_depth": p['inv_depth'][2] = np.inf
    if what == "zero_quat": p['pose'][4, 3:] = 0
    if what == "nan_obs": p['obs'][5, 0] = np.nan
    if what == "huge_prior": p['prior'] = dict(p['prior'], J0=p['prior']['J0'] * 1e200)
    h.ba_upload([p], [0]); h.ba_run_async()
    st_, sm_, pr_ = h.ba_download(allow_numeric_failure=True)
    assert sm_[0]['status'] in (0, -4), (what, sm_[0]['status'])
    if what in ("nan_pose", "zero_quat", "nan_obs"):
        assert sm_[0]['status'] == -4 and pr_[0] is None, what      # no prior from a failed window
    n_ok += 1
# sequences: bad frame inputs
src = synth.FrameSource(synth.SyntheticSequence(21, n_frames=14, K=14, L=40), noise_seed=1)
prob, tracks = synth.sequence_inputs(src.initial_window(11, 0))
for bad in range(4):
    t = {k: np.array(v).copy() for k, v in tracks.items()}
    if bad == 0: t['nobs'][0] = 0
    if bad == 1: t['start'][1] = 10
    if bad == 2: t['start'][2] = -1
    if bad == 3: t['nobs'][3] = 30
    try:
        h.seq_begin([prob], [t], max_features=128, max_new_obs=128)
        raise SystemExit("a malformed track table was accepted: case %%d" %% bad)
    except RuntimeError as e:
        assert "status -" in str(e), e
        n_err += 1
h.seq_begin([prob], [tracks], max_features=128, max_new_obs=128)
ids, rows = src.image(10)
pose, sb = src.guess(10)
for bad in range(3):
    f = dict(pose=pose, sb=sb, imu_new=src.seq.imu[9], imu_merged=None, ids=ids.copy(), obs=rows.copy())
    if bad == 0: f['ids'][3] = f['ids'][2]                  # repeated id
    if bad == 1: f['ids'] = f['ids'][::-1].copy()
    if bad == 2: f['ids'] = np.concatenate([f['ids']] * 4); f['obs'] = np.concatenate([f['obs']] * 4)   # more than max_new_obs (and not ascending)
    try:
        h.seq_step([f])
        raise SystemExit("a malformed frame was accepted: case %%d" %% bad)
    except RuntimeError as e:
        assert "status -" in str(e), e
        n_err += 1
h.seq_step([dict(pose=pose, sb=sb, imu_new=src.seq.imu[9], imu_merged=None, ids=ids, obs=rows)])
assert h.seq_info()[0]['status'] == 0
h.seq_end()
# front end: refused calls and calls with hostile values (NaN / huge coordinates, degenerate correspondences) that must run clean
from vins_mono_amd import fe
W, H = 256, 160
a = synth.synth_frame(1, W, H); b = synth.warp_frame(a, 2)
tr = fe.FrontEnd(h, W, H, 1, 32)
tr.push_frames([a]); tr.push_frames([b])
pts = rng.uniform([0, 0], [W, H], (200, 2)).astype(np.float32)
intr = [461.6, 460.3, 363.0, 248.1, -0.2917, 0.08228, 5.333e-05, -1.578e-04]
for name, fn in (("track n > max_points", lambda: tr.track(0, pts)), ("track cam out of range", lambda: tr.track(3, pts[:10])),
                 ("detect more than max_points", lambda: tr.detect(0, 500)), ("detect tiny min_dist", lambda: tr.detect(0, 20, 0.01, 1.0)),
                 ("reject_with_f n < 8", lambda: tr.reject_with_f(pts[:5], pts[:5], 1.0))):
    try:
        fn()
        raise SystemExit("accepted: " + name)
    except RuntimeError as e:
        assert "status -" in str(e), e
        n_err += 1
nxt, st, err = tr.track(0, np.full((5, 2), np.nan, np.float32)); assert not st.any()
nxt, st, err = tr.track(0, np.full((5, 2), 1e30, np.float32)); assert not st.any()
tr.detect(0, 20, -1.0, 10.0)
tr.set_mask([pts[:20]], [np.ones(20, np.int32)], 0)
tr.undistort(np.full((5, 2), np.nan, np.float32), intr)
stat, Fm = tr.reject_with_f(np.zeros((20, 2), np.float32), np.zeros((20, 2), np.float32), 1.0)
assert stat.all()                                           # no model: nothing is rejected
n_ok += 6
# the one-call frame: refused argument sets, then hostile but legal ones (NaN points, all points identical, a callback that misbehaves)
for name, fn in (("read_image n > max_points", lambda: tr.read_image(b, pts[:100], True, intr, max_cnt=20)),
                 ("read_image max_cnt > max_points", lambda: tr.read_image(b, pts[:10], True, intr, max_cnt=500)),
                 ("read_image f_threshold 0", lambda: tr.read_image(b, pts[:10], True, intr, max_cnt=20, f_threshold=0.0)),
                 ("read_image order not a permutation", lambda: tr.read_image(b, pts[:30], True, intr, max_cnt=30, min_dist=10,
                                                                              order=lambda st, sf, fw, n2: np.zeros(n2, np.int32))),
                 ("read_image order of the wrong length", lambda: tr.read_image(b, pts[:30], True, intr, max_cnt=30, min_dist=10,
                                                                                order=lambda st, sf, fw, n2: np.arange(n2 + 1)))):
    try:
        fn()
        raise SystemExit("accepted: " + name)
    except RuntimeError as e:
        assert "status -" in str(e), e
        n_err += 1
o = tr.read_image(a, np.full((20, 2), np.nan, np.float32), True, intr, max_cnt=30, min_dist=10); assert o["n1"] == 0 and o["n_new"] > 0
o = tr.read_image(b, np.full((30, 2), 77.0, np.float32), True, intr, max_cnt=30, min_dist=10)       # 30 identical correspondences
assert o["n_kept"] <= 1
o = tr.read_image(a, np.zeros((0, 2), np.float32), False, intr, max_cnt=20); assert o["n_final"] == 0
n_ok += 3
print("OK refused", n_err, "accepted", n_ok)
"""


def test_malformed_inputs_come_back_as_status_codes():
    r0 = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    rt = r0.stdout.strip()
    if r0.returncode != 0 or not os.path.isabs(rt) or not os.path.exists(rt):
        pytest.skip("no libasan in this toolchain")
    # (the sanitizer build lives outside the repository, where tests/test_simt_asan.py puts it: the GPU pool refuses snapshots that
    #  carry -fsanitize=address objects)
    import tempfile
    asan_out = os.path.join(tempfile.gettempdir(), "vins_simt_build_asan_%d" % os.getuid())
    b = subprocess.run(["make", "-C", SIMT, "-j", str(os.cpu_count() or 4), "asan", "ASAN_OUT=" + asan_out], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout[-3000:] + b.stderr[-3000:]
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0",
               VINS_SIMT_LIB=os.path.join(asan_out, "libvinsgpu_simt.so"))
    r = subprocess.run([sys.executable, "-c", _CHILD % dict(root=ROOT)], capture_output=True, text=True, env=env, timeout=1500)
    assert "AddressSanitizer" not in r.stdout + r.stderr, (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0 and "OK refused" in r.stdout, (r.stdout + r.stderr)[-4000:]
