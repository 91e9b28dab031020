"""MANUAL (not collected): how often do the reference's loop and the drop-in take different trust-region decisions, and what does it
cost?  (quantify the allowance of tests/test_dropin_gpu.py::_compare.)

For `n` seeds the reference's own processIMU / processImage loop runs a 24-frame synthetic sequence (14 solves each) twice: with its
own Ceres-style optimization() (oracle/_ref/libvins_ref.so, CPU) and with the product's drop-in (libvins_ref_gpu.so), the latter in
both forms of the prior factor (pivoted-Cholesky square root = default, the reference's eigen form = vins_gpu_set_option(e, 2, 1)) and
(round 6) with the IMU factors' sqrt_info formed as imu_factor.h:64 spells it, inverse() then LLT
(vins_gpu_set_option(e, 3, 1)), alone and together with the eigen prior.  Per frame: same iteration count and accept / reject
sequence?  state error (relative position / quaternion / velocity / biases, max).
    python tests/manual/gpu_flip_stats.py [n_seeds] [n_frames] > profiles/<tag>_flip_stats.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import conftest  # noqa: E402,F401
from oracle import ref as R  # noqa: E402
from vins_mono_amd import synth  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_run = int(sys.argv[2]) if len(sys.argv) > 2 else 24          # frames fed to the loop (10 fill the window: n_run - 10 solves)


def rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def frame_err(r, g):
    return max(rel(g['pose'][:, :3], r['pose'][:, :3]), float(np.abs(g['pose'][:, 3:] - r['pose'][:, 3:]).max()), rel(g['sb'][:, :3], r['sb'][:, :3]),
               float(np.abs(g['sb'][:, 3:] - r['sb'][:, 3:]).max()))


out = {}
ref_runs = {}                                                # the reference's own loop: once per seed
for mode in ("sqrt", "eigen", "imu_reference", "eigen+imu_reference"):
    gpu_options = {"sqrt": None, "eigen": {2: 1}, "imu_reference": {3: 1}, "eigen+imu_reference": {2: 1, 3: 1}}[mode]      # vins_gpu_set_option
    n_frames = n_flip_frames = n_seq_with_flip = n_bookkeeping_diff = 0
    err_same, err_flip, err_after = [], [], []          # no flip so far in the sequence / the frame of a flip / the frames after one
    for seed in range(n_seeds):
        mp = 10.0 / 460.0 if seed % 2 == 0 else 0.1     # every other sequence also takes MARGIN_SECOND_NEW
        if seed not in ref_runs:
            ref_runs[seed] = R.run_sequence(synth.SyntheticSequence(1000 + seed, n_frames=n_run + 2, K=n_run + 2, L=500), n_run, L=R.lib(), min_parallax=mp, collect_priors=False)
        a = ref_runs[seed]
        b = R.run_sequence(synth.SyntheticSequence(1000 + seed, n_frames=n_run + 2, K=n_run + 2, L=500), n_run, L=R.lib_gpu(), min_parallax=mp, collect_priors=False, gpu_options=gpu_options)
        flipped = False
        for r, g in zip(a, b):
            n_frames += 1
            if r['flag'] != g['flag'] or set(r['depth']) != set(g['depth']):
                n_bookkeeping_diff += 1
            same = r['trace'].shape == g['trace'].shape and np.array_equal(r['trace'][:, :2], g['trace'][:, :2])
            e = frame_err(r, g)
            if not same:
                n_flip_frames += 1
                err_flip.append(e)
                flipped = True
            elif flipped:
                err_after.append(e)
            else:
                err_same.append(e)
        n_seq_with_flip += int(flipped)

    def dist(v):
        v = np.array(v) if len(v) else np.zeros(1)
        return {"n": len(v), "median": float(np.median(v)), "p99": float(np.percentile(v, 99)), "max": float(v.max())}
    out[mode] = {"sequences": n_seeds, "frames": n_frames, "frames_with_a_different_decision": n_flip_frames, "sequences_with_one": n_seq_with_flip,
                 "frames_with_different_keyframe_flag_or_tracks": n_bookkeeping_diff, "error_before_any_flip": dist(err_same),
                 "error_in_the_frame_of_a_flip": dist(err_flip), "error_in_later_frames_of_such_a_sequence": dist(err_after)}
out["what"] = (f"reference loop (Ceres-style optimization() of oracle/_ref) vs the drop-in on the GPU, {n_run}-frame synthetic sequences, {n_run - 10} solves each; "
               "a 'flip' = a frame whose solve took a different number of iterations or a different accept / reject sequence; errors = max of "
               "relative position, quaternion, relative velocity, bias differences over the window")
print(json.dumps(out, indent=1))
