"""MANUAL (not collected; CPU only): how often does the REFERENCE disagree WITH ITSELF about a trust-region decision when nothing but
the rounding of its arithmetic changes?  The yardstick for tests/manual/gpu_flip_stats.py.

The reference's own processIMU / processImage loop (oracle/_ref: the reference's translation units compiled unchanged) runs the same
synthetic sequences as gpu_flip_stats.py twice: on the build every test uses (-O2 -ffp-contract=off) and on a second legitimate
compilation of the same sources (`make -C oracle ref_fma`: -march=x86-64-v3 -ffp-contract=fast, i.e. fused multiply-adds).  Per
frame: same iteration count and accept / reject sequence?  state error (relative position / quaternion / velocity / biases, max).
    python tests/manual/cpu_ref_self_flip_stats.py [n_seeds] [n_frames] [workers] > profiles/<tag>_ref_self_flip_stats.json"""
import ctypes as C
import json
import os
import subprocess
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_run = int(sys.argv[2]) if len(sys.argv) > 2 else 24
workers = int(sys.argv[3]) if len(sys.argv) > 3 else max(1, (os.cpu_count() or 2) - 2)
FMA = os.path.join(ROOT, "oracle", "_ref", "fma", "libvins_ref.so")


def rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def frame_err(r, g):
    return max(rel(g['pose'][:, :3], r['pose'][:, :3]), float(np.abs(g['pose'][:, 3:] - r['pose'][:, 3:]).max()), rel(g['sb'][:, :3], r['sb'][:, :3]),
               float(np.abs(g['sb'][:, 3:] - r['sb'][:, 3:]).max()))


def one_seed(seed):
    import conftest  # noqa: F401
    from oracle import ref as R
    from vins_mono_amd import synth
    lib_fma = R._prepare(C.CDLL(FMA))
    mp = 10.0 / 460.0 if seed % 2 == 0 else 0.1
    a = R.run_sequence(synth.SyntheticSequence(1000 + seed, n_frames=n_run + 2, K=n_run + 2, L=500), n_run, L=R.lib(), min_parallax=mp, collect_priors=False)
    b = R.run_sequence(synth.SyntheticSequence(1000 + seed, n_frames=n_run + 2, K=n_run + 2, L=500), n_run, L=lib_fma, min_parallax=mp, collect_priors=False)
    rows, flipped = [], False
    for r, g in zip(a, b):
        same = r['trace'].shape == g['trace'].shape and np.array_equal(r['trace'][:, :2], g['trace'][:, :2])
        kind = "flip" if not same else ("after" if flipped else "same")
        flipped = flipped or not same
        rows.append((kind, frame_err(r, g), int(r['flag'] != g['flag'] or set(r['depth']) != set(g['depth']))))
    return rows


if __name__ == "__main__":
    if os.path.exists("/root/reference/vins_estimator/src/estimator.cpp"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "ref_fma"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with ProcessPoolExecutor(max_workers=workers) as ex:
        per_seed = list(ex.map(one_seed, range(n_seeds)))
    err = {"same": [], "flip": [], "after": []}
    n_frames = n_book = n_seq = 0
    for rows in per_seed:
        n_seq += int(any(k == "flip" for k, _, _ in rows))
        for k, e, bk in rows:
            n_frames += 1
            n_book += bk
            err[k].append(e)

    def dist(v):
        v = np.array(v) if len(v) else np.zeros(1)
        return {"n": len(v), "median": float(np.median(v)), "p99": float(np.percentile(v, 99)), "max": float(v.max())}
    out = {"reference_vs_reference_with_fma": {
        "sequences": n_seeds, "frames": n_frames, "frames_with_a_different_decision": len(err["flip"]), "sequences_with_one": n_seq,
        "frames_with_different_keyframe_flag_or_tracks": n_book, "error_before_any_flip": dist(err["same"]),
        "error_in_the_frame_of_a_flip": dist(err["flip"]), "error_in_later_frames_of_such_a_sequence": dist(err["after"])},
        "what": (f"the reference's loop (Ceres-style optimization() of oracle/_ref) against ITSELF compiled with fused multiply-adds "
                 f"(make -C oracle ref_fma), {n_run}-frame synthetic sequences, {n_run - 10} solves each, the seeds of gpu_flip_stats.py; same "
                 "definitions of 'flip' and of the errors")}
    print(json.dumps(out, indent=1))
