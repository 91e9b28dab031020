"""MANUAL (not collected): a longer end-to-end run of tests/e2e_vio.py through `vins_replay vio` -- both drop-ins in one process --
with the drift against the ground truth and the wall time per frame of FeatureTracker::readImage and of the estimator's solve.
    python tests/manual/e2e_long.py [n_frames] [--emulated]
On a GPU box it uses vins-mono_amd/lib/vins_replay (the stand-in for BASELINE configs[0]'s per-frame latency on one stream: the
reference's budget is the 50 ms frame interval of a 20 Hz camera); with --emulated the CPU emulator's binary (timings meaningless)."""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import conftest  # noqa: E402,F401  (package alias)
import e2e_vio as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
emulated = "--emulated" in sys.argv
exe = os.path.join(ROOT, "tests", "simt", "_build", "vins_replay_simt") if emulated else os.path.join(ROOT, "vins-mono_amd", "lib", "vins_replay")
scene = E.Scene(3, n)
frames = [scene.render(f) for f in range(n)]
with tempfile.TemporaryDirectory() as tmp:
    out = E.run_vio_replay(exe, scene, frames, tmp)
tm = E.run_vio_replay.timing
P = scene.seq.P
for o in out[::max(1, len(out) // 12)] + [out[-1]]:
    print("frame %3d  error %s m  |%.3f|  travelled %.2f m  flag %d  tracks %d  status %d" % (o[0], np.round(o[1] - P[o[0]], 3), np.linalg.norm(o[1] - P[o[0]]),
          np.linalg.norm(P[o[0]] - P[10]), o[2], o[3], o[4]))
print("readImage ms per frame: median %.3f  max %.3f | solve (processImage) ms: median %.3f  max %.3f" % (np.median(tm[:, 0]), tm[:, 0].max(), np.median(tm[:, 1]), tm[:, 1].max()))
