"""Phase profile of ba_solve_kernel inside a FULL batch (256 windows, one workgroup per CU): cycles of thread 0 per phase and round,
for a few windows of the batch.  Library built with -DBA_PROFILE (see gpu_phase_profile.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libvinsgpu_prof.so")
from vins_mono_amd import ba, synth
import bench
h = ba.Handle()
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
probs, seqs = bench.make_windows(h, ba, synth, nwin, seed0=1)
flags = [ba.VG_MARGIN_NONE] * nwin
names = ["judge", "assemble", "dg", "build", "chain", "schur", "chol", "back", "chain_back", "lm_y", "norms", "cand", "tail"]
h.ba_upload(probs, flags)
for rep in range(3):
    h.ba_run_async()
    st, sm, _ = h.ba_download()
P = np.array([s['prof'][:13] for s in sm])
it = np.array([s['num_iterations'] for s in sm])
per = P / np.maximum(it, 1)[:, None]
print("windows", nwin, "iterations", it.min(), it.max())
print("%-12s %10s %10s %10s" % ("phase", "median", "min", "max"), " (cycles per round, thread 0)")
for k, n in enumerate(names):
    print("%-12s %10.0f %10.0f %10.0f" % (n, np.median(per[:, k]), per[:, k].min(), per[:, k].max()))
print("%-12s %10.0f %10.0f %10.0f" % ("total", np.median(per.sum(1)), per.sum(1).min(), per.sum(1).max()))
