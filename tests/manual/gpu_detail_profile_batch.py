"""gpu_detail_profile.py inside a full batch: sub-phase timers of chain_schur / cholesky_aug / assemble (workgroup 0)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libvinsgpu_dprof.so")
from vins_mono_amd import ba, synth
import bench
h = ba.Handle()
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
probs, seqs = bench.make_windows(h, ba, synth, nwin, seed0=1)
h.ba_upload(probs, [ba.VG_MARGIN_NONE] * nwin)
out = np.zeros(64)
for rep in range(2):
    h.ba_run_async(); h.sync()
h.lib.vg_debug_detail_profile(out.ctypes.data_as(C.POINTER(C.c_double)), 1)
h.ba_run_async(); h.ba_download()
h.lib.vg_debug_detail_profile(out.ctypes.data_as(C.POINTER(C.c_double)), 1)
names = ["chain A work", "chain A wait", "chain B work", "chain B wait", "chol diag", "chol wait1", "chol panel", "chol wait2", "chol trailing",
         "chol wait3", "chain C (MFMA)", "schur trip wait", "schur stage", "schur wait", "schur fetch+mfma", "schur end wait",
         "assemble copy + clear", "assemble wait", "chain A: stage issue", "chain A: raw col + scale", "chain A: update", "assemble end wait"]
print("windows", nwin, "(cycles per round, workgroup 0)")
print(f"{'phase':<24}{'thread 0':>12}{'thread 128':>12}")
for i, n in enumerate(names):
    print(f"{n:<24}{out[i] / 8:>12.0f}{out[32 + i] / 8:>12.0f}")
