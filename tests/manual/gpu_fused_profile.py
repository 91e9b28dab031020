"""Sub-phase timers of ba_linacc_proj_kernel inside a full batch (library built with -DBA_PROFILE_DETAIL, see gpu_detail_profile.py):
cycles of thread 0 / thread 128 of workgroup 0 per round."""
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
h.ba_run_async(); st, sm, _ = h.ba_download()
h.lib.vg_debug_detail_profile(out.ctypes.data_as(C.POINTER(C.c_double)), 1)
names = {22: "IMU pass", 23: "prior pass + cost sum", 24: "chunk table, pair-table clear", 25: "linearise (per solve: all chunks)", 26: "pair blocks (MFMA)",
         27: "landmark sums + W stores", 28: "tail: zero columns, Sp / gp, cost sum", 29: "  (pair: start table reads)", 30: "  (pair: MFMA trips)"}
it = 8
print("window 0: F =", int(sum(probs[0]['lm_nobs']) - len(probs[0]['lm_nobs'])), " (cycles per round)")
for i, n in names.items():
    print(f"{n:<44}{out[i] / it:>12.0f}{out[32 + i] / it:>12.0f}")
print(f"{'total':<44}{sum(out[i] for i in names if i < 29) / it:>12.0f}")
hi = np.zeros(64)
h.lib.vg_debug_detail_profile_hi(hi.ctypes.data_as(C.POINTER(C.c_double)), 1)
h.ba_run_async(); h.sync()
h.lib.vg_debug_detail_profile_hi(hi.ctypes.data_as(C.POINTER(C.c_double)), 1)
pro = {45: "factor kernel: fetch (all rounds)", 46: "factor kernel: sort (all rounds)", 40: "prologue: clears, state copy", 41: "prologue: IMU sqrt_info", 42: "prologue: J0^T copy", 43: "prologue: J0 staged", 44: "prologue: J0^T J0 (MFMA)"}
for i, n in pro.items():
    print(f"{n:<44}{hi[i - 32]:>12.0f}{hi[i]:>12.0f}   (per solve)")
