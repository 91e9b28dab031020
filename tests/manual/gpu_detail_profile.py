"""Sub-phase timers of chain_schur / cholesky_aug (library built with -DBA_PROFILE_DETAIL:
make -C vins-mono_amd/csrc OBJDIR=../build_dprof LIB=../lib/libvinsgpu_dprof.so EXTRA=-DBA_PROFILE_DETAIL).  Cycles of thread 0
(wavefront 0) and thread 128 (wavefront 2) of workgroup 0, summed over the rounds of one solve."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libvinsgpu_dprof.so")
from vins_mono_amd import ba, synth
h = ba.Handle()
seq = synth.SyntheticSequence(5, L=150)
p1 = seq.window(0)
st, sm, pr = h.ba_optimize(p1, ba.VG_MARGIN_OLD)
prob = seq.next_window(st, pr, 1)
h.ba_optimize(prob, ba.VG_MARGIN_NONE)
out = np.zeros(64)
h.lib.vg_debug_detail_profile(out.ctypes.data_as(C.POINTER(C.c_double)), 1)
st2, sm2, _ = h.ba_optimize(prob, ba.VG_MARGIN_NONE)
h.lib.vg_debug_detail_profile(out.ctypes.data_as(C.POINTER(C.c_double)), 1)
names = ["chain A work", "chain A wait", "chain B work", "chain B wait", "chol diag", "chol wait1", "chol panel", "chol wait2", "chol trailing",
         "chol wait3", "schur lsc+chain rows", "schur trip wait", "schur stage", "schur wait", "schur fetch+mfma", "schur end wait",
         "assemble copy + clear", "assemble wait", "assemble IMU scatter", "assemble IMU wait", "assemble prior scatter", "assemble end wait"]
it = max(sm2['num_iterations'], 1)
print("iterations", it, "(cycles per round)")
print(f"{'phase':<24}{'thread 0':>12}{'thread 128':>12}")
for i, n in enumerate(names):
    print(f"{n:<24}{out[i] / it:>12.0f}{out[32 + i] / it:>12.0f}")
