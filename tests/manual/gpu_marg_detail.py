"""Sub-phase timers of the one-sided Jacobi rounds of ba_marg_kernel (library built with -DBA_PROFILE_DETAIL)."""
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
h.ba_optimize(prob, ba.VG_MARGIN_OLD)
out = np.zeros(16)
h.lib.vg_debug_marg_profile(out.ctypes.data_as(C.POINTER(C.c_double)), 1)
rel = np.zeros(16)
h.lib.vg_debug_marg_rel(rel.ctypes.data_as(C.POINTER(C.c_double)))
h.ba_optimize(prob, ba.VG_MARGIN_OLD)
h.lib.vg_debug_marg_profile(out.ctypes.data_as(C.POINTER(C.c_double)), 1)
h.lib.vg_debug_marg_rel(rel.ctypes.data_as(C.POINTER(C.c_double)))
print("max relative |g_p.g_q| / (|g_p| |g_q|) of the rotated pairs per sweep", ["%.1e" % np.sqrt(v) for v in rel[:11]])
for n, v in zip(["loads + dot", "row16 reduce", "norm reads + wave barrier", "rotation + stores", "barrier"], out):
    print(f"{n:<28}{v:>12.0f}")
print("total", out[:5].sum())
print("rotations per sweep", [int(v) for v in out[5:16]])
