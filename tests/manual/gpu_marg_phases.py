"""Phase stamps of ba_marg_kernel on one window (library built with -DBA_PROFILE: libvinsgpu_prof.so); VG_DEBUG_MARG prints them."""
import os, sys
os.environ["VG_DEBUG_MARG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libvinsgpu_prof.so")
from vins_mono_amd import ba, synth
h = ba.Handle()
seq = synth.SyntheticSequence(5, L=150)
p1 = seq.window(0)
st, sm, pr = h.ba_optimize(p1, ba.VG_MARGIN_OLD)
prob = seq.next_window(st, pr, 1)
for _ in range(2):
    h.ba_optimize(prob, ba.VG_MARGIN_OLD)
