"""Phase stamps of ba_marg_kernel on one window (library built with -DBA_PROFILE: libvinsgpu_prof.so); VG_DEBUG_MARG prints them."""
import os, sys
os.environ["VG_DEBUG_MARG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libvinsgpu_prof.so")
from vins_mono_amd import ba, synth
h = ba.Handle()
import numpy as np
seq = synth.SyntheticSequence(5, n_frames=18, L=150)
prob = seq.window(0)
for w0 in range(1, int(sys.argv[1]) if len(sys.argv) > 1 else 2):      # walk the window along the sequence: more landmarks anchored at frame 0
    st, sm, pr = h.ba_optimize(prob, ba.VG_MARGIN_OLD)
    prob = seq.next_window(st, pr, w0)
    print("window", w0, "landmarks", len(prob['lm_start']), "anchored at frame 0:", int((np.asarray(prob['lm_start']) == 0).sum()), file=sys.stderr, flush=True)
for _ in range(2):
    h.ba_optimize(prob, ba.VG_MARGIN_OLD)
