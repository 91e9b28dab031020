"""Phase stamps of ba_marg_kernel inside a full batch (library built with -DBA_PROFILE; VG_DEBUG_MARG=1 makes unpack_priors print them)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["VG_DEBUG_MARG"] = "1"
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libvinsgpu_prof.so")
from vins_mono_amd import ba, synth
import bench
h = ba.Handle()
nwin = int(sys.argv[1]) if len(sys.argv) > 1 else 256
probs, seqs = bench.make_windows(h, ba, synth, nwin, seed0=1)
h.ba_upload(probs, [ba.VG_MARGIN_OLD] * nwin)
for rep in range(2):
    h.ba_run_async(); h.sync()
h.ba_run_async()
h.ba_download()
