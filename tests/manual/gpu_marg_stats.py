"""Distribution of Jacobi sweeps / Cholesky attempts of the marginalization kernel over a 256-window batch
(VG_DEBUG_MARG=1 makes vg_ba_batch_download print one line per window)."""
import os, sys
os.environ["VG_DEBUG_MARG"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
g.load_package()
from vins_mono_amd import ba, synth
import bench
h = ba.Handle()
probs = bench.make_windows(h, ba, synth, 256, seed0=1)
os.environ["VG_DEBUG_MARG"] = "1"
h.ba_upload(probs, [ba.VG_MARGIN_OLD] * 256)
h.ba_run_async()
h.ba_download()
