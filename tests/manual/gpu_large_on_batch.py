"""Development: the large-window path forced on the 256-window EuRoC batch (solve pipeline only), against the ordinary pipeline."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from vins_mono_amd import ba, synth
import bench
h = ba.Handle()
probs = bench.make_windows(h, ba, synth, 256, seed0=1)
packed = [ba.PackedProblem(p) for p in probs]
for large in (False, True):
    hs = [ba.Handle() for _ in range(3)]
    for hh in hs:
        hh.ba_set_large_window(large)
        hh.ba_upload(packed, [ba.VG_MARGIN_NONE] * 256)
    for hh in hs:
        hh.ba_run_async()
    for hh in hs:
        hh.sync()
    one = min(hs[0].ba_run_timed()[0] for _ in range(5))
    t0 = time.perf_counter()
    for k in range(12):
        hs[k % 3].ba_run_async()
    for hh in hs:
        hh.sync()
    dt = (time.perf_counter() - t0) / 12 * 1e3
    st, sm, _ = hs[0].ba_download()
    print("large" if large else "small", "one batch alone %.3f ms; 3 in flight %.3f ms per batch; ok %d" % (one, dt, sum(s['status'] == 0 for s in sm)))
    for hh in hs:
        hh.close()
