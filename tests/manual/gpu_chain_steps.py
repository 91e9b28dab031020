"""Device time and upload phases of every frame of the chained boundary loop of bench.py."""
import os, sys, time
os.environ["VG_DEBUG_UPLOAD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
graft.load_package()
from vins_mono_amd import ba, synth
import bench
n = 256
h = ba.Handle()
probs, seqs = bench.make_windows(h, ba, synth, n, seed0=1000)
flags = [ba.VG_MARGIN_OLD] * n
packed = [ba.PackedProblem(p) for p in probs]
chain = bench.make_chain(h, ba, seqs, packed, flags)
for rep in range(2):
    for k, cb in enumerate(chain):
        t0 = time.perf_counter(); h.ba_upload(cb, flags); up = (time.perf_counter() - t0) * 1e3
        a, b = h.ba_run_timed()
        st, sm, _ = h.ba_download()
        it = np.array([s['num_iterations'] for s in sm])
        print("frame %d: upload %.2f ms, solve %.3f ms, marg %.3f ms, L mean %.0f max %d, iterations mean %.1f, rounds %d" %
              (k, up, a, b, np.mean([p.L for p in cb.packed]), max(p.L for p in cb.packed), it.mean(), max(p.struct.max_iters for p in cb.packed)), flush=True)
        # run_timed is synchronous; run again so that the next frame's resident prior is this frame's
