"""Phase times of vg_ba_batch_upload / download for the bench batch (VG_DEBUG_UPLOAD=1)."""
import os, sys, time
os.environ["VG_DEBUG_UPLOAD"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
graft.load_package()
from vins_mono_amd import ba, synth
import bench
h = ba.Handle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
probs, seqs = bench.make_windows(h, ba, synth, n, seed0=1000)
flags = [ba.VG_MARGIN_OLD] * n
pb = ba.PackedBatch(probs)
for _ in range(4):
    h.ba_upload(pb, flags)
h.ba_run_async()
st, sm, pr = h.ba_download()
nxt = ba.PackedBatch([q.next_window(st[i], 'resident', 2) for i, q in enumerate(seqs)])
for _ in range(3):
    h.ba_upload(pb, flags); h.ba_run_async(); h.sync()
    t0 = time.perf_counter(); h.ba_upload(nxt, flags); print("resident upload call %.3f ms" % ((time.perf_counter() - t0) * 1e3)); h.ba_run_async()
    dl = h.ba_prepare_download()
    t0 = time.perf_counter(); h.ba_download_state_raw(dl); print("state download (incl. wait for the solve) %.3f ms" % ((time.perf_counter() - t0) * 1e3))
    h.sync()
    t0 = time.perf_counter(); h.ba_download_state_raw(dl); print("state download alone %.3f ms" % ((time.perf_counter() - t0) * 1e3))
    t0 = time.perf_counter(); h.ba_download_raw(); print("full download alone %.3f ms" % ((time.perf_counter() - t0) * 1e3))
