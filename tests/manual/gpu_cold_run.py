"""Is the first run after an upload slower than a repeated run on the same buffers?"""
import os, sys, time
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
pb = ba.PackedBatch(probs)
h.ba_upload(pb, flags)
for _ in range(3): h.ba_run_timed()
for rep in range(4):
    h.ba_upload(pb, flags)
    a = h.ba_run_timed(); b = h.ba_run_timed(); c = h.ba_run_timed()
    print("after upload: solve %.3f marg %.3f | again: %.3f %.3f | again: %.3f %.3f" % (a + b + c))
