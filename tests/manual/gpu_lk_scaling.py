"""How the LK launch scales with the number of tracks: is it bound by throughput or by its slowest track?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
g.load_package()
from vins_mono_amd import ba, synth, fe
h = ba.Handle()
W, H, N = 752, 480, 150
base = [synth.synth_frame(1000 + c) for c in range(8)]
nxt = [synth.warp_frame(b, 2000 + c) for c, b in enumerate(base)]
for cams in (1, 4, 16, 34, 64, 128, 256):
    tr = fe.FrontEnd(h, W, H, cams, N)
    fa = [base[c % 8] for c in range(cams)]
    fb = [nxt[c % 8] for c in range(cams)]
    tr.push_frames(fa)
    tr.detect_upload([N] * cams); tr.detect_async(); corners = tr.detect_download()
    tr.upload_frames(fb)
    tr.track_upload(corners)
    tr.select_frames(tr.frame_slot()); tr.build_async(False)
    for _ in range(3):
        tr.track_async()
    h.sync()
    t = []
    for _ in range(10):
        h.timer_start(); tr.track_async(); t.append(h.timer_stop())
    ntr = sum(len(c) for c in corners)
    print("cams %3d tracks %6d  LK launch %.1f us  -> %.1f M tracks/s" % (cams, ntr, np.median(t) * 1e3, ntr / np.median(t) * 1e-3), flush=True)
