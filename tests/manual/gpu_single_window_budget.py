"""Where one window's 1 ms goes: per-launch durations of every kernel class for a single-window batch (HIP events after
every launch), the gaps between launches (pipeline total - sum of launches), and the host-side pieces of the boundary."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
if len(sys.argv) > 1:                      # another build of the library: vins-mono_amd/lib/<name>
    pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), sys.argv[1])
from vins_mono_amd import ba, synth
h = ba.Handle()
seq = synth.SyntheticSequence(5, L=150)
st, sm, pr = h.ba_optimize(seq.window(0), ba.VG_MARGIN_OLD)
prob = ba.PackedProblem(seq.next_window(st, pr, 1))
h.ba_upload([prob], [ba.VG_MARGIN_OLD])
for _ in range(5):
    h.ba_run_timed()
tim = np.array([h.ba_run_timed() for _ in range(30)])
print("events around the whole pipeline: solve %.3f ms, marginalization %.3f ms" % tuple(np.median(tim, axis=0)))
acc = {}
for _ in range(30):
    for k, (ms, n) in h.ba_run_profiled().items():
        a = acc.setdefault(k, [[], n]); a[0].append(ms)
tot = 0.0
for k, (v, n) in acc.items():
    m = float(np.median(v)); tot += m
    print("  %-52s %2d launches  %.1f us each  %.3f ms" % (k, n, m / max(n, 1) * 1e3, m))
print("  sum of the launch classes (an event after every launch, so launch-to-launch gaps are inside) %.3f ms" % tot)
# boundary pieces
ts = []
for _ in range(30):
    t0 = time.perf_counter(); h.ba_upload([prob], [ba.VG_MARGIN_OLD]); t1 = time.perf_counter()
    h.ba_run_async(); t2 = time.perf_counter()
    dl = h.ba_prepare_download(); t3 = time.perf_counter()
    h.ba_download_state_raw(dl); t4 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1, t4 - t3))
ts = np.median(np.array(ts), axis=0) * 1e3
print("host side: upload call %.3f ms, run_async (enqueue of all launches) %.3f ms, state download call (waits for the solve) %.3f ms" % tuple(ts))
