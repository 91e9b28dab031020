"""FE tracking step of bench.py (256 streams x 150 tracks) split into its launches: pyramid build and LK, HIP events.
usage: python tests/manual/gpu_fe_step.py [libname.so ...]   (each library in its own process, same box: A/B)"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] != "--one":
    for rep in (1, 2):
        for name in sys.argv[1:]:
            r = subprocess.run([sys.executable, __file__, "--one", name], stdout=subprocess.PIPE, text=True)
            print(rep, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "failed", flush=True)
    sys.exit(0)

import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
name = sys.argv[2] if len(sys.argv) > 2 else "libvinsgpu.so"
pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), name)
from vins_mono_amd import ba, synth, fe
h = ba.Handle()
W, H, N, cams = 752, 480, 150, 256
base = [synth.synth_frame(1000 + c) for c in range(8)]
nxt = [synth.warp_frame(b, 2000 + c) for c, b in enumerate(base)]
tr = fe.FrontEnd(h, W, H, cams, N)
tr.push_frames([base[c % 8] for c in range(cams)])
tr.detect_upload([N] * cams); tr.detect_async(); corners = tr.detect_download()
tr.upload_frames([nxt[c % 8] for c in range(cams)])
tr.track_upload(corners)
slot = tr.frame_slot()                  # as bench.py: the steps alternate between the two resident frames (A -> B, B -> A)
def step(tb=None, tl=None):
    global slot
    tr.select_frames(slot)
    if tb is not None: h.timer_start()
    tr.build_async(False)
    if tb is not None: tb.append(h.timer_stop()); h.timer_start()
    tr.track_async()
    if tb is not None: tl.append(h.timer_stop())
    slot ^= 1
for _ in range(4):
    step()
h.sync()
tb, tl, ts = [], [], []
for _ in range(12):
    step(tb, tl)
for _ in range(6):
    h.timer_start(); step(); step(); ts.append(h.timer_stop() / 2)
tc = []
for _ in range(6):
    tr.select_frames(slot); h.timer_start(); tr.build_async(True); tc.append(h.timer_stop()); slot ^= 1
tg = []
tr.detect_upload([N] * cams)
for _ in range(6):
    h.timer_start(); tr.detect_async(); tg.append(h.timer_stop())
cor = tr.detect_download()
res = tr.track_download()
import zlib
chk = 0
for xy, st, er in res:
    chk = zlib.crc32(np.ascontiguousarray(xy).tobytes() + st.tobytes() + np.ascontiguousarray(er).tobytes(), chk)
print(json.dumps({"lib": name, "pyramid_us": round(np.median(tb) * 1e3, 1), "lk_us": round(np.median(tl) * 1e3, 1), "lk_us_by_direction": [round(np.median(tl[0::2]) * 1e3, 1), round(np.median(tl[1::2]) * 1e3, 1)],
                  "step_us": round(np.median(ts) * 1e3, 1), "Mfeat_s": round(cams * N / np.median(ts) * 1e-3, 2), "clahe_plus_pyramid_us": round(np.median(tc) * 1e3, 1), "gftt_us": round(np.median(tg) * 1e3, 1), "gftt_crc": zlib.crc32(b"".join(np.ascontiguousarray(c).tobytes() for c in cor)), "checksum": chk}))
