"""MANUAL: wall time of each C-ABI call a published frame of FeatureTracker::readImage makes, one stream (752x480, 150 corners)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import conftest  # noqa
import fe_scene
from vins_mono_amd import ba, fe
h = ba.Handle()
tr = fe.FrontEnd(h, 752, 480, 1, 600)
frames = fe_scene.moving_scene(30, seed=3)
intr = np.array([461.6, 460.3, 363.0, 248.1, -2.917e-01, 8.228e-02, 5.333e-05, -1.578e-04])
T = {}
def tm(name, f):
    t0 = time.perf_counter(); r = f(); T.setdefault(name, []).append((time.perf_counter() - t0) * 1e3); return r
tr.push_frames([frames[0]], equalize=True)
pts = tr.detect(0, 150)
cnt = np.ones(len(pts), np.int32)
for k in range(1, 30):
    tm("push_frames(CLAHE+pyramid, H2D 361KB)", lambda: tr.push_frames([frames[k]], equalize=True))
    nxt, st, err = tm("track (H2D pts, LK, D2H)", lambda: tr.track(0, pts))
    keep = st.astype(bool)
    cur, nxt = pts[keep], nxt[keep]; cnt = cnt[keep] + 1
    un1 = (tr.undistort(cur, intr).astype(np.float64) * 460 + [376, 240]).astype(np.float32)
    un2 = (tr.undistort(nxt, intr).astype(np.float64) * 460 + [376, 240]).astype(np.float32)
    stf = tm("reject_with_f", lambda: tr.reject_with_f(un1, un2, 1.0))[0].astype(bool)
    nxt, cnt = nxt[stf], cnt[stf]
    kept = tm("set_mask", lambda: tr.set_mask([nxt], [cnt], 30))[0]
    nxt, cnt = nxt[kept], cnt[kept]
    new = tm("detect_masked", lambda: tr.detect_masked(0, 150 - len(nxt)))
    pts = np.concatenate([nxt, new]).astype(np.float32); cnt = np.concatenate([cnt, np.ones(len(new), np.int32)])
    tm("undistort", lambda: tr.undistort(pts, intr))
for k, v in T.items():
    print("%-42s median %.3f ms  (min %.3f)" % (k, np.median(v[3:]), np.min(v[3:])))
