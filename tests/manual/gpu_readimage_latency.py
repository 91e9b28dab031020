"""MANUAL (not collected): wall time of one camera stream's whole img_callback() -> FeatureTracker::readImage() per frame
(the reference's own timer is feature_tracker_node.cpp:203, its budget the 50 ms of a 20 Hz camera).

Both runs go through the reference's unchanged node (oracle/_ref/libvins_ref_fe*.so): once with the reference's FeatureTracker on the
restated OpenCV algorithms (oracle/fe_cpu.cpp, one host core), once with the product's drop-in members on the GPU.  EuRoC
configuration (752x480, CLAHE on, 150 corners, FREQ 10 on a 20 Hz stream: every other frame runs rejectWithF + setMask +
goodFeaturesToTrack).      python tests/manual/gpu_readimage_latency.py [n_frames]"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import conftest  # noqa: E402,F401
import fe_scene  # noqa: E402
from oracle import ref_fe as RF  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 80
frames = fe_scene.moving_scene(n, seed=3)
cfg = RF.write_config(os.path.join(tempfile.gettempdir(), "lat_cfg.yaml"))
out = {}
runs = {}
for kind in ("ref", "gpu"):
    L = RF._load(kind)
    for rep in range(2):                              # the second pass is timed (first: allocations, code upload)
        node = RF.Node(L, cfg)
        t, pub = [], []
        for k, f in enumerate(frames):
            f = np.ascontiguousarray(f)
            t0 = time.perf_counter()
            L.vfe_image(RF.C.c_double(100.0 + 0.05 * k), f.ctypes.data_as(RF.C.c_void_p), f.shape[1], f.shape[0], f.strides[0])
            t.append((time.perf_counter() - t0) * 1e3)
            pub.append(bool(L.vfe_pub_this_frame()))
    t, pub = np.array(t[4:]), np.array(pub[4:])
    runs[kind] = [node.tracks(), node.published()]
    out[kind] = {"published_frames_ms": {"median": float(np.median(t[pub])), "max": float(t[pub].max())},
                 "other_frames_ms": {"median": float(np.median(t[~pub])), "max": float(t[~pub].max())}, "frames": int(len(t))}
assert RF.same_tracks(runs["ref"][0], runs["gpu"][0]) and len(runs["ref"][1]) == len(runs["gpu"][1])
out["speedup_published"] = out["ref"]["published_frames_ms"]["median"] / out["gpu"]["published_frames_ms"]["median"]
out["speedup_other"] = out["ref"]["other_frames_ms"]["median"] / out["gpu"]["other_frames_ms"]["median"]
out["what"] = ("one stream, wall time of img_callback per frame incl. H2D of the 361 KB frame and every D2H the class needs; ref = the reference's "
               "FeatureTracker on oracle/fe_cpu.cpp (1 core), gpu = the drop-in members; identical tracks on both sides")
print(json.dumps(out))
