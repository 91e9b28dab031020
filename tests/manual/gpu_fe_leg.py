"""MANUAL: the front-end leg of bench.py alone (256 streams at 752 x 480: KLT features/s, GFTT frames/s), for A/B of the FE kernels
without the BA legs.   python tests/manual/gpu_fe_leg.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

graft.load_package()
from vins_mono_amd import ba, synth  # noqa: E402

bench.FE_DISTINCT = 8                      # (generator time; the kernels do the same work)
h = ba.Handle()
out = bench.bench_fe(h, synth, int(sys.argv[1]) if len(sys.argv) > 1 else 20, 3, 0, False)
print(json.dumps({k: out[k] for k in out if k in ("value", "klt_ms_per_batch", "gftt_frames_per_s", "gftt_ms_per_batch", "klt_clahe_features_per_s")} | {"all_keys": sorted(out)}))
