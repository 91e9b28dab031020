"""MANUAL: the front-end leg of bench.py alone (256 streams at 752 x 480: KLT features/s, GFTT frames/s), for A/B of the FE kernels
without the BA legs.   [VINS_AB_LIB=libvinsgpu_x.so] python tests/manual/gpu_fe_leg.py [steps]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as graft  # noqa: E402

pkg = graft.load_package()
if os.environ.get("VINS_AB_LIB"):          # same-box A/B: another build of the library in vins-mono_amd/lib/
    pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), os.environ["VINS_AB_LIB"])
from vins_mono_amd import ba, synth  # noqa: E402

bench.FE_DISTINCT = int(os.environ.get("VINS_FE_DISTINCT", "8"))      # (generator time; LK iterations depend on the content: 256 = the bench workload)
h = ba.Handle()
out = bench.bench_fe(h, synth, int(sys.argv[1]) if len(sys.argv) > 1 else 20, 3, 0, False)
print(json.dumps({k: out[k] for k in out if k in ("value", "klt_ms_per_batch", "gftt_frames_per_s", "gftt_ms_per_batch", "klt_clahe_features_per_s")} | {"all_keys": sorted(out)}))
