#!/bin/bash
# usage (on the GPU box, from the repo root):  bash profiles/run_round_end.sh <tag>
# One call that fits a short GPU budget, most important first: the GPU test suite, the headline bench line (direct launches),
# the same with the solve pipeline replayed as a hipGraph, the rocprofv3 kernel trace of the bench.  Every part has its own
# timeout, so a slow part cannot starve the ones behind it.  Copy gpurun_out/<tag>_* into profiles/ afterwards.
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out
date +%s > gpurun_out/${TAG}_t0
( timeout 270 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 200 2>&1 | tail -60 ) > gpurun_out/${TAG}_tests.log
date +%s > gpurun_out/${TAG}_t1
( timeout 170 python bench.py --steps 20 --warmup 5 2> gpurun_out/${TAG}_bench.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench.json
date +%s > gpurun_out/${TAG}_t2
( timeout 120 python bench.py --steps 20 --warmup 5 --launch-mode graph --no-cpu-baseline 2> gpurun_out/${TAG}_bench_graph.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_graph.json
date +%s > gpurun_out/${TAG}_t3
timeout 170 bash profiles/run_profile.sh "$TAG"_trace --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
rm -rf gpurun_out/prof_${TAG}_trace
date +%s > gpurun_out/${TAG}_t4
tail -3 gpurun_out/${TAG}_tests.log
python - <<PY
import json
for n in ("bench", "bench_graph"):
    try:
        d = json.load(open("gpurun_out/${TAG}_%s.json" % n))
        hb = d["host_boundary_inclusive"]
        print(n, "value %.0f" % d["value"], "launch", d["config"]["launch"], "single", d.get("single_window"), "resident", hb.get("resident_sequence_solves_per_s"), hb.get("resident_sequence"), "chained", hb.get("chained_solves_per_s"))
    except Exception as ex:
        print(n, "unreadable:", ex)
PY
