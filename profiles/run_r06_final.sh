#!/bin/bash
# usage (on the GPU box, from the repo root):  bash profiles/run_r06_final.sh <tag>
# The measurement set of a finished tree, every part under its own timeout: GPU tests + bench + graph bench + kernel trace of the
# bench (run_round_end.sh), the three PMC passes (run_pmc.sh), a clean kernel trace of the TIMED LOOP alone (bench.py --profile-loop,
# one stream and sixteen), the enlarged-window bench, the front end through the reference's node, the sequence-flip statistics.
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out
bash profiles/run_round_end.sh $TAG > gpurun_out/${TAG}_round_end.log 2>&1
timeout 500 bash profiles/run_pmc.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1
timeout 150 bash profiles/run_profile.sh ${TAG}_timed_loop --profile-loop --no-cpu-baseline --steps 20 --warmup 5 > /dev/null 2>&1
rm -rf gpurun_out/prof_${TAG}_timed_loop
timeout 150 bash profiles/run_profile.sh ${TAG}_timed_loop_one_stream --profile-loop --no-cpu-baseline --steps 20 --warmup 5 --in-flight 1 > /dev/null 2>&1
rm -rf gpurun_out/prof_${TAG}_timed_loop_one_stream
( timeout 200 python bench.py --config sharded --no-cpu-baseline 2> gpurun_out/${TAG}_sharded_bench.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_sharded_bench.json
( timeout 200 python tests/manual/gpu_readimage_latency.py 2> gpurun_out/${TAG}_readimage_latency.err | tail -1 ) > gpurun_out/${TAG}_readimage_latency.json
( timeout 200 python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench_200.json
( timeout 400 python tests/manual/gpu_flip_stats.py 100 24 2> gpurun_out/${TAG}_flip_stats.err ) > gpurun_out/${TAG}_flip_stats.json
tail -3 gpurun_out/${TAG}_tests.log
python - <<PY
import json
for n in ("bench", "bench_graph", "bench_200", "sharded_bench"):
    try:
        d = json.loads([l for l in open("gpurun_out/${TAG}_%s.json" % n) if l.startswith("{")][-1])
        print(n, "value %.1f" % d["value"], "ms_per_step %.4f" % d["ms_per_step"])
    except Exception as ex:
        print(n, "unreadable:", ex)
PY
head -20 gpurun_out/${TAG}_pmc_hbm.txt
