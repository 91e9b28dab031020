#!/bin/bash
# usage (GPU box, repo root): bash profiles/run_box_kind.sh <tag>
# What kind of box is this?  (the pool has boxes that run the latency-bound BA kernels 1.2-1.5 x apart.)  One call:
# the memory / launch / instruction-fetch probe, the clock probe inside a short bench line (classifies the box by the solve
# kernel's launch time), then two counter passes of the same short bench: instruction-cache requests / hits / misses of the SQC and
# wave / wait / VALU cycles of the SQ, per kernel.  Output: gpurun_out/<tag>_box_*.
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"; mkdir -p gpurun_out
timeout 60 profiles/ubench/mem_latency > gpurun_out/${TAG}_box_mem_latency.json 2>&1
timeout 200 python bench.py --no-cpu-baseline --steps 20 --quick-fe 2>/dev/null | grep '^{' | tail -1 > gpurun_out/${TAG}_box_bench.json
OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 1 --in-flight 1 --quick-fe"
timeout 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace -d "$OUT/ic" -o ic -- $CMD > "$OUT/ic.log" 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/sq" -o sq -- $CMD > "$OUT/sq.log" 2>&1
cd "$ROOT"
python profiles/summarize_counters.py $(find $OUT/ic -name '*.db' | head -1) > gpurun_out/${TAG}_box_icache.txt 2>&1 || tail -5 "$OUT/ic.log" >> gpurun_out/${TAG}_box_icache.txt
python profiles/summarize_counters.py $(find $OUT/sq -name '*.db' | head -1) > gpurun_out/${TAG}_box_sq.txt 2>&1 || tail -5 "$OUT/sq.log" >> gpurun_out/${TAG}_box_sq.txt
rm -rf "$OUT"
python - <<P
import json
d = json.load(open("gpurun_out/${TAG}_box_bench.json"))
k = d["roofline"]["kernels"]
print("box", "${TAG}", "solves/s", round(d["value"]), {n.replace("ba_", "").replace("_kernel", ""): round(v["ms_per_launch"] * 1e3, 1) for n, v in k.items()}, d["device"].get("clock_probe"))
P
cat gpurun_out/${TAG}_box_mem_latency.json | tail -2
grep -E "ba_solve_kernel|ba_final_kernel|ba_linacc|kernel  " gpurun_out/${TAG}_box_icache.txt | head -8
