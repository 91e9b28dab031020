#!/bin/bash
# usage (GPU box, repo root): bash tests/manual/prof_fe_step.sh <tag> [lib]  -> gpurun_out/<tag>_fe_step_trace.txt (per-grid kernel durations of the FE step)
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o "$TAG" -- python "$ROOT/tests/manual/gpu_fe_step.py" --one "${2:-libvinsgpu.so}" > "$OUT/run.log" 2>&1
DB=$(find "$OUT" -name '*.db' | head -1)
python "$ROOT/profiles/summarize_rocpd.py" "$DB" > "$ROOT/gpurun_out/${TAG}_fe_step_trace.txt" 2>&1
cat "$ROOT/gpurun_out/${TAG}_fe_step_trace.txt" | grep -v rocclr
rm -rf "$OUT"
