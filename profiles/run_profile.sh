#!/bin/bash
# usage (on the GPU box, from the repo root):  bash profiles/run_profile.sh <tag> [bench args]
# rocprofv3 kernel trace + stats of `python bench.py`, summarised into gpurun_out/<tag>_kernel_trace_stats.txt
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o "$TAG" -- python "$ROOT/bench.py" "$@" > "$OUT/bench.log" 2>&1
DB=$(find "$OUT" -name '*.db' | head -1)
python "$ROOT/profiles/summarize_rocpd.py" "$DB" > "$ROOT/gpurun_out/${TAG}_kernel_trace_stats.txt" 2>&1
grep "^{" "$OUT/bench.log" | tail -1 > "$ROOT/gpurun_out/${TAG}_bench.json"
cat "$ROOT/gpurun_out/${TAG}_kernel_trace_stats.txt"
