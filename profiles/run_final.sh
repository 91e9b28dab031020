#!/bin/bash
# usage (on the GPU box, from the repo root):  bash profiles/run_final.sh <tag>
# End-of-round check on a short GPU budget: the GPU test suite, then the two HBM PMC passes (FETCH_SIZE, WRITE_SIZE; separate
# rocprofv3 runs with --kernel-trace only) of `python bench.py`, summarised into gpurun_out/<tag>_pmc_hbm.{txt,json}
# (= profiles/pmc_latest.json, which bench.py reads for roofline.traffic when its `build` tag matches the kernel sources).
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out
( timeout 150 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 150 --tb=short 2>&1 | tail -40 ) > gpurun_out/${TAG}_tests.log
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 1 --in-flight 1"
timeout 110 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
timeout 110 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
cd "$ROOT"
BUILD=$(python -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.csrc_tag())")
python "$ROOT/profiles/summarize_pmc.py" "$(find $OUT/fetch -name '*.db' | head -1)" "$(find $OUT/write -name '*.db' | head -1)" \
       "$ROOT/gpurun_out/${TAG}_pmc_hbm.json" "$TAG: python bench.py --no-cpu-baseline --steps 5 --in-flight 1" "$BUILD" > "$ROOT/gpurun_out/${TAG}_pmc_hbm.txt" 2>&1
grep '^{' "$OUT/fetch.log" | tail -1 > gpurun_out/${TAG}_bench_under_pmc.json
rm -rf "$OUT"
tail -4 gpurun_out/${TAG}_tests.log; head -30 gpurun_out/${TAG}_pmc_hbm.txt
