#!/bin/bash
# usage (on the GPU box, from the repo root):  bash profiles/run_pmc.sh <tag>
# HBM traffic and SQ activity of `python bench.py` per kernel: three SEPARATE rocprofv3 --pmc passes (FETCH_SIZE and
# WRITE_SIZE do not fit one TCC pass; SQ counters in their own pass), each with --kernel-trace only
# (MI355X_MICROARCH.md "rocprofv3 PMC slots").  Summaries: gpurun_out/<tag>_pmc_hbm.{txt,json}, <tag>_sq_counters.txt.
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --no-cpu-baseline --steps 5 --warmup 1 --in-flight 1 --quick-fe"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/fetch" -o fetch -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/write" -o write -- $CMD > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
          --kernel-trace -d "$OUT/sq" -o sq -- $CMD > "$OUT/sq.log" 2>&1
BUILD=$(python -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.csrc_tag())")
python "$ROOT/profiles/summarize_pmc.py" "$(find $OUT/fetch -name '*.db' | head -1)" "$(find $OUT/write -name '*.db' | head -1)" \
       "$ROOT/gpurun_out/${TAG}_pmc_hbm.json" "$TAG: python bench.py --no-cpu-baseline --steps 5 --in-flight 1" "$BUILD" > "$ROOT/gpurun_out/${TAG}_pmc_hbm.txt" 2>&1
python "$ROOT/profiles/summarize_counters.py" --merge-into "$ROOT/gpurun_out/${TAG}_pmc_hbm.json" "$(find $OUT/sq -name '*.db' | head -1)" > "$ROOT/gpurun_out/${TAG}_sq_counters.txt" 2>&1
cat "$ROOT/gpurun_out/${TAG}_pmc_hbm.txt"; tail -12 "$ROOT/gpurun_out/${TAG}_sq_counters.txt"
rm -rf "$OUT"
