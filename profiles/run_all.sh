#!/bin/bash
# usage (on the GPU box, from the repo root):  bash profiles/run_all.sh <tag>
# Everything profiles/README.md lists for a tag: kernel trace of the headline bench and of the sharded-window bench, then
# the three PMC passes.  Copy gpurun_out/<tag>_* into profiles/ afterwards (and <tag>_pmc_hbm.json to pmc_latest.json).
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash "$ROOT/profiles/run_profile.sh" "$TAG" --steps 20 --warmup 5 > /dev/null
bash "$ROOT/profiles/run_profile.sh" "${TAG}_sharded" --config sharded --steps 10 --warmup 2 > /dev/null
bash "$ROOT/profiles/run_pmc.sh" "$TAG" > /dev/null
ls -la "$ROOT"/gpurun_out/${TAG}*
# the raw rocprofv3 databases stay on the box (gpurun merges at most 64 MiB back): only the summaries travel
rm -rf "$ROOT"/gpurun_out/prof_${TAG} "$ROOT"/gpurun_out/prof_${TAG}_sharded "$ROOT"/gpurun_out/pmc_${TAG}
