#!/bin/bash
# usage: bash profiles/run_last.sh <tag>   (the sequence / reserve GPU tests, then the default bench line exactly as the driver runs it)
set -u
TAG=$1
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( timeout 100 python -m pytest tests/test_seq_gpu.py -q -p no:cacheprovider --timeout 90 --tb=short 2>&1 | tail -25 ) > gpurun_out/${TAG}_seq_tests.log
( timeout 150 python bench.py 2> gpurun_out/${TAG}_bench.err | grep '^{' | tail -1 ) > gpurun_out/${TAG}_bench.json
tail -3 gpurun_out/${TAG}_seq_tests.log
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_bench.json')); print('value', d['value'], 'long_run', d['long_run'], 'traffic', d['roofline']['traffic'], 'frac', d['roofline']['frac'])"
