cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
( timeout 200 python -m pytest tests/test_seq_gpu.py -q -p no:cacheprovider --timeout 150 --tb=short -rA 2>&1 | tail -120 ) > gpurun_out/r03r_seq_tests.log
( timeout 150 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 150 --tb=line 2>&1 | tail -25 ) > gpurun_out/r03r_tests.log
tail -5 gpurun_out/r03r_seq_tests.log; tail -5 gpurun_out/r03r_tests.log
