cd ${GRAFT_REPO_ROOT:-.}; mkdir -p gpurun_out
( timeout 60 python -m pytest tests/test_seq_gpu.py -q -p no:cacheprovider --timeout 50 --tb=short -s 2>&1 | tail -30 ) > gpurun_out/r03u_seq_tests.log
tail -4 gpurun_out/r03u_seq_tests.log
