python -m pytest tests/test_ba_gpu.py tests/test_ba_large_gpu.py -q -m gpu -x > gpurun_out/ab_tests.log 2>&1; tail -1 gpurun_out/ab_tests.log
python bench.py --no-cpu-baseline --steps 20 > gpurun_out/ab_new.json 2>gpurun_out/ab_new.err
python tests/manual/gpu_detail_profile.py > gpurun_out/ab_detail.txt 2>&1
python tests/manual/gpu_phase_profile.py > gpurun_out/ab_phase.txt 2>&1
python tests/manual/gpu_phase_profile_large.py 0 > gpurun_out/ab_phase_large.txt 2>&1
cp vins-mono_amd/lib/libvinsgpu.so /tmp/new.so
cp vins-mono_amd/lib/libvinsgpu_prev.so vins-mono_amd/lib/libvinsgpu.so
python bench.py --no-cpu-baseline --steps 20 > gpurun_out/ab_old.json 2>gpurun_out/ab_old.err
for f in new old; do python - $f <<'P'
import json,sys
f=sys.argv[1]
d=json.loads([l for l in open(f"gpurun_out/ab_{f}.json") if l.startswith("{")][0])
print(f, round(d["value"]), round(d["roofline"]["kernels"]["ba_solve_kernel"]["ms_per_launch"]*1e3,1), round(d["single_window"]["solve_pipeline_ms"],3), round(d["fe"]["value"]))
P
done
