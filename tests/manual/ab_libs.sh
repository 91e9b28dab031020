# Same-box A/B of library variants (vins-mono_amd/lib/<name>) with the working tree's bench: gpurun -- 'bash tests/manual/ab_libs.sh tag libA.so libB.so ...'
TAG=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    python tests/manual/bench_with_lib.py $lib --no-cpu-baseline --steps 40 > gpurun_out/${TAG}_${lib%.so}_${rep}.json 2> gpurun_out/${TAG}_${lib%.so}_${rep}.err
    python - $TAG ${lib%.so} $rep <<'P'
import json, sys
tag, lib, rep = sys.argv[1:4]
try:
    d = json.loads([l for l in open(f"gpurun_out/{tag}_{lib}_{rep}.json") if l.startswith("{")][0])
    k = d["roofline"]["kernels"]
    print(lib, rep, "solves/s", round(d["value"]), "| us/launch", {n.replace("ba_", "").replace("_kernel", ""): round(v["ms_per_launch"] * 1e3, 1) for n, v in k.items()},
          "| single window", round(d["single_window"]["solve_pipeline_ms"], 3), round(d["single_window"]["marginalization_ms"], 3))
except Exception as e:
    print(lib, rep, "failed:", e)
P
  done
done
