# Same-box A/B of the working tree against the tree exported under ab_old/ (git archive of a reference commit, built in place):
#   gpurun -- 'bash tests/manual/ab_tree.sh <tag> [pytest selection]'
# GPU tests of the working tree, then bench of both trees back to back (twice, interleaved) -> gpurun_out/<tag>_*.
TAG=${1:-ab}
SEL=${2:-tests}
python -m pytest $SEL -q -m gpu -x > gpurun_out/${TAG}_tests.log 2>&1; tail -3 gpurun_out/${TAG}_tests.log
for rep in 1 2; do
  python bench.py --no-cpu-baseline --steps 40 > gpurun_out/${TAG}_new${rep}.json 2> gpurun_out/${TAG}_new${rep}.err
  (cd ab_old && python bench.py --no-cpu-baseline --steps 40 > ../gpurun_out/${TAG}_old${rep}.json 2> ../gpurun_out/${TAG}_old${rep}.err)
done
python - $TAG <<'P'
import json, sys
tag = sys.argv[1]
for f in ("new1", "old1", "new2", "old2"):
    try:
        d = json.loads([l for l in open(f"gpurun_out/{tag}_{f}.json") if l.startswith("{")][0])
        k = d["roofline"]["kernels"]
        print(f, "solves/s", round(d["value"]), "long", round(d["long_run"]["value"]), "| us/launch",
              {n.replace("ba_", "").replace("_kernel", ""): round(v["ms_per_launch"] * 1e3, 1) for n, v in k.items()},
              "| single window ms", round(d["single_window"]["solve_pipeline_ms"], 3), round(d["single_window"]["marginalization_ms"], 3),
              "states", round(d["single_window"]["states_on_host_ms"], 3), "| KLT M/s", round(d["fe"]["value"] / 1e6, 2))
    except Exception as e:
        print(f, "failed:", e)
P
