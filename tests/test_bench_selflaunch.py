"""`python bench.py --gpus N` without WORLD_SIZE starts its own ranks, for both configs: two ranks on the emulated
kernels over gloo.  (A file of its own so that the xdist workers of the `not gpu` suite run it beside tests/test_bench_contract.py.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks_for_n_gpus():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start two ranks itself (torch.distributed.run, 127.0.0.1)
    and rank 0 must print ONE line with n_gpus: 2 and the SUM of both ranks' solves -- a driver that runs the N = 1 command shape with
    --gpus 8 can then never time one GPU silently.  Here: two ranks on the emulated kernels over gloo."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emulated", "--backend", "gloo", "--windows", "2",
                        "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    # value = windows of BOTH ranks x steps / timed region (max over ranks)
    assert abs(d["value"] - 2 * 2 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    assert "x2" in d["config"]["parallelism"]
    assert d["fe"]["value_all_gpus"] > d["fe"]["value"]                      # (the front-end figure is summed over the ranks too)
    # a rank count that contradicts --gpus is refused instead of being measured
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emulated"], capture_output=True, text=True, timeout=300,
                       cwd=ROOT, env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stdout + r.stderr)


def test_sharded_bench_starts_its_own_ranks():
    """The same for `--config sharded` (BASELINE configs[4]): two ranks, each with its landmark shard of the ONE window, the reduced
    camera system summed between them (gloo here, ncclAllReduce inside the library on GPUs); strong scaling, n_gpus: 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emulated", "--backend", "gloo", "--config", "sharded",
                        "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "shards x2" in d["config"]["parallelism"] and d["value"] > 0
