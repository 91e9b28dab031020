"""bench.py's output contract, checked WITHOUT a GPU: `python bench.py --emulated` runs every leg of the default bench (timed loop,
per-kernel events, boundary loops, resident sequences, CPU baseline + parity, front end, single window) against the kernel sources
under the CPU fiber emulator with tiny loop counts.  The numbers are meaningless; what is checked is that the file still runs end to
end and that the ONE JSON line has the fields and types the driver reads (a bench that crashes at round end is unmeasured work)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_shape():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emulated", "--windows", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # exactly ONE JSON line
    d = json.loads(lines[0])
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict),
                 ("cpu_baseline", dict)):
        assert k in d and isinstance(d[k], t), (k, type(d.get(k)))
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "solves/s"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = units of all ranks / timed region
    assert abs(d["value"] - 2 * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernels"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["traffic"] is None or rf["traffic"] > 0
    # HBM bytes of a whole step (per-launch PMC traffic x launches per step): present, null only while the PMC summary is for other device code
    assert "traffic_per_step" in rf and (rf["traffic_per_step"] is None or rf["traffic_per_step"] >= rf["traffic"])
    assert set(rf["kernels"]) >= {"ba_prologue_kernel", "ba_accumulate_kernel", "ba_solve_kernel", "ba_marg_kernel"}
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["cores"] == 1 and cb["unit"] == "solves/s"
    assert d["parity"]["windows"] == 2 and d["parity"]["identical_accept_reject_traces"] == 2 and d["parity"]["max_rel_pose_error"] < 1e-4
    assert d["long_run"]["steps"] >= 2 and d["long_run"]["value"] > 0
    hb = d["host_boundary_inclusive"]
    assert hb["resident_sequence_solves_per_s"] and "error" not in hb["resident_sequence"], hb["resident_sequence"]
    fe = d["fe"]
    # (the front end is booked against the roof its SQ counters name -- VALU issue -- when profiles/pmc_latest.json was collected on the
    #  device code of this tree, against HBM as SURVEY 8(d) says otherwise; the HBM figure stays as a side value either way)
    rf_fe = fe["roofline"]
    assert fe["unit"] == "features/s" and fe["tracked_last_step"] > 0
    assert (rf_fe["bound"] == "hbm" and rf_fe["unit"] == "GB/s") or (rf_fe["bound"] == "valu" and rf_fe["hbm"]["unit"] == "GB/s" and 0 < rf_fe["frac"])
    assert d["single_window"]["solve_pipeline_ms"] > 0 and d["single_window_latency_ms"] > 0


def test_sharded_bench_line_runs_end_to_end():
    """`bench.py --config sharded` (BASELINE configs[4]) under the emulator on a small window: the large-window path with its on-demand
    rounds, the per-kernel events, the marginalization of the sharded window, the CPU baseline and parity fields."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emulated", "--config", "sharded", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["unit"] == "solves/s" and d["n_gpus"] == 1 and d["vs_baseline"] is None
    k = d["roofline"]["kernels"]
    assert {"ba_big_schur_kernel", "ba_solve_big_kernel", "ba_big_step_kernel"} <= set(k)
    assert k["ba_solve_big_kernel"]["launches_per_step"] <= 10          # 8 iterations + the round that judges the last candidate: no empty slack rounds
    m = d["marginalization_of_the_sharded_window"]
    assert m["kept_dimension"] > 0 and m["dropped_dimension"] >= 15 and m["ms"] > 0
    assert d["cpu_baseline"]["parity"]["final_cost_rel"] < 1e-6


def test_smoke_entry_point_runs_against_the_emulated_library():
    """__graft_entry__.smoke() (the driver runs it on the GPU before the bench): the same code with the emulated library standing in
    for libvinsgpu.so -- one BA window and one front-end frame pair against the oracles."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conftest
    import __graft_entry__ as graft
    pkg = graft.load_package()
    conftest._build_simt()
    saved = (pkg._lib, pkg.LIB_PATH)
    pkg._lib, pkg.LIB_PATH = ctypes.CDLL(conftest.SIMT_LIB, mode=ctypes.RTLD_LOCAL), conftest.SIMT_LIB
    try:
        graft.smoke()
    finally:
        pkg._lib, pkg.LIB_PATH = saved
