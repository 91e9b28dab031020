#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected SEPARATELY: the TCC block cannot hold both,
MI355X_MICROARCH.md "Counter capacity") into per-kernel HBM bytes per launch.

usage: summarize_pmc.py <fetch_results.db> <write_results.db> <out.json> [label] [build tag]

Corrections, exactly as the guide's HBM section prescribes:
  * FETCH_SIZE / WRITE_SIZE are reported in KB (x1024 B);
  * on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> doubled before use ("fetch_x2");
  * WRITE_SIZE is uncalibrated -> used as reported and flagged.
Values are summed over the XCD instances of a dispatch by rocprofv3 (one row per dispatch) and averaged over the launches
on the grid that carries most of each kernel's work (the 256-window batch launches; the single-window launches of bench.py's
latency line and the few 512-window launches of its two-per-CU pass are left out)."""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    # bench.py launches every kernel on the 256-window batch (thousands of launches), on a single window (its latency line) and,
    # since round 5, on 512 windows (the two-per-CU pass: a few launches with the LARGEST grid).  "Per launch" means per BATCH launch:
    # the grid that carries the most work-items in total (launches x grid size) of each kernel -- the 256-window one.
    q = ("select c.kernel_name, count(*), avg(c.value), min(c.value), max(c.value), c.grid_size from counters_collection c "
         "join (select k, g from (select kernel_name k, grid_size g, count(*) * grid_size w from counters_collection where counter_name = ? "
         "group by kernel_name, grid_size order by w) group by k having w = max(w)) m "
         "on c.kernel_name = m.k and c.grid_size = m.g where c.counter_name = ? group by c.kernel_name")
    return {r[0]: dict(launches=r[1], avg=r[2], min=r[3], max=r[4], grid=r[5]) for r in db.execute(q, (counter, counter))}


def main():
    fetch_db, write_db, out = sys.argv[1:4]
    label = sys.argv[4] if len(sys.argv) > 4 else ""
    build = sys.argv[5] if len(sys.argv) > 5 else ""
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(f) | set(w)):
        short = name.split("(")[0].split("<")[0].strip()
        short = short.split()[-1] if " " in short else short
        fk = f.get(name, {}).get("avg", 0.0)
        wk = w.get(name, {}).get("avg", 0.0)
        kernels[short] = dict(
            launches=f.get(name, w.get(name))["launches"], grid_size=f.get(name, w.get(name))["grid"],
            fetch_size_kb_reported=fk, write_size_kb_reported=wk,
            fetch_bytes_x2=2.0 * fk * 1024.0, write_bytes=wk * 1024.0,
            hbm_bytes_per_launch=2.0 * fk * 1024.0 + wk * 1024.0)
    json.dump(dict(label=label, build=build, note="FETCH_SIZE doubled (gfx950 128-B requests tallied at 64 B); WRITE_SIZE uncalibrated; "
                                     "KB = 1024 B; separate --pmc passes of `python bench.py`", kernels=kernels),
              open(out, "w"), indent=1)
    print(f"{'kernel':<28}{'launches':>9}{'fetch_KB':>14}{'write_KB':>14}{'HBM_MB/launch':>16}")
    for k, v in kernels.items():
        print(f"{k[:27]:<28}{v['launches']:>9}{v['fetch_size_kb_reported']:>14.1f}{v['write_size_kb_reported']:>14.1f}"
              f"{v['hbm_bytes_per_launch'] / 1e6:>16.3f}")


if __name__ == "__main__":
    main()
