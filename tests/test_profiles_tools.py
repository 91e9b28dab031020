"""profiles/summarize_pmc.py: "per launch" must mean per BATCH launch -- the grid that carries most of a kernel's work -- not the largest
grid (bench.py's two-per-CU pass launches every kernel a few times on 512 windows) and not the single-window launches of its latency
line.  (Round 6: the 512-window launches had silently doubled every figure of profiles/pmc_latest.json.)"""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _db(path, counter, rows):
    db = sqlite3.connect(path)
    db.execute("create table counters_collection (kernel_name text, value real, grid_size int, counter_name text)")
    db.executemany("insert into counters_collection values (?,?,?,?)", [(k, v, g, counter) for (k, v, g) in rows])
    db.commit()
    db.close()


def test_pmc_summary_takes_the_batch_grid(tmp_path):
    rows = [("k1(int)", 100.0, 131072)] * 50 + [("k1(int)", 200.0, 262144)] * 4 + [("k1(int)", 1.0, 512)] * 20 + [("k2", 7.0, 512)] * 3
    f, w, out = str(tmp_path / "f.db"), str(tmp_path / "w.db"), str(tmp_path / "o.json")
    _db(f, "FETCH_SIZE", rows)
    _db(w, "WRITE_SIZE", rows)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "summarize_pmc.py"), f, w, out, "label", "tag"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    d = json.load(open(out))
    k1 = d["kernels"]["k1"]
    assert k1["grid_size"] == 131072 and k1["launches"] == 50
    # FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B), KB = 1024 B, WRITE_SIZE as reported
    assert abs(k1["hbm_bytes_per_launch"] - (2 * 100.0 + 100.0) * 1024) < 1e-6
    assert d["kernels"]["k2"]["launches"] == 3 and d["build"] == "tag"
