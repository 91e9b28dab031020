#!/usr/bin/env python3
"""Per-kernel averages of arbitrary rocprofv3 --pmc counters (rocpd sqlite results).
usage: summarize_counters.py [--merge-into pmc.json] <results.db> [<results2.db> ...]
   -> table on stdout (one row per kernel, one column per counter); with --merge-into the derived SQ fractions (valu_issue_frac,
      waves_parked_frac, mfma_busy_frac, lds_conflict_frac) are added to the kernel entries of that PMC summary (bench.py reads them)"""
import json
import sqlite3
import sys
from collections import defaultdict

args = sys.argv[1:]
merge = None
if args and args[0] == "--merge-into":
    merge, args = args[1], args[2:]
vals = defaultdict(dict)
names = []
for path in args:
    db = sqlite3.connect(path)
    q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
    for k, c, v, n in db.execute(q):
        short = k.split("(")[0].split("<")[0].strip().split()[-1]
        vals[short][c] = v
        vals[short]["launches"] = n
        if c not in names:
            names.append(c)
print(f"{'kernel':<26}" + "".join(f"{n[:22]:>24}" for n in names))
for k in sorted(vals):
    print(f"{k[:25]:<26}" + "".join(f"{vals[k].get(n, float('nan')):>24.4g}" for n in names))
print()
print("derived (per launch, summed over the chip unless noted):")
derived = defaultdict(dict)
for k in sorted(vals):
    v = vals[k]
    out = []
    if v.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
        derived[k]["mfma_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] * 1024)
    if v.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in v:
        derived[k]["lds_conflict_frac"] = v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]
    if v.get("SQ_WAVE_CYCLES") and "SQ_WAIT_ANY" in v:
        derived[k]["waves_parked_frac"] = v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"]
    if v.get("SQ_WAVE_CYCLES") and "SQ_ACTIVE_INST_VALU" in v:
        derived[k]["valu_issue_frac"] = v["SQ_ACTIVE_INST_VALU"] / v["SQ_WAVE_CYCLES"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v and v["GRBM_GUI_ACTIVE"]:
        out.append(f"MfmaUtil = MFMA_BUSY / (GUI_ACTIVE * 1024 SIMDs) = {100 * v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] * 1024):.3f} %")
    if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"):
        out.append(f"LDS bank-conflict cycles / LDS active cycles = {100 * v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.1f} %")
    if "SQ_WAIT_ANY" in v and v.get("SQ_WAVE_CYCLES"):
        out.append(f"waves parked (s_waitcnt / barrier) {100 * v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.1f} % of wave-cycles")
    if "SQ_ACTIVE_INST_VALU" in v and v.get("SQ_WAVE_CYCLES"):
        out.append(f"VALU issue {100 * v['SQ_ACTIVE_INST_VALU'] / v['SQ_WAVE_CYCLES']:.1f} % of wave-cycles")
    if out:
        print(f"  {k}: " + "; ".join(out))

if merge:
    with open(merge) as f:
        j = json.load(f)
    for k, d in derived.items():
        j.setdefault("kernels", {}).setdefault(k, {}).update(d)
    j["sq_note"] = ("valu_issue_frac = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, waves_parked_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES, mfma_busy_frac = "
                    "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs), lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; own --pmc pass")
    with open(merge, "w") as f:
        json.dump(j, f, indent=1)
