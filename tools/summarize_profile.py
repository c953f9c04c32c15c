#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/dev/profile_r1.sh into profiles/<tag>_*.{csv,md}.

usage: python tools/summarize_profile.py gpurun_out/prof_r1 profiles/r01
Per-launch averages of the PMC counters for the solve kernel; FETCH_SIZE/WRITE_SIZE are reported in
bytes (rocprofv3 reports KiB).  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) FETCH_SIZE counts
128-B requests of streaming reads at 64 B; tools/dev/calib_fetch.hip (profiles/r01_fetch_calibration.md) confirms
exactly 1/2 for the 4, 8 and 16 B/lane unit-stride reads this kernel issues and exact WRITE_SIZE for 8 B/lane
stores, so traffic = 2*FETCH_SIZE + WRITE_SIZE (KiB).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
stats = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], dst + "_kernel_stats.csv")
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if "k_solve" in r["Kernel_Name"]:
            per[r["Dispatch_Id"]][r["Counter_Name"]] = per[r["Dispatch_Id"]].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    for d in per.values():
        for k, v in d.items():
            agg[k].append(v)
out = {k: sum(v) / len(v) for k, v in agg.items()}
lines = ["# PMC per k_solve launch (average over %d launches)" % max((len(v) for v in agg.values()), default=0), ""]
for k in sorted(out):
    lines.append(f"{k}: {out[k]:.6g}")
if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
    rd, wr = out["FETCH_SIZE"] * 1024, out["WRITE_SIZE"] * 1024
    lines += ["", f"HBM read bytes (2*FETCH_SIZE*1024, calibrated): {2*rd:.4g}  (raw counter: {rd:.4g})",
              f"HBM write bytes (WRITE_SIZE*1024): {wr:.4g}", f"traffic per launch: {2*rd+wr:.4g} bytes"]
    out["traffic_bytes"] = 2 * rd + wr
if "SQ_WAVE_CYCLES" in out:
    lines.append(f"wait fraction SQ_WAIT_ANY/SQ_WAVE_CYCLES: {out['SQ_WAIT_ANY']/out['SQ_WAVE_CYCLES']:.3f}")
if "TCC_HIT_sum" in out:
    lines.append(f"L2 hit rate: {out['TCC_HIT_sum']/(out['TCC_HIT_sum']+out['TCC_MISS_sum']):.3f}")
open(dst + "_pmc.md", "w").write("\n".join(lines) + "\n")
json.dump(out, open(dst + "_pmc.json", "w"), indent=1)
print("\n".join(lines))
if stats:
    print(open(stats[0]).read())
