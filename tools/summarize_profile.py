#!/usr/bin/env python3
"""Summarise rocprofv3 output of tools/dev/profile_r1.sh into profiles/<tag>_*.{csv,md}.

usage: python tools/summarize_profile.py gpurun_out/prof_r2 profiles/r02_v2 ["MPC02 batch=1024" [prefix]]
The optional third argument names the workload the passes were taken on; together with a hash of the kernel sources
it is recorded in the JSON so that bench.py only quotes `roofline.traffic` from a summary of the SAME code and workload.
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
workload = sys.argv[3] if len(sys.argv) > 3 else "MPC02 batch=1024"
prefix = sys.argv[4] if len(sys.argv) > 4 else ""  # sub-run inside src: "" (headline), "soc_", "tile_"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import kernel_source_hash
os.makedirs(os.path.dirname(dst) or ".", exist_ok=True)
stats = glob.glob(os.path.join(src, prefix + "stats", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], dst + "_kernel_stats.csv")
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, prefix + "pmc_*", "*", "*counter_collection.csv")):
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
out["kernel_source_sha256"] = kernel_source_hash()
out["workload"] = workload
lines += ["", f"workload: {workload}", f"kernel sources sha256[:16]: {out['kernel_source_sha256']}",
          "note: the read factor 2 is calibrated on unit-stride streams only (r01_fetch_calibration.md); k_solve also issues scattered",
          "8-byte gathers, for which the factor is between 1 and 2 -- read traffic is an upper bound."]
open(dst + "_pmc.md", "w").write("\n".join(lines) + "\n")
json.dump(out, open(dst + "_pmc.json", "w"), indent=1)
print("\n".join(lines))
if stats:
    print(open(stats[0]).read())
