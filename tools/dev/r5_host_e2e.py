#!/usr/bin/env python3
"""Round 5: the host-array round trip (updateData(double*...) -> solve -> solution) of the headline batch, pageable and pinned, for the
current EICOS_COPY_THREADS.  usage: python tools/dev/r5_host_e2e.py"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench, eicos_amd
pat, sets = eicos_amd.read_problem(os.path.join(bench.ROOT, "tests", "golden", "MPC02.epb"))
r = bench.host_e2e(pat, sets, 1024, 0, 1.0, steps=5, warmup=1)
print("copy threads", os.environ.get("EICOS_COPY_THREADS", "default"), {k: (round(v["value"]), round(v["ms_per_step"], 2), round(v["update_ms"], 2), v["update_path"]) for k, v in r.items() if isinstance(v, dict)})
