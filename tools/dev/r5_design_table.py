#!/usr/bin/env python3
"""Round 5: rewrite the numbers of DESIGN.md section 0's table (and the host-array figures of sections 0 / 5) from the committed evidence --
profiles/<tag>_bench.json (the default bench line) and profiles/<tag>*_pmc.json (HBM traffic per launch).  usage: python tools/dev/r5_design_table.py r05_v2"""
import json, os, re, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
tag = sys.argv[1] if len(sys.argv) > 1 else "r05_v2"
b = json.loads(open(os.path.join(ROOT, "profiles", f"{tag}_bench.json")).read())
S = b["config"]["summary"]
tr = {}
for t, k in (("", "headline"), ("_soc", "soc"), ("_tile", "dense_front_b512"), ("_b512", "mpc_b512"), ("_b4096", "mpc_b4096"), ("_afiro", "lp_afiro_b256"),
             ("_bandm", "lp_bandm_b256"), ("_25fv47", "lp_25fv47_b256")):
    tr[k] = json.load(open(os.path.join(ROOT, "profiles", f"{tag}{t}_pmc.json")))["traffic_bytes"] / 1e9
kf = lambda v: f"{v / 1e6:.2f} M" if v >= 1e6 else f"{v / 1e3:.1f} k"
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
lines = s.split("\n")


def patch(prefix, cols):
    for i, l in enumerate(lines):
        if l.startswith(prefix):
            cells = l.split("|")
            for idx, val in cols.items():
                cells[idx] = " " + val + " "
            lines[i] = "|".join(cells)
            return
    raise SystemExit("row not found " + prefix)


ratio = lambda k: tr[k] / S[k]["algo_GB"]
for key, pre in (("headline", "| `headline`:"), ("soc", "| `soc`:"), ("lp_bandm_b256", "| `lp_bandm_b256`"), ("lp_25fv47_b256", "| `lp_25fv47_b256`"),
                 ("mpc_b512", "| `mpc_b512`"), ("mpc_b4096", "| `mpc_b4096`")):
    a = S[key]
    patch(pre, {3: kf(a["value"]), 4: f'{a["frac"]:.3f}', 5: f'{tr[key]:.1f} GB = {ratio(key):.2f} ×' + (" algorithmic" if key == "headline" else ""), 6: f'{a["x_cpu"]:.1f}'})
d = S["dense_front_b512"]
patch("| `dense_front_b512`", {3: kf(d["value"]), 4: f'{d["frac"]:.3f} (`frac_dual` {d["frac_dual"]:.3f})',
                               5: f'{tr["dense_front_b512"]:.1f} GB = {ratio("dense_front_b512"):.2f} × ({tr["dense_front_b512"] / (d["algo_GB"] * d["frac_dual"] / d["frac"]):.2f} × the bytes a dual solve needs)',
                               6: f'{d["x_cpu"]:.1f}'})
a = S["lp_afiro_b256"]
patch("| `lp_afiro_b256`", {3: kf(a["value"]), 4: f'{a["frac"]:.3f}', 6: f'{a["x_cpu"]:.2f}'})
h, cpu = S["host_e2e"], S["headline"]["cpu"]
patch("| `host_e2e`:", {2: "pageable / registered in place / pinned", 3: " / ".join(kf(h[v]["value"]) for v in ("pageable", "registered", "pinned")),
                        6: " / ".join(f'{h[v]["value"] / cpu:.1f}' for v in ("pageable", "registered", "pinned")) + " (the CPU baseline reads host arrays too: this is the like-for-like ratio)"})
s = "\n".join(lines)
s = re.sub(r"\*\*pageable [0-9.]+ k iter/s = 0\.[0-9]+ ×, pinned in place [0-9.]+ k = 0\.[0-9]+ × the device-resident [0-9.]+ k\*\*; `updateData` [0-9.]+ / [0-9.]+ ms",
           f'**pageable {kf(h["pageable"]["value"])} iter/s = {h["pageable"]["x_device_resident"]:.3f} ×, pinned in place {kf(h["registered"]["value"])} = {h["registered"]["x_device_resident"]:.3f} × the device-resident {kf(S["headline"]["value"])}**; `updateData` {h["pageable"]["update_ms"]:.2f} / {h["registered"]["update_ms"]:.2f} ms', s)
s = re.sub(r"Device-resident step [0-9.]+ ms \([0-9.]+ k iter/s\); from pageable host arrays [0-9.]+ ms \([0-9.]+ k = 0\.[0-9]+ ×; `updateData` [0-9.]+ ms of HIP-event",
           f'Device-resident step {b["ms_per_step"]:.1f} ms ({kf(S["headline"]["value"])} iter/s); from pageable host arrays {1024 * S["headline"]["mean_iter"] / h["pageable"]["value"] * 1e3:.1f} ms ({kf(h["pageable"]["value"])} = {h["pageable"]["x_device_resident"]:.3f} ×; `updateData` {h["pageable"]["update_ms"]:.2f} ms of HIP-event', s)
s = re.sub(r"\([0-9.]+ k = 0\.[0-9]+ ×; `updateData` [0-9.]+ ms = 55 GB/s over the link",
           f'({kf(h["pinned"]["value"])} = {h["pinned"]["x_device_resident"]:.3f} ×; `updateData` {h["registered"]["update_ms"]:.2f} ms = 55 GB/s over the link', s)
open(path, "w").write(s)
for l in s.split("\n"):
    if l.startswith(("| `headline`", "| `soc`", "| `dense", "| `lp_", "| `mpc_", "| `host_e2e`")):
        print(l[:200])
