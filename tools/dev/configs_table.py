"""Dev: markdown table from a configs jsonl (tools/dev/r5_configs.sh: one bench.py line per pattern); reads `config.summary.headline`."""
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
print("| config | batch | path, threads | levels | mean iter | OPTIMAL | GPU iter/s | 16-core oracle iter/s | ratio | frac | launch ms (p95 / max instance ms) | code / iter equal |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for o in rows:
    c, s = o["config"], o["config"]["summary"]["headline"]
    w = c["workload"]
    name = w.split(",")[1].strip().split(" pattern")[0] + (" (perturbed)" if "perturbed" in w else "")
    print(f"| {name} | {s['batch']} | {s['path']}, T={c['threads_per_block']} | {c['levels']} | {s['mean_iter']:.1f} | {s['optimal']} | {s['value'] / 1e3:.1f} k | "
          f"{s['cpu'] / 1e3:.1f} k | {s['x_cpu']:.2f}× | {s['frac']:.3f} | {s['kernel_ms']:.2f} ({s['inst_ms_p95']:.2f} / {s['inst_ms_max']:.2f}) | {s['codes_equal']} / {s['iters_equal']} |")
