"""Dev: markdown table of DESIGN.md section 5.1 from a configs jsonl (tools/dev/configs_r3.sh) + the default bench line."""
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.strip().startswith("{")]
head = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]) if len(sys.argv) > 2 else None
def line(name, o, cfg, roof, cpu):
    tr = roof.get("traffic")
    return (f"| {name} | {cfg['batch_per_gpu'] if 'batch_per_gpu' in cfg else cfg['instances']} | {cfg['factor_path']}{', slabs in LDS' if cfg.get('lds_resident') else ''}, T={cfg['threads_per_block']} | {cfg['levels']} | "
            f"{cfg['mean_iter']:.1f} | {cfg['optimal']} | {o['value']/1e3:.1f} k | {cpu['value']/1e3:.1f} k | {o['value']/cpu['value']:.2f}× | {roof['frac']:.3f} | "
            f"{'%.1f GB = %.2f×' % (tr/1e9, tr/roof['algorithmic_bytes_per_launch']) if tr else '—'} | {cpu['exitcodes_equal']}/{cpu['iters_equal']} of {cpu['instances_compared']} |")
print("| config | batch | path | levels | mean iter | OPTIMAL | GPU iter/s | CPU iter/s | ratio | frac | HBM traffic per launch (PMC) | code / iter equal |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
if head:
    print(line("MPC02 pattern, generated feasible (config 1, headline)", head, head["config"], head["roofline"], head["cpu_baseline"]))
    s = head["soc"]; print(line("MPC-SOC (332 cones of dim 3), in the headline line", s, s, s["roofline"], s["cpu_baseline"]))
for o in rows:
    w = o["config"]["workload"]
    name = w.split(",")[1].strip().split(" pattern")[0] + (" (perturbed)" if "perturbed" in w else "")
    print(line(name, o, o["config"], o["roofline"], o["cpu_baseline"]))
