"""bench lines of the other BASELINE.json configurations (gpurun_out/b_<cfg>.log) -> profiles/r03_other_configs.md"""
import json, sys
out = ["# Other BASELINE.json configurations on one MI355X (round 3) — `python bench.py --config <cfg> --steps 10 --warmup 3 --bam-reads -1`\n",
       "Parity for these configurations is tested against the oracle / goldens on subsamples (`tests/`); these lines are throughput only.\n",
       "| config | workload | reads/s | ms/step | scan launch ms | roofline frac | CPU port reads/s (sample) |", "|---|---|---|---|---|---|---|"]
lines = []
for c in ("cfg1", "cfg2", "cfg5"):
    for line in open("gpurun_out/b_%s.log" % c):
        if line.startswith("{"):
            d = json.loads(line)
            cb = d.get("cpu_baseline", {})
            out.append("| %s | %s | %.3g | %.1f | %.3f | %.3f | %s |" % (
                c, d["config"]["workload"].split(":")[1].split(", full")[0].strip(), d["value"], d["ms_per_step"],
                d["roofline"]["launch_ms"], d["roofline"]["frac"], ("%.0f (%s)" % (cb["value"], cb["sample"].split(",")[0])) if cb else "-"))
            lines += ["", "    " + line.strip()]
open(sys.argv[1] if len(sys.argv) > 1 else "profiles/r03_other_configs.md", "w").write("\n".join(out + lines) + "\n")
print("\n".join(o for o in out if o.startswith("|")))
