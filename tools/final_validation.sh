#!/bin/bash
# the round's last GPU call: whole suite + smoke + the driver's command at HEAD, then the small configurations' lines
bash tools/evidence.sh suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/ev
for c in cogview-tiny-18M vqvae; do
  timeout 900 python bench.py --config $c --steps 20 --warmup 5 > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  python - $OUT/bench_$c.json <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"][:40], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 3), "ms/step", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline_reference", {}).get("value"))
P
done
