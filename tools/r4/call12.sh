#!/bin/bash
# round 4, GPU call 12: rocprofv3 kernel statistics of the fp16 step + GEMM HBM-side traffic (two PMC passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
(cd /tmp; export TMPDIR=/tmp; rm -rf $R/gpurun_out/r4/prof; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/prof -- python $R/bench.py --dtype fp16 --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/r4/prof.log 2>&1)
cp $(ls $R/gpurun_out/r4/prof/*/*kernel_stats.csv | head -1) $R/gpurun_out/r4/kernel_stats_fp16.csv
rm -rf $R/gpurun_out/r4/prof
bash tools/collect_traffic.sh --dtype fp16 > gpurun_out/r4/traffic.log 2>&1
cp gpurun_out/gemm_traffic.json gpurun_out/r4/gemm_traffic_fp16.json 2>/dev/null
rm -rf gpurun_out/traffic_FETCH_SIZE gpurun_out/traffic_WRITE_SIZE
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4/kernel_stats_fp16.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print(f"{float(r['TotalDurationNs'])/tot*100:6.2f}% {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.1f}us  {r['Name'][:100]}")
PY
tail -12 gpurun_out/r4/traffic.log
