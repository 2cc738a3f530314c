#!/bin/bash
# Round-end evidence for profiles/: GPU test log, rocprofv3 kernel statistics of the bench command, HBM-side traffic of
# the GEMM family (two PMC passes), the default bench line and the 336M line.  Run on the GPU box from the repo root.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $R/gpurun_out/gpu_tests_final.log
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_final.log 2>&1)
cp $(ls $R/gpurun_out/prof_final/*/*kernel_stats.csv | head -1) $R/gpurun_out/kernel_stats_final.csv
bash $R/tools/collect_traffic.sh > $R/gpurun_out/traffic_final.log 2>&1
python $R/bench.py > $R/gpurun_out/bench_final.json 2> $R/gpurun_out/bench_final.err
python $R/bench.py --config cogview-small-336M > $R/gpurun_out/bench_336M_final.json 2> $R/gpurun_out/bench_336M_final.err
tail -3 $R/gpurun_out/gpu_tests_final.log; head -8 $R/gpurun_out/kernel_stats_final.csv | cut -c1-160; tail -3 $R/gpurun_out/traffic_final.log; tail -c 600 $R/gpurun_out/bench_final.json | head -c 300; echo; tail -c 1500 $R/gpurun_out/bench_336M_final.json | head -c 400
