#!/bin/bash
# Round-3 evidence for profiles/: GPU test log, smoke, default bench line (bf16 + fp16 leg, measured logits error), rocprofv3
# kernel statistics of the 4B bench command, HBM-side traffic of the GEMM family (two PMC passes), 336M and VQ-VAE lines, decode.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r3final; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^\.\.\." | tail -60 > $O/gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
python bench.py > $O/bench_4B.json 2> $O/bench_4B.err
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_4B -- python $R/bench.py --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/prof_4B.log 2>&1)
cp $(ls $O/prof_4B/*/*kernel_stats.csv | head -1) $O/kernel_stats_4B.csv
python bench.py --config cogview-small-336M --dtype bf16 > $O/bench_336M.json 2> $O/bench_336M.err
python bench.py --config vqvae > $O/bench_vqvae.json 2> $O/bench_vqvae.err
MB_DECODE_GRAPH_ONLY=1 python tools/mb_decode.py 2>&1 | grep GraphDecoder > $O/decode.log
bash $R/tools/collect_traffic.sh --dtype bf16 > $O/traffic.log 2>&1
cp $R/gpurun_out/gemm_traffic.json $O/gemm_traffic.json 2>/dev/null
find $O $R/gpurun_out/traffic_FETCH_SIZE $R/gpurun_out/traffic_WRITE_SIZE -name "*.db" -delete 2>/dev/null
rm -rf $O/prof_4B/*/*kernel_trace.csv 2>/dev/null
tail -3 $O/gpu_tests.log; tail -1 $O/smoke.log; cat $O/decode.log
for f in bench_4B bench_336M bench_vqvae; do python - <<PY
import json
s=open("$O/$f.json").read(); s=s[s.index('{"metric"'):]; d=json.loads(s)
print("$f", round(d["value"],1), round(d["ms_per_step"],1), round(d["mfma_roofline_frac_end_to_end"],4), round(d["roofline"]["achieved"],1), d.get("fp16_leg",{}).get("value"), d["config"].get("logits_rel_l2_vs_fp32_reference"))
PY
done
head -8 $O/kernel_stats_4B.csv | cut -c1-140; tail -4 $O/traffic.log
