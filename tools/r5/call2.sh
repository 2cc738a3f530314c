#!/bin/bash
# round 5, second GPU call: (1) attention LDS ring of two stages (build/ab/libcogview_attn2.so) against three: tests, kernel
# times, step time; (2) decode: non-temporal loads with the chain limited to 4 rows at batch 8, batches 2 / 4; (3) 336M line.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=gpurun_out/r5; mkdir -p $OUT
echo "== attention tests, two-stage ring"
COGVIEW_HIP_LIB=build/ab/libcogview_attn2.so timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or sparse" > $OUT/c2_tests_attn2.log 2>&1; tail -3 $OUT/c2_tests_attn2.log
echo "== attention tests, default (three-stage) build"
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "attention or sparse" > $OUT/c2_tests_attn3.log 2>&1; tail -3 $OUT/c2_tests_attn3.log
echo "== attention kernel times"
for rep in 1 2; do
  for lib in "" build/ab/libcogview_attn2.so; do
    COGVIEW_HIP_LIB=$lib timeout 300 python tools/r5/mb_attn_train.py 2>&1 | grep '"rep": [12]'
  done
done 2>&1 | tee $OUT/c2_attention_stages_ab.log
echo "== bench A/B: attention ring stages"
for rep in 1 2; do
  for lib in "" build/ab/libcogview_attn2.so; do
    COGVIEW_HIP_LIB=$lib timeout 600 python bench.py --steps 12 --warmup 3 --no-second-dtype --no-cpu-baseline > $OUT/c2_bench_tmp.json 2> $OUT/c2_bench_tmp.err
    python - <<P
import json
d=json.loads(open("$OUT/c2_bench_tmp.json").read().strip().splitlines()[-1])
f=d["roofline"]["by_family"]
print("lib=${lib:-default} rep $rep", round(d["value"]), "tok/s", round(d["ms_per_step"],2), "ms", "attention", round(f["attention"]["achieved"],1), "TF", round(f["attention"]["share_of_step_time"],4), {k:v["avg_ms"] for k,v in f["by_launch"].items() if k.startswith("attention")})
P
  done
done 2>&1 | tee $OUT/c2_bench_attn_stages_ab.log
echo "== decode"
for cfg in "1 8" "8 4" "8 4 build/ab/libcogview_nont.so" "2 8" "4 8" "2 8 build/ab/libcogview_nont.so" "4 8 build/ab/libcogview_nont.so"; do
  set -- $cfg
  echo "-- batch $1 chain rows <= $2 lib=${3:-default(nt)}"
  COGVIEW_HIP_LIB=$3 COGV_DECODE_CHAIN_MAX_ROWS=$2 MB_DECODE_BATCH=$1 MB_DECODE_GRAPH_ONLY=1 timeout 300 python tools/mb_decode.py 2>&1 | grep "captured"
done 2>&1 | tee $OUT/c2_decode_ab.log
echo "== 336M"
timeout 600 python bench.py --config cogview-small-336M --steps 20 --warmup 5 --no-cpu-baseline --no-parity > $OUT/c2_bench_336M.json 2> $OUT/c2_bench_336M.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r5/c2_bench_336M.json").read().strip().splitlines()[-1])
print("336M", round(d["value"]), d["ms_per_step"], d["mfma_roofline_frac_end_to_end"], d["roofline"]["achieved"], "bf16", d.get("bf16_leg",{}).get("value"))
P
grep -A30 "launches by kernel family" gpurun_out/r5/c2_bench_336M.err | head -34
