#!/bin/bash
# round 5, fourth GPU call: the lean dropout-replay LayerNorm backward (COGV_LN_BWD_LEAN) -- test, A/B in the 4B step -- and the
# driver's own command (both dtypes, parity legs, cpu_baseline).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=gpurun_out/r5; mkdir -p $OUT
echo "== tests"
timeout 600 python -m pytest tests/test_stream_kernels_gpu.py tests/test_kernels_gpu.py -q -x -k "layernorm or sandwich or ln_" > $OUT/c4_tests.log 2>&1; tail -3 $OUT/c4_tests.log
echo "== bench A/B: lean LayerNorm backward"
for rep in 1 2; do
  for lean in 0 1; do
    COGV_LN_BWD_LEAN=$lean timeout 600 python bench.py --steps 12 --warmup 3 --no-second-dtype --no-cpu-baseline > $OUT/c4_bench_tmp.json 2> $OUT/c4_bench_tmp.err
    python - <<P
import json
d=json.loads(open("$OUT/c4_bench_tmp.json").read().strip().splitlines()[-1])
f=d["roofline"]["by_family"]
print("COGV_LN_BWD_LEAN=$lean rep $rep", round(d["value"]), "tok/s", round(d["ms_per_step"],2), "ms", "layernorm", round(f["layernorm"]["achieved"]), "GB/s share", round(f["layernorm"]["share_of_step_time"],4), {k.replace("layernorm ",""):(v["avg_ms"], v["gbytes_per_s"]) for k,v in f["by_launch"].items() if "dropout replay" in k})
P
  done
done 2>&1 | tee $OUT/c4_bench_ln_bwd_lean_ab.log
echo "== the driver's command"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/c4_bench_default.json 2> $OUT/c4_bench_default.err
tail -c 6000 $OUT/c4_bench_default.json | head -c 6000
grep -E "logits rel-L2|real" $OUT/c4_bench_default.err
