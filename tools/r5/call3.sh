#!/bin/bash
# round 5, third GPU call: (1) restructured depth tests (time), tile guard, new stream-kernel tests; (2) Sandwich-LN row order A/B
# in the 4B step; (3) LDS counters of the attention kernels; (4) decode at the new defaults + combine-in-prologue under capture.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=gpurun_out/r5; mkdir -p $OUT
echo "== tests"
( time timeout 1200 python -m pytest tests/test_stream_kernels_gpu.py tests/test_gemm_bench_scale_gpu.py tests/test_depth_parity_gpu.py -q -x -s --durations=12 ) > $OUT/c3_tests.log 2>&1; grep -E "rel-L2|passed|failed|real|^[0-9.]+s " $OUT/c3_tests.log | tail -40
echo "== bench A/B: Sandwich-LN row order"
for rep in 1 2; do
  for rv in 0 3 15; do
    COGV_LN_REVERSE=$rv timeout 600 python bench.py --steps 12 --warmup 3 --no-second-dtype --no-cpu-baseline > $OUT/c3_bench_tmp.json 2> $OUT/c3_bench_tmp.err
    python - <<P
import json
d=json.loads(open("$OUT/c3_bench_tmp.json").read().strip().splitlines()[-1])
f=d["roofline"]["by_family"]
print("COGV_LN_REVERSE=$rv rep $rep", round(d["value"]), "tok/s", round(d["ms_per_step"],2), "ms", "layernorm", round(f["layernorm"]["achieved"]), "GB/s share", round(f["layernorm"]["share_of_step_time"],4), {k.replace("layernorm ",""):(v["avg_ms"], v["gbytes_per_s"]) for k,v in f["by_launch"].items() if k.startswith("layernorm") and v["launches"]>10})
P
  done
done 2>&1 | tee $OUT/c3_bench_ln_reverse_ab.log
echo "== attention LDS counters"
cd /tmp; export TMPDIR=/tmp
rm -rf $R/$OUT/pmc_attn_c
PMC_ATTN_DTYPE=fp16 timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
  -d $R/$OUT/pmc_attn_c -- python $R/tools/pmc_attn.py > $R/$OUT/pmc_attn_c.log 2>&1
cd $R
python tools/pmc_report.py $OUT/pmc_attn_c attn 2>&1 | tee $OUT/c3_attention_lds_pmc.txt
echo "== decode"
for cfg in "1 -" "1 1" "2 -" "4 -" "8 -"; do
  set -- $cfg
  echo "-- batch $1 COGV_DECODE_FUSE_COMBINE=$2"
  if [ "$2" = "-" ]; then MB_DECODE_BATCH=$1 MB_DECODE_GRAPH_ONLY=1 timeout 300 python tools/mb_decode.py 2>&1 | grep "captured"
  else COGV_DECODE_FUSE_COMBINE=$2 MB_DECODE_BATCH=$1 MB_DECODE_GRAPH_ONLY=1 timeout 300 python tools/mb_decode.py 2>&1 | grep "captured"; fi
done 2>&1 | tee $OUT/c3_decode.log
