#!/bin/bash
# round 5, first GPU call: (1) the new weight-gradient queue -- bit-identity test, A/B of the 4B step with it on / off;
# (2) decode: non-temporal weight / cache loads A/B (build/ab/libcogview_nont.so = COGV_DECODE_NT=0), chain row limit A/B;
# (3) the yardstick library's kernel names on the 4B shapes; (4) counters of the attention kernels.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=gpurun_out/r5; mkdir -p $OUT
echo "== tests"; ( time timeout 900 python -m pytest tests/test_model_gpu.py tests/test_stream_kernels_gpu.py -q -x --durations=12 ) > $OUT/c1_tests.log 2>&1; tail -25 $OUT/c1_tests.log
echo "== decode tests"; timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemv or decode" > $OUT/c1_tests_decode.log 2>&1; tail -3 $OUT/c1_tests_decode.log
echo "== bench A/B: weight-gradient queue"
for rep in 1 2; do
  for q in 1 0; do
    COGV_WGRAD_QUEUE=$q timeout 600 python bench.py --steps 12 --warmup 3 --no-second-dtype --no-cpu-baseline > $OUT/c1_bench_q${q}_$rep.json 2> $OUT/c1_bench_q${q}_$rep.err
    python - <<P
import json
d=json.loads(open("$OUT/c1_bench_q${q}_$rep.json").read().strip().splitlines()[-1])
print("queue=$q rep $rep", round(d["value"]), "tok/s", round(d["ms_per_step"],2), "ms", "gemm", round(d["roofline"]["achieved"],1), {k:(round(v.get("frac",0),3), round(v.get("share_of_step_time",0),3)) for k,v in d["roofline"]["by_family"].items() if isinstance(v,dict) and "frac" in v})
P
  done
done 2>&1 | tee $OUT/c1_bench_queue_ab.log
grep -A40 "launches by kernel family" $OUT/c1_bench_q1_2.err | head -60
echo "== decode nt A/B"
for b in 1 8; do
  for lib in "" build/ab/libcogview_nont.so; do
    echo "-- batch $b lib=${lib:-default(nt)}"
    COGVIEW_HIP_LIB=$lib MB_DECODE_BATCH=$b MB_DECODE_GRAPH_ONLY=1 timeout 300 python tools/mb_decode.py 2>&1 | grep "captured"
  done
done 2>&1 | tee $OUT/c1_decode_nt_ab.log
echo "== decode chain rows A/B"
MB_ROWS_BATCHES=8 bash tools/r5/call1_decode_rows.sh
echo "== hipBLASLt names"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/hbl -- python $R/tools/probes/hipblaslt_names.py > $R/$OUT/c1_hipblaslt.log 2>&1
cd $R; cat $OUT/c1_hipblaslt.log | grep hipBLASLt
python - <<'P'
import csv, glob
for f in glob.glob("gpurun_out/r5/hbl/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "Cijk" in r["Name"] or "gemm" in r["Name"].lower():
            print(r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Name"][:400])
P
echo "== attention PMC"
bash tools/r5/pmc_attn.sh
