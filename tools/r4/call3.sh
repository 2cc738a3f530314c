#!/bin/bash
# round 4, GPU call 3: tests of the advisor fixes + keep bits in the model, dK/dV async-read A/B, bench with / without per-GEMM events
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_stream_kernels_gpu.py tests/test_model_gpu.py tests/test_checkpoint_gpu.py -m gpu -x -q > gpurun_out/r4/c3_tests.log 2>&1
tail -3 gpurun_out/r4/c3_tests.log
for lib in new kbsync new kbsync; do
  if [ $lib = new ]; then unset COGVIEW_HIP_LIB; else export COGVIEW_HIP_LIB=$R/build/ab/libcogview_$lib.so; fi
  echo "lib=$lib" >> gpurun_out/r4/c3_attn_ab.log
  timeout 300 python tools/r4/mb_attn_keepbits.py 2>/dev/null | grep '"H": 40' >> gpurun_out/r4/c3_attn_ab.log
done
unset COGVIEW_HIP_LIB
cat gpurun_out/r4/c3_attn_ab.log
for t in "" "--no-kernel-timing" "" "--no-kernel-timing"; do
  timeout 600 python bench.py --dtype fp16 --no-cpu-baseline --steps 10 --warmup 3 $t > gpurun_out/r4/c3_b.json 2> gpurun_out/r4/c3_b.err
  python - <<PY >> gpurun_out/r4/c3_bench_events_ab.log
import json
d=json.loads(open("gpurun_out/r4/c3_b.json").read().strip().splitlines()[-1])
print("timing='$t'", round(d["value"],1), "tok/s", round(d["ms_per_step"],2), "ms", d.get("roofline",{}).get("achieved"))
PY
done
cat gpurun_out/r4/c3_bench_events_ab.log; tail -20 gpurun_out/r4/c3_b.err
