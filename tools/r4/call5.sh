#!/bin/bash
# round 4, GPU call 5: LayerNorm backward prefetch distance A/B (COGV_LN_BWD_PF), LN tests, idle-gap trace of the 4B step
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
for pf in 1 2 1 2; do
  echo "COGV_LN_BWD_PF=$pf" >> gpurun_out/r4/c5_ln_pf.log
  COGV_LN_BWD_PF=$pf timeout 300 python tools/r3/mb_ln_stream.py 2>/dev/null | grep '"h": 2560' >> gpurun_out/r4/c5_ln_pf.log
done
cat gpurun_out/r4/c5_ln_pf.log
COGV_LN_BWD_PF=2 timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_stream_kernels_gpu.py -m gpu -x -q -k "ln or layernorm or sandwich" 2>&1 | tail -2
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/r4/trace4b
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r4/trace4b -- python $R/bench.py --dtype fp16 --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 2 > $R/gpurun_out/r4/c5_trace_bench.json 2> $R/gpurun_out/r4/c5_trace_bench.err
cd $R
python tools/r4/idle_gaps.py gpurun_out/r4/trace4b > gpurun_out/r4/c5_idle_gaps.txt 2>&1
cat gpurun_out/r4/c5_idle_gaps.txt
rm -rf gpurun_out/r4/trace4b
