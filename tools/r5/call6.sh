#!/bin/bash
# round 5, sixth GPU call: lean LayerNorm backward after the contraction fix (statistics, tests, A/B); bf16 logits written in fp32 on fresh weights
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=gpurun_out/r5; mkdir -p $OUT
timeout 300 python tools/r5/dbg_ln_lean.py 2>&1 | grep -v amdgpu.ids | tee $OUT/c6_dbg_ln_lean.log
timeout 600 python -m pytest tests/test_stream_kernels_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -q -x -k "layernorm or sandwich or ln_ or train_steps or golden or recompute" > $OUT/c6_tests.log 2>&1; tail -3 $OUT/c6_tests.log
bash tools/evidence.sh bench-ab COGV_LN_BWD_LEAN 0 1 2>&1 | grep -v "^    " | tail -8
cp gpurun_out/ev/bench_ab_COGV_LN_BWD_LEAN.log $OUT/c6_bench_ln_bwd_lean_ab.log
timeout 900 python -m pytest tests/test_depth_parity_gpu.py -q -s -k "test_logits and dtype1" 2>&1 | grep -E "rel-L2|passed|failed" | tee $OUT/c6_bf16_logits_fp32_out.log
