#!/bin/bash
# round 4, GPU call 7: whole GPU suite except the full-depth oracle tests (run in call 1), then the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q --ignore=tests/test_depth_parity_gpu.py > gpurun_out/r4/c7_tests.log 2>&1
tail -5 gpurun_out/r4/c7_tests.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r4/c7_bench.json 2> gpurun_out/r4/c7_bench.err
tail -25 gpurun_out/r4/c7_bench.err; cat gpurun_out/r4/c7_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype','mfma_roofline_frac_end_to_end')}, d['roofline']['achieved'], d['roofline']['frac'], d['roofline'].get('sampled_steps'))
print('bf16', {k:d['bf16_leg'][k] for k in ('value','ms_per_step','mfma_roofline_frac_end_to_end')})
print(d['config']['logits_rel_l2_vs_fp32_reference'].get('measured'), d['bf16_leg']['logits_rel_l2_vs_fp32_reference'].get('measured'))
print(d.get('cpu_baseline'))
"
