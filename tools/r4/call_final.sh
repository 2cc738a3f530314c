#!/bin/bash
# round 4: the whole GPU suite at HEAD, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
git rev-parse HEAD > gpurun_out/r4/final_head.txt 2>/dev/null
( time timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r4/final_gpu_tests.log 2>&1
tail -6 gpurun_out/r4/final_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r4/final_smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r4/final_bench.json 2> gpurun_out/r4/final_bench.err
python -c "
import json
d=json.loads(open('gpurun_out/r4/final_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype','mfma_roofline_frac_end_to_end')}, d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['traffic'])
print('bf16', {k:d['bf16_leg'][k] for k in ('value','ms_per_step','mfma_roofline_frac_end_to_end')})
print(d['config']['logits_rel_l2_vs_fp32_reference'].get('measured'), d['bf16_leg']['logits_rel_l2_vs_fp32_reference'].get('measured'), d['cpu_baseline']['value'])
"
