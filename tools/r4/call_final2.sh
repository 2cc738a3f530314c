#!/bin/bash
# round 4, second session: the WHOLE GPU suite at HEAD (with the depth file), smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
( time timeout 1500 python -m pytest tests -m gpu -q ) > gpurun_out/r4/final2_gpu_tests.log 2>&1
tail -6 gpurun_out/r4/final2_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r4/final2_smoke.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r4/final2_bench.json 2> gpurun_out/r4/final2_bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4/final2_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype','mfma_roofline_frac_end_to_end')}, d['roofline']['achieved'], d['roofline']['frac'])
print('bf16', {k:d['bf16_leg'][k] for k in ('value','ms_per_step','mfma_roofline_frac_end_to_end')})
P
