#!/bin/bash
# round 4, second session: GPU suite at HEAD without the 6-minute depth file, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
( time timeout 900 python -m pytest tests -m gpu -q --ignore=tests/test_depth_parity_gpu.py ) > gpurun_out/r4/c14_gpu_tests.log 2>&1
tail -8 gpurun_out/r4/c14_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r4/c14_smoke.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r4/c14_bench.json 2> gpurun_out/r4/c14_bench.err
tail -c 600 gpurun_out/r4/c14_bench.err
python - <<'P'
import json
d=json.loads(open('gpurun_out/r4/c14_bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','dtype','mfma_roofline_frac_end_to_end')}, d['roofline']['achieved'], d['roofline']['frac'])
print('bf16', {k:d['bf16_leg'][k] for k in ('value','ms_per_step','mfma_roofline_frac_end_to_end')})
P
