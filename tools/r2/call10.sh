#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c10; mkdir -p $O
cd $R
python -m pytest tests/test_kernels_gpu.py tests/test_vqvae_gpu.py -m gpu -q 2>&1 | tail -5
python bench.py --config vqvae --steps 5 --warmup 2 --no-cpu-baseline > $O/v.json 2> $O/v.err
python - <<PY
import json
s=open("$O/v.json").read(); s=s[s.index('{"metric"'):]; d=json.loads(s)
print(round(d["value"],1), round(d["ms_per_step"],1), round(d["roofline"]["achieved"],1))
PY
grep "conv\|vq" $O/v.err | head -12
