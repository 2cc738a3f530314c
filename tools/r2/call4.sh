#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c4; mkdir -p $O
cd $R
python -m pytest tests/test_vqvae_gpu.py -m gpu -x -q 2>&1 | tail -6 > $O/tests.log
python bench.py --config vqvae --steps 5 --warmup 2 --no-cpu-baseline > $O/v.json 2> $O/v.err
tail -4 $O/tests.log
python - <<PY
import json
s=open("$O/v.json").read(); s=s[s.index('{"metric"'):]; d=json.loads(s)
print(round(d["value"],1), round(d["ms_per_step"],1), round(d["roofline"]["achieved"],1))
PY
grep "conv\|vq" $O/v.err | head -12
