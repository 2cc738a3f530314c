#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c8; mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $O/tests.log; tail -4 $O/tests.log
python tools/mb_attn.py 2>/dev/null | tail -2 | tee $O/mb_attn.log
python bench.py --dtype bf16 --no-cpu-baseline > $O/b4.json 2> $O/b4.err
python bench.py --config cogview-small-336M --dtype bf16 --no-cpu-baseline > $O/b336.json 2> $O/b336.err
for f in b4 b336; do python - <<PY
import json
s=open("$O/$f.json").read(); s=s[s.index('{"metric"'):]; d=json.loads(s)
print("$f", round(d["value"]), round(d["ms_per_step"],1), round(d["mfma_roofline_frac_end_to_end"],4), round(d["roofline"]["achieved"],1), round(d["roofline"]["share_of_step_time"],3))
PY
done
