#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c7; mkdir -p $O
cd $R
for c in 512 768 1024 2048; do echo "LN blocks cap $c"; COGV_LN_BWD_BLOCKS=$c python tools/mb_ln.py 2>/dev/null | tail -2; done | tee $O/mb_ln.log
python tools/mb_decode.py 2>&1 | tail -6 | tee $O/mb_decode.log
python bench.py --config cogview-small-336M --dtype bf16 --no-cpu-baseline > $O/b336.json 2> $O/b336.err; tail -c 300 $O/b336.json | head -c 10; python - <<PY
import json
s=open("$O/b336.json").read(); s=s[s.index('{"metric"'):]; d=json.loads(s)
print("336M", round(d["value"]), round(d["ms_per_step"],1), round(d["mfma_roofline_frac_end_to_end"],3), round(d["roofline"]["achieved"],1))
PY
grep -A14 "GEMM launches" $O/b336.err
