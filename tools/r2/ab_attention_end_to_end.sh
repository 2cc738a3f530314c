#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c11; mkdir -p $O
cd $R
for lib in old new; do
  if [ $lib = old ]; then export COGVIEW_HIP_LIB=$R/build/ab/libcogview_old_attn.so; else unset COGVIEW_HIP_LIB; fi
  python bench.py --dtype bf16 --no-cpu-baseline --steps 8 --warmup 2 > $O/b4_$lib.json 2> $O/b4_$lib.err
  python bench.py --config cogview-small-336M --dtype bf16 --no-cpu-baseline > $O/b336_$lib.json 2> $O/b336_$lib.err
  for f in b4 b336; do python - <<PY
import json
s=open("$O/${f}_$lib.json").read(); s=s[s.index('{"metric"'):]; d=json.loads(s)
print("$f $lib", round(d["value"]), round(d["ms_per_step"],1), round(d["mfma_roofline_frac_end_to_end"],4), round(d["roofline"]["achieved"],1), round(d["roofline"]["share_of_step_time"],3))
PY
  done
done
