#!/bin/bash
# Round 2, GPU call 3: conv kernel raster + prefetch depth.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c3; mkdir -p $O
cd $R
python -m pytest tests/test_vqvae_gpu.py tests/test_checkpoint_gpu.py -m gpu -x -q 2>&1 | tail -6 > $O/tests.log
V="python bench.py --config vqvae --steps 5 --warmup 2 --no-cpu-baseline"
COGV_CONV_PF=1 $V > $O/v_pf1.json 2> $O/v_pf1.err
COGV_CONV_PF=2 $V > $O/v_pf2.json 2> $O/v_pf2.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc -- python $R/bench.py --config vqvae --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_tcc.log 2>&1
cd $R
python tools/pmc_report.py $O/pmc_tcc conv_kernel > $O/pmc_report.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null
tail -4 $O/tests.log
for f in pf1 pf2; do python - <<PY
import json
s=open("$O/v_$f.json").read(); s=s[s.index('{"metric"'):]; d=json.loads(s)
print("$f", round(d["value"],1), round(d["ms_per_step"],1), round(d["roofline"]["achieved"],1))
PY
grep "conv\|vq" $O/v_$f.err | head -12
done
cat $O/pmc_report.txt | cut -c1-300
