#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c5; mkdir -p $O
cd $R
for e in 0 1 2 3 4 6 7; do
COGV_CONV_EXP=$e python bench.py --config vqvae --steps 3 --warmup 1 --no-cpu-baseline > $O/v_$e.json 2> $O/v_$e.err
echo "EXP=$e $(grep 'conv kind2 256x128x128' $O/v_$e.err)"
done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- python $R/bench.py --config vqvae --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc.log 2>&1
cd $R; python tools/pmc_report.py $O/pmc conv_kernel | cut -c1-300
grep "conv_kernel" $O/pmc/*/*kernel_trace.csv | head -3 | cut -c1-300
find $O -name "*.db" -delete
