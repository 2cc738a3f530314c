#!/bin/bash
# Round 2, GPU call 2: stored-gelu' epilogues, wgrad tail split, raster group height, conv / GEMM counters.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c2; mkdir -p $O
cd $R
python -m pytest tests/test_checkpoint_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/tests.log
B="python bench.py --dtype bf16 --no-cpu-baseline --steps 8 --warmup 2"
$B > $O/b_base.json 2> $O/b_base.err
COGV_WGRAD_TAIL_SPLIT=2 $B > $O/b_tail2.json 2> $O/b_tail2.err
COGV_WGRAD_TAIL_SPLIT=3 $B > $O/b_tail3.json 2> $O/b_tail3.err
COGV_GEMM_GROUP_M=2 $B > $O/b_gm2.json 2> $O/b_gm2.err
COGV_GEMM_GROUP_M=8 $B > $O/b_gm8.json 2> $O/b_gm8.err
COGV_GEMM_GROUP_M=16 $B > $O/b_gm16.json 2> $O/b_gm16.err
cd /tmp; export TMPDIR=/tmp
V="python $R/bench.py --config vqvae --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_conv1 -- $V > $O/pmc_conv1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_conv2 -- $V > $O/pmc_conv2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_conv3 -- $V > $O/pmc_conv3.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_conv4 -- $V > $O/pmc_conv4.log 2>&1
G="python $R/bench.py --dtype bf16 --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_gemm_tcc -- $G > $O/pmc_gemm_tcc.log 2>&1
cd $R
for d in pmc_conv1 pmc_conv2 pmc_conv3 pmc_conv4; do echo "== $d"; python tools/pmc_report.py $O/$d conv_kernel; done > $O/pmc_conv_report.txt 2>&1
echo "== gemm tcc" >> $O/pmc_conv_report.txt; python tools/pmc_report.py $O/pmc_gemm_tcc gemm >> $O/pmc_conv_report.txt 2>&1
find $O -name "*.db" -delete 2>/dev/null
tail -6 $O/tests.log
for f in base tail2 tail3 gm2 gm8 gm16; do python - <<PY
import json
s=open("$O/b_$f.json").read(); s=s[s.index('{"metric"'):]; d=json.loads(s)
print("$f", round(d["value"]), round(d["ms_per_step"],1), round(d["roofline"]["achieved"],1))
PY
done
grep -A3 "NN_dgrad 26112x10240x2560\|NT_fwd 26112x10240x2560\|grouped" $O/b_base.err | head -12
cat $O/pmc_conv_report.txt | cut -c1-400
