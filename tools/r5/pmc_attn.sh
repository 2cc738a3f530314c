#!/bin/bash
# rocprofv3 counters of the three dense attention kernels at the bench shape (b = 24, 40 heads, s = 1088, dropout 0.1, stored
# keep bits: tools/pmc_attn.py); two --pmc passes, each with --kernel-trace only.  -> profiles/r05_attention_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r5; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/pmc_attn_a $OUT/pmc_attn_b
PMC_ATTN_DTYPE=${PMC_ATTN_DTYPE:-fp16} timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv \
  -d $OUT/pmc_attn_a -- python $R/tools/pmc_attn.py > $OUT/pmc_attn_a.log 2>&1
PMC_ATTN_DTYPE=${PMC_ATTN_DTYPE:-fp16} timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv \
  -d $OUT/pmc_attn_b -- python $R/tools/pmc_attn.py > $OUT/pmc_attn_b.log 2>&1
cd $R
# executed score elements: 24 x 40 (batch x heads) x 153 visited 64 x 64 blocks x 4096
python tools/r5/pmc_kernels_report.py --match attn --scores 6.016e8 $OUT/pmc_attn_a $OUT/pmc_attn_b > $OUT/attention_pmc.txt 2>&1
cat $OUT/attention_pmc.txt
