#!/bin/bash
# rocprofv3 counters of the dominant kernel (round-3 verdict item 5): MFMA-busy cycles, busy cycles, shader clock
# (GRBM_GUI_ACTIVE / kernel wall time), MFMA instruction count -- gemm_w4_kernel on the 4B step's GEMM launches
# (tools/r4/mb_gemm_ab.py: real shapes and epilogues).  Counters only with --kernel-trace (no other trace domain).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/r4/pmc_gemm_a $R/gpurun_out/r4/pmc_gemm_b
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
  -d $R/gpurun_out/r4/pmc_gemm_a -- python $R/tools/r4/mb_gemm_ab.py pmc > $R/gpurun_out/r4/pmc_gemm_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv \
  -d $R/gpurun_out/r4/pmc_gemm_b -- python $R/tools/r4/mb_gemm_ab.py pmc > $R/gpurun_out/r4/pmc_gemm_b.log 2>&1
cd $R
python tools/r4/pmc_gemm_report.py gpurun_out/r4/pmc_gemm_a gpurun_out/r4/pmc_gemm_b > gpurun_out/r4/pmc_gemm_report.txt 2>&1
cat gpurun_out/r4/pmc_gemm_report.txt
