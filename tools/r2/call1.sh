#!/bin/bash
# Round 2, GPU call 1: new parity tests + whole GPU suite, default bench (bf16 + fp16 leg), VQ-VAE bench + rocprof, attention PMC.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c1; mkdir -p $O
cd $R
python -m pytest tests/test_checkpoint_gpu.py tests/test_gemm_bench_scale_gpu.py -m gpu -x -q 2>&1 | tail -15 > $O/new_tests.log
python -m pytest tests -m gpu -q 2>&1 | tail -15 > $O/gpu_tests.log
python bench.py > $O/bench_4B.json 2> $O/bench_4B.err
python bench.py --config vqvae > $O/bench_vqvae.json 2> $O/bench_vqvae.err
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_vqvae -- python $R/bench.py --config vqvae --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/prof_vqvae.log 2>&1)
cp $(ls $O/prof_vqvae/*/*kernel_stats.csv | head -1) $O/vqvae_kernel_stats.csv
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_attn -- python $R/tools/pmc_attn.py > $O/pmc_attn.log 2>&1)
python tools/pmc_report.py $O/pmc_attn attn > $O/pmc_attn_report.txt 2>&1
rm -rf $O/prof_vqvae/*/*.db 2>/dev/null
tail -5 $O/new_tests.log; tail -4 $O/gpu_tests.log; cat $O/bench_4B.json | head -c 1500; echo; tail -25 $O/bench_vqvae.err; cat $O/bench_vqvae.json | head -c 1200; echo; head -12 $O/vqvae_kernel_stats.csv | cut -c1-200; cat $O/pmc_attn_report.txt | head -30
