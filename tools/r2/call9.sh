#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for pre in tests/test_model_gpu.py::test_two_rank_sharded_optimizer_step_on_one_gpu tests/test_kernels_gpu.py tests/test_gemm_bench_scale_gpu.py tests/test_checkpoint_gpu.py "tests/test_model_gpu.py -k not_sharded"; do
  echo "== $pre"; python -m pytest $pre tests/test_vqvae_gpu.py::test_small_vs_reference_golden -m gpu -q 2>&1 | tail -3
done
