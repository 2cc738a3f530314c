#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2c6; mkdir -p $O
cd $R
python -m pytest tests/test_model_gpu.py tests/test_checkpoint_gpu.py -m gpu -x -q 2>&1 | tail -12 > $O/tests.log
tail -5 $O/tests.log
export COGV_BENCH_ONE_DEVICE=1
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1"
timeout 600 $T --master-port 29611 bench.py --gpus 2 --model-parallel 2 --config cogview-small-336M --batch 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/mp2.json 2> $O/mp2.err; echo "mp2 rc=$?"; tail -c 700 $O/mp2.json; tail -3 $O/mp2.err
timeout 600 $T --master-port 29612 bench.py --gpus 2 --shard-optimizer --config cogview-small-336M --batch 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/dp2s.json 2> $O/dp2s.err; echo "dp2-shard rc=$?"; tail -c 500 $O/dp2s.json; tail -3 $O/dp2s.err
timeout 600 $T --master-port 29613 bench.py --gpus 2 --config cogview-small-336M --batch 4 --steps 2 --warmup 1 --no-cpu-baseline > $O/dp2.json 2> $O/dp2.err; echo "dp2 rc=$?"; tail -c 400 $O/dp2.json
