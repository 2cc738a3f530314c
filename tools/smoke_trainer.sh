#!/bin/bash
# End-to-end smoke of the trainer entry point on one GPU: tiny model, synthetic CompactBinaryDataset file, checkpoint + resume.
set -e
D=$(mktemp -d)
python - <<PY
import numpy as np, sys
sys.path.insert(0, ".")
from cogview_amd.data_utils import write_compact_binary
rs = np.random.RandomState(0)
write_compact_binary("$D/train.bin", [rs.randint(8192, 58192, rs.randint(2, 30)).tolist() for _ in range(64)], rs.randint(0, 8192, (64, 1024)))
PY
COMMON="--num-layers 2 --hidden-size 256 --num-attention-heads 4 --batch-size 2 --fp16 --train-data $D/train.bin --log-interval 2 --num-workers 0 --save $D/ck --save-interval 4 --lr 1e-3 --warmup 0.1"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 -m cogview_amd.pretrain_gpt2 $COMMON --train-iters 6 2>&1 | grep -E "iteration|saved|Error|error" | tail -8
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 -m cogview_amd.pretrain_gpt2 $COMMON --train-iters 8 --load $D/ck 2>&1 | grep -E "iteration|loaded|saved|Error|error" | tail -6
ls $D/ck
