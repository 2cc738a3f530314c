import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
M, N, K = 32640, 4096, 1024
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
for v in (3, 5):
    for _ in range(3):
        ops.gemm(x, w, variant=v)
for _ in range(3):
    ops.gemm(dy, x, trans_a=True, trans_b=True, variant=3)
torch.cuda.synchronize()
