import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
M, N, K = 26112, 2560, 2560
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * 0.05
dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3):
    ops.gemm(x, w, variant=9)                                   # NT
    ops.gemm(dy, w, trans_b=True, variant=9)                    # NN (B contraction-strided)
    ops.gemm(dy, x, trans_a=True, trans_b=True, variant=9, splitk=2)      # TT
torch.cuda.synchronize()
