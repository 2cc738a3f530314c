"""Print which hipBLASLt kernels torch.matmul picks for the hot-path GEMM shapes (run under rocprofv3 --kernel-trace)."""
import torch
M = 32640
for N, K in [(3072, 1024), (4096, 1024), (1024, 4096), (1024, 1024)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    dy = torch.randn(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(x, w.t()); torch.matmul(dy, w); torch.matmul(dy.t(), x)
torch.cuda.synchronize()
