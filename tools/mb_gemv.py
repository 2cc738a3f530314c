"""Bandwidth of the skinny-M (decode) GEMM path at the 4B model's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
dt = torch.bfloat16
for M in (1, 4):
    for N, K in ((7680, 2560), (2560, 2560), (10240, 2560), (2560, 10240), (58240, 2560)):
        # several independent weight copies so that the working set exceeds the 256-MB Infinity Cache
        ws = [torch.randn(N, K, device="cuda", dtype=dt) * 0.02 for _ in range(max(2, int(6e8 // (N * K * 2))))]
        x = torch.randn(M, K, device="cuda", dtype=dt)
        b = torch.randn(N, device="cuda", dtype=dt)
        def f():
            for w in ws:
                ops.gemm(x, w, bias=b)
        t = timeit(f, iters=5, warm=2) / len(ws)
        print(f"M={M} N={N:6d} K={K:6d}: {t*1e6:8.1f} us  {N*K*2/t/1e12:6.2f} TB/s", flush=True)
