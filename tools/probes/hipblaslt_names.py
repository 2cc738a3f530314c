"""Print which hipBLASLt kernels torch.matmul picks for the hot-path GEMM shapes (run under rocprofv3 --kernel-trace --stats):
the yardstick's winning solutions by name (macro tile MT, MFMA shape MI, wave tile, prefetch flags are part of the name).
HBL_SHAPES=4B (default) | 336M."""
import os
import torch
which = os.environ.get("HBL_SHAPES", "4B")
M, shapes = (26112, [(7680, 2560), (2560, 2560), (10240, 2560), (2560, 10240)]) if which == "4B" else \
            (32640, [(3072, 1024), (4096, 1024), (1024, 4096), (1024, 1024)])
dt = torch.float16 if os.environ.get("HBL_DTYPE", "fp16") == "fp16" else torch.bfloat16
for N, K in shapes:
    x = torch.randn(M, K, device="cuda", dtype=dt)
    w = torch.randn(N, K, device="cuda", dtype=dt)
    dy = torch.randn(M, N, device="cuda", dtype=dt)
    for _ in range(3):
        torch.matmul(x, w.t()); torch.matmul(dy, w); torch.matmul(dy.t(), x)
    torch.cuda.synchronize()
    for name, f, fl in (("fwd NT", lambda: torch.matmul(x, w.t()), 2.0 * M * N * K), ("dgrad NN", lambda: torch.matmul(dy, w), 2.0 * M * N * K),
                        ("wgrad TN", lambda: torch.matmul(dy.t(), x), 2.0 * M * N * K)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record(); torch.cuda.synchronize()
        print(f"hipBLASLt {name} M={M} N={N} K={K}: {fl * 10 / e0.elapsed_time(e1) / 1e9:7.1f} TFLOP/s", flush=True)
