"""GEMM micro-benchmark + correctness screen for the three layouts (GPU box)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from tools.microbench import timeit

def main():
    dt = torch.bfloat16
    b = int(os.environ.get("MB_BATCH", "16"))
    M = b * 1088
    shapes = [(3072, 1024, "S.qkv"), (1024, 1024, "S.dense"), (4096, 1024, "S.h4h"), (1024, 4096, "S.4hh"),
              (7680, 2560, "B.qkv"), (2560, 2560, "B.dense"), (10240, 2560, "B.h4h"), (2560, 10240, "B.4hh")]
    if os.environ.get("MB_SMALL"):
        shapes = shapes[:4]
    variants = [int(v) for v in os.environ.get("MB_VARIANTS", "1,0").split(",")]
    for N, K, name in shapes:
        x = torch.randn(M, K, device="cuda", dtype=dt)
        w = torch.randn(N, K, device="cuda", dtype=dt) * 0.05
        dy = torch.randn(M, N, device="cuda", dtype=dt)
        fl = 2.0 * M * N * K
        refs = {"fwd": torch.matmul(x, w.t()).float(), "dgrad": torch.matmul(dy, w).float(), "wgrad": torch.matmul(dy.t(), x).float()}
        calls = {"fwd": lambda v: ops.gemm(x, w, variant=v), "dgrad": lambda v: ops.gemm(dy, w, trans_b=True, variant=v),
                 "wgrad": lambda v: ops.gemm(dy, x, trans_a=True, trans_b=True, variant=v)}
        row = {"shape": name}
        for kind in ("fwd", "dgrad", "wgrad"):
            for v in variants:
                y = calls[kind](v)
                err = ((y.float() - refs[kind]).norm() / refs[kind].norm()).item()
                t = timeit(lambda: calls[kind](v), iters=10, warm=2)
                row[f"{kind}.v{v}"] = round(fl / t / 1e12)
                if err > 4e-3:
                    row[f"{kind}.v{v}.ERR"] = float(f"{err:.1e}")
        t = timeit(lambda: torch.matmul(x, w.t()), iters=10, warm=2)
        row["hipblaslt_fwd"] = round(fl / t / 1e12)
        t = timeit(lambda: torch.matmul(dy.t(), x), iters=10, warm=2)
        row["hipblaslt_wgrad"] = round(fl / t / 1e12)
        print(json.dumps(row), flush=True)

if __name__ == "__main__":
    main()
