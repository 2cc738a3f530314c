"""GEMM epilogue cost at the 4B shapes (GPU box): the same NT / NN launch with different fused epilogues."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
dt = torch.bfloat16
_w = torch.randn(8192, 8192, device="cuda", dtype=dt)
for _ in range(20): _w @ _w                     # clocks and power state settle before the first row
for M, N, K in ((26112, 7680, 2560), (26112, 10240, 2560), (26112, 2560, 10240), (26112, 2560, 2560),
                (32670, 3072, 1024), (32670, 4096, 1024), (32670, 1024, 4096), (32670, 1024, 1024)):
    x = torch.randn(M, K, device="cuda", dtype=dt); w = torch.randn(N, K, device="cuda", dtype=dt) * 0.02
    bias = torch.randn(N, device="cuda", dtype=dt); aux = torch.empty(M, N, device="cuda", dtype=dt)
    slot = torch.zeros(1, device="cuda")
    fl = 2.0 * M * N * K
    row = {"M": M, "N": N, "K": K}
    for name, fn in (("none", lambda: ops.gemm(x, w)), ("bias", lambda: ops.gemm(x, w, bias=bias)),
                     ("bias_gelu", lambda: ops.gemm(x, w, bias=bias, gelu=True)),
                     ("bias_gelu_daux", lambda: ops.gemm(x, w, bias=bias, gelu=True, gelu_daux=aux)),
                     ("bias_drop_absmax", lambda: ops.gemm(x, w, bias=bias, dropout=(0.1, 1, 2), absmax=slot))):
        t = timeit(fn, iters=10, warm=2)
        row[name + "_us"] = round(t * 1e6); row[name + "_TF"] = round(fl / t / 1e12)
    print(json.dumps(row), flush=True)
