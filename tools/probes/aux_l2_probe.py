"""Probe: how much of the stored-tensor GEMM epilogues' cost is the memory side of the aux operand?  The dGeLU-side dgrad
(out = (dY . W) * aux + column sums) and the GeLU forward (stores gelu' into aux) are timed with the real [M, N] aux tensor and
with an aux whose row stride is 0 (every row reads / writes the same N elements: L2 resident, same instruction stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
g = torch.Generator(device="cuda").manual_seed(1)
M, h = 26112, 2560
dt = torch.float16
rn = lambda *s: torch.randn(*s, device="cuda", dtype=dt, generator=g)
x = rn(M, h)
w_1, w_2 = rn(4 * h, h) * 0.02, rn(h, 4 * h) * 0.02
b1 = rn(4 * h) * 0.02
aux = rn(M, 4 * h)
aux0 = aux[:1].expand(M, 4 * h)
cs = torch.zeros(4 * h, device="cuda", dtype=dt)
out = torch.empty(M, 4 * h, device="cuda", dtype=dt)
out0 = out[:1].expand(M, 4 * h)
fl = 2.0 * M * 4 * h * h
cases = [
    ("dgrad mulaux+colsum, aux [M,N]", lambda: ops.gemm(x, w_2, trans_b=True, mul_aux=aux, colsum_out=cs, out=out)),
    ("dgrad mulaux+colsum, aux row stride 0", lambda: ops.gemm(x, w_2, trans_b=True, mul_aux=aux0, colsum_out=cs, out=out)),
    ("dgrad plain (no aux, no colsum)", lambda: ops.gemm(x, w_2, trans_b=True, out=out)),
    ("dgrad colsum only", lambda: ops.gemm(x, w_2, trans_b=True, colsum_out=cs, out=out)),
    ("fwd bias+gelu+daux, daux [M,N]", lambda: ops.gemm(x, w_1, bias=b1, gelu=True, gelu_daux=aux, out=out)),
    ("fwd bias+gelu+daux, daux row stride 0", lambda: ops.gemm(x, w_1, bias=b1, gelu=True, gelu_daux=aux0, out=out)),
    ("fwd bias+gelu (nothing stored beside C)", lambda: ops.gemm(x, w_1, bias=b1, gelu=True, out=out)),
    ("fwd bias only", lambda: ops.gemm(x, w_1, bias=b1, out=out)),
]
for rep in range(2):
    for name, f in cases:
        t = min(timeit(f, iters=8, warm=2) for _ in range(2))
        print(f"{name:44s} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF", flush=True)
