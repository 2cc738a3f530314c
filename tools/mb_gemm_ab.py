"""The 4B / 336M step's GEMM launches with their real epilogues, one library per process (COGVIEW_HIP_LIB); the driver script
alternates libraries.  GPU box.   python tools/mb_gemm_ab.py <tag>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
tag = sys.argv[1]
g = torch.Generator(device="cuda").manual_seed(1)
dt = torch.bfloat16
rn = lambda *s: torch.randn(*s, device="cuda", dtype=dt, generator=g)
M, h = 26112, 2560
x, x4 = rn(M, h), rn(M, 4 * h)
w_qkv, w_d, w_1, w_2 = rn(3 * h, h) * 0.02, rn(h, h) * 0.02, rn(4 * h, h) * 0.02, rn(h, 4 * h) * 0.02
b1, bh, b3 = rn(4 * h) * 0.02, rn(h) * 0.02, rn(3 * h) * 0.02
aux = torch.empty(M, 4 * h, device="cuda", dtype=dt)
cs = torch.zeros(4 * h, device="cuda", dtype=dt)
amax = torch.zeros(1, device="cuda", dtype=torch.float32)
gw = [torch.zeros(n, k, device="cuda", dtype=dt) for n, k in ((3 * h, h), (h, h), (4 * h, h), (h, 4 * h))]
dy3, dy4 = rn(M, 3 * h), rn(M, 4 * h)
cases = [
    ("fwd  qkv    bias", 2.0 * M * 3 * h * h, lambda: ops.gemm(x, w_qkv, bias=b3)),
    ("fwd  h->h   bias+drop+amax", 2.0 * M * h * h, lambda: ops.gemm(x, w_d, bias=bh, dropout=(0.1, 1, 2), absmax=amax)),
    ("fwd  h->4h  bias+gelu+daux", 2.0 * M * 4 * h * h, lambda: ops.gemm(x, w_1, bias=b1, gelu=True, gelu_daux=aux)),
    ("fwd  4h->h  bias+drop+amax", 2.0 * M * 4 * h * h, lambda: ops.gemm(x4, w_2, bias=bh, dropout=(0.1, 1, 2), absmax=amax)),
    ("dgrad 4h<-h mulaux+colsum", 2.0 * M * 4 * h * h, lambda: ops.gemm(x, w_2, trans_b=True, mul_aux=aux, colsum_out=cs)),
    ("dgrad h<-4h plain", 2.0 * M * 4 * h * h, lambda: ops.gemm(x4, w_1, trans_b=True)),
    ("dgrad h<-h  plain", 2.0 * M * h * h, lambda: ops.gemm(x, w_d, trans_b=True)),
    ("dgrad h<-3h plain", 2.0 * M * 3 * h * h, lambda: ops.gemm(dy3, w_qkv, trans_b=True)),
    ("wgrad grouped x4 accumulate", 2.0 * M * 12 * h * h, lambda: ops.gemm_grouped(
        [(dy3, x, gw[0]), (x, x, gw[1]), (dy4, x, gw[2]), (x, x4, gw[3])], accumulate=True)),
]
M2, h2 = 32640, 1024
y2, w2q, w21, b2q, b21 = rn(M2, h2), rn(3 * h2, h2) * 0.02, rn(4 * h2, h2) * 0.02, rn(3 * h2) * 0.02, rn(4 * h2) * 0.02
aux2 = torch.empty(M2, 4 * h2, device="cuda", dtype=dt)
cases += [
    ("336M fwd qkv  bias", 2.0 * M2 * 3 * h2 * h2, lambda: ops.gemm(y2, w2q, bias=b2q)),
    ("336M fwd h->4h bias+gelu+daux", 2.0 * M2 * 4 * h2 * h2, lambda: ops.gemm(y2, w21, bias=b21, gelu=True, gelu_daux=aux2)),
    ("336M dgrad h<-h plain", 2.0 * M2 * h2 * h2, lambda: ops.gemm(y2, w2q[:h2].contiguous(), trans_b=True)),
]
for nm, y, ref in (("NT", ops.gemm(x, w_d), x.float() @ w_d.float().t()), ("NN", ops.gemm(x, w_d, trans_b=True), x.float() @ w_d.float())):
    print(f"[{tag:8s}] check {nm}: rel-L2 {((y.float() - ref).norm() / ref.norm()).item():.2e}", flush=True)
for name, fl, f in cases:
    t = min(timeit(f, iters=8, warm=2) for _ in range(2))
    print(f"[{tag:8s}] {name:32s} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF", flush=True)
