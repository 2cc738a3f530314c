"""dgrad launch with the stored-gelu' multiply and the bias-gradient column sums (the slowest GEMM of the step), per library."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
M, h = 26112, 2560
dt = torch.bfloat16
dy = torch.randn(M, h, device="cuda", dtype=dt); w2 = torch.randn(h, 4 * h, device="cuda", dtype=dt) * 0.02
aux = torch.randn(M, 4 * h, device="cuda", dtype=dt); cs = torch.zeros(4 * h, device="cuda", dtype=dt)
x = torch.randn(M, h, device="cuda", dtype=dt); w1 = torch.randn(4 * h, h, device="cuda", dtype=dt) * 0.02; b1 = torch.randn(4 * h, device="cuda", dtype=dt)
for _ in range(3):
    t = min(timeit(lambda: ops.gemm(dy, w2, trans_b=True, mul_aux=aux, colsum_out=cs, colsum_accumulate=False), iters=10, warm=3) for _ in range(2))
    t2 = min(timeit(lambda: ops.gemm(x, w1, bias=b1, gelu=True, gelu_daux=aux), iters=10, warm=3) for _ in range(2))
    print(f"{os.environ.get('COGVIEW_HIP_LIB', 'new')[-16:]:16s} dgrad mulaux+colsum {t*1e6:7.1f} us {2.0*M*4*h*h/t/1e12:7.1f} TF | fwd gelu+daux {t2*1e6:7.1f} us {2.0*M*4*h*h/t2/1e12:7.1f} TF", flush=True)
