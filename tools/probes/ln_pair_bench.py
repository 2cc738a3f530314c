"""LN2' + LN3' of a 4B layer: the two launches against cogv_sandwich_ln_bwd_pair (h = 2560, micro-batch 30 rows).  GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
rows, h, dt = 32640, 2560, torch.float16
g = torch.Generator(device="cuda").manual_seed(1)
rn = lambda *s: torch.randn(*s, device="cuda", generator=g)
a, w = rn(rows, 64).to(dt), (rn(h, 64) * 0.2).to(dt)
drop = (0.1, 31, 4)
slot, slot_y = ops.new_absmax_slot(a.device), ops.new_absmax_slot(a.device)
ao = ops.gemm(a, w, dropout=drop, absmax=slot)
gam = (torch.rand(h, device="cuda", generator=g) + 0.5).to(dt)
bet = torch.zeros(h, dtype=dt, device="cuda")
x = rn(rows, h)
y, m3, r3 = ops.sandwich_ln_fwd(ao, gam, bet, 1e-5, slot, residual=x, absmax_out=slot_y)
_, m2, r2 = ops.sandwich_ln_fwd(y, gam, bet, 1e-5, slot_y)
dc, dout = rn(rows, h).to(dt), rn(rows, h)
P = [torch.zeros(h, dtype=dt, device="cuda") for _ in range(5)]


def two():
    dy = ops.sandwich_ln_bwd(dc, y, gam, m2, r2, add_in=dout, dgamma=P[0], dbeta=P[1])
    return ops.sandwich_ln_bwd(dy, ao, gam, m3, r3, dropout=drop, dgamma=P[2], dbeta=P[3], colsum=P[4], marked=True)


def pair():
    return ops.sandwich_ln_bwd_pair(dc, y, gam, m2, r2, dout, ao, gam, m3, r3, dropout_p=0.1, dgamma2=P[0], dbeta2=P[1],
                                    dgamma3=P[2], dbeta3=P[3], colsum=P[4])


for blocks in os.environ.get("PAIR_BLOCKS", "256").split(","):
    os.environ["COGV_LN_BWD_PAIR_BLOCKS"] = blocks
for rep in range(3):
    t2 = timeit(two, iters=20, warm=3)
    tp = timeit(pair, iters=20, warm=3)
    print(f"two launches {t2*1e6:7.1f} us ({rows*h*22/t2/1e12:.2f} TB/s of 22 B/element)   pair {tp*1e6:7.1f} us ({rows*h*18/tp/1e12:.2f} TB/s of 18 B/element)", flush=True)
