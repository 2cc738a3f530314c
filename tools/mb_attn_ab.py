"""A/B inside one process: attention backward with and without the fused QKV-bias-gradient column sums (4B shape)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
b, H, s = 24, 40, 1088
dt = torch.bfloat16
qkv = torch.randn(b, s, 3 * H * 64, device="cuda", dtype=dt)
q, k, v = [qkv[:, :, i * H * 64:(i + 1) * H * 64].view(b, s, H, 64) for i in range(3)]
do = torch.randn(b, s, H, 64, device="cuda", dtype=dt)
drop = (0.1, 1, 2)
o, lse = ops.attention_fwd(q, k, v, dropout=drop)
dqkv = torch.empty_like(qkv)
outs = dict(dq=dqkv[:, :, :H * 64].view(b, s, H, 64), dk=dqkv[:, :, H * 64:2 * H * 64].view(b, s, H, 64), dv=dqkv[:, :, 2 * H * 64:].view(b, s, H, 64))
cs = torch.zeros(3 * H * 64, device="cuda", dtype=dt)
for rep in range(3):
    t0 = timeit(lambda: ops.attention_bwd(do, q, k, v, o, lse, dropout=drop, **outs), iters=10)
    t1 = timeit(lambda: ops.attention_bwd(do, q, k, v, o, lse, dropout=drop, colsum_out=cs, **outs), iters=10)
    t2 = timeit(lambda: ops.colsum(dqkv.view(-1, 3 * H * 64), out=cs, accumulate=True), iters=10)
    tf = timeit(lambda: ops.attention_fwd(q, k, v, dropout=drop), iters=10)
    print(f"bwd plain {t0*1e6:.1f} us | bwd + fused colsum {t1*1e6:.1f} us | separate colsum {t2*1e6:.1f} us | fwd {tf*1e6:.1f} us", flush=True)
