"""Does the attention kernels' time depend on where their 128-byte rows lie?  The same kernels, the same work (960 (batch, head)
units of 1088 x 1088 scores, dropout 0.1, stored keep bits): (a) as the train step runs them -- q / k / v read in place from the fused
[b, s, 3h] QKV activation (row stride 15 KiB, dO 5 KiB) -- and (b) on head-major tensors (B = 960, H = 1: rows of a unit contiguous)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
s, drop = 1088, (0.1, 1, 2)
dt = torch.float16


def case(name, bb, H, fused):
    if fused:
        qkv = torch.randn(bb, s, 3 * H * 64, device="cuda", dtype=dt)
        q, k, v = [qkv[:, :, i * H * 64:(i + 1) * H * 64].view(bb, s, H, 64) for i in range(3)]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = [dqkv[:, :, i * H * 64:(i + 1) * H * 64].view(bb, s, H, 64) for i in range(3)]
    else:
        q, k, v = [torch.randn(bb, s, H, 64, device="cuda", dtype=dt) for _ in range(3)]
        dq, dk, dv = [torch.empty_like(q) for _ in range(3)]
    do = torch.randn(bb, s, H, 64, device="cuda", dtype=dt)
    o, lse, bits = ops.attention_fwd(q, k, v, dropout=drop, keep_bits=True)
    for rep in range(2):
        tf = timeit(lambda: ops.attention_fwd(q, k, v, dropout=drop, keep_bits=True), iters=10, warm=2)
        tb = timeit(lambda: ops.attention_bwd(do, q, k, v, o, lse, dropout=drop, keep_bits=bits, dq=dq, dk=dk, dv=dv), iters=10, warm=2)
        print(f"{name:54s} fwd {tf*1e6:7.1f} us   bwd {tb*1e6:7.1f} us", flush=True)


case("(a) b=24 H=40, rows inside the fused QKV activation", 24, 40, True)
case("(b) B=960 H=1, head-major (a unit's rows contiguous)", 960, 1, False)
case("(c) b=24 H=40, separate q / k / v tensors [b,s,h]", 24, 40, False)
case("(a) again", 24, 40, True)
