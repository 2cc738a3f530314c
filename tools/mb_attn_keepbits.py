"""Attention with dropout 0.1 at the 4B / 336M bench shapes: the regenerating backward (keep_bits off) against the stored keep
bits (forward writes them, dQ / dK.dV read them), interleaved in one process.  GPU box."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
s = 1088
drop = (0.1, 1, 2)
for H, bb in ((40, 24), (16, 30)):
    qkv = torch.randn(bb, s, 3 * H * 64, device="cuda", dtype=torch.bfloat16)
    q, k, v = [qkv[:, :, i * H * 64:(i + 1) * H * 64].view(bb, s, H, 64) for i in range(3)]
    do = torch.randn(bb, s, H, 64, device="cuda", dtype=torch.bfloat16)
    dq, dk, dv = [torch.empty(bb, s, H, 64, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
    o, lse = ops.attention_fwd(q, k, v, dropout=drop)
    o2, lse2, bits = ops.attention_fwd(q, k, v, dropout=drop, keep_bits=True)
    assert torch.equal(o, o2)
    for rep in range(3):
        row = {"H": H, "b": bb, "rep": rep}
        row["fwd_regen_us"] = round(timeit(lambda: ops.attention_fwd(q, k, v, dropout=drop), iters=10, warm=2) * 1e6, 1)
        row["fwd_store_us"] = round(timeit(lambda: ops.attention_fwd(q, k, v, dropout=drop, keep_bits=True), iters=10, warm=2) * 1e6, 1)
        row["bwd_regen_us"] = round(timeit(lambda: ops.attention_bwd(do, q, k, v, o, lse, dropout=drop, dq=dq, dk=dk, dv=dv), iters=10, warm=2) * 1e6, 1)
        row["bwd_bits_us"] = round(timeit(lambda: ops.attention_bwd(do, q, k, v, o, lse, dropout=drop, dq=dq, dk=dk, dv=dv, keep_bits=bits), iters=10, warm=2) * 1e6, 1)
        print(json.dumps(row), flush=True)
