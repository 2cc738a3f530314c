"""Attention forward / backward micro-benchmark at the two hot-path head counts (GPU box).  COGVIEW_HIP_LIB selects the
library, so two builds can be compared in one call (A/B on the same box: box-to-box variance is ~3 %)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
s = 1088
for H, bb in ((16, 30), (40, 24)):
    qkv = torch.randn(bb, s, 3 * H * 64, device="cuda", dtype=torch.bfloat16)
    q, k, v = [qkv[:, :, i * H * 64:(i + 1) * H * 64].view(bb, s, H, 64) for i in range(3)]
    do = torch.randn(bb, s, H, 64, device="cuda", dtype=torch.bfloat16)
    fl = 4.0 * bb * H * s * s * 64
    row = {"H": H, "b": bb}
    for p in (0.0, 0.1):
        drop = None if p == 0 else (p, 1, 2)
        t = timeit(lambda: ops.attention_fwd(q, k, v, dropout=drop), iters=10, warm=2)
        row[f"fwd_p{p}_us"] = round(t * 1e6); row[f"fwd_p{p}_TF"] = round(fl / t / 1e12)
        o, lse = ops.attention_fwd(q, k, v, dropout=drop)
        t = timeit(lambda: ops.attention_bwd(do, q, k, v, o, lse, dropout=drop), iters=10, warm=2)
        row[f"bwd_p{p}_us"] = round(t * 1e6); row[f"bwd_p{p}_TF"] = round(2.5 * fl / t / 1e12)
    print(json.dumps(row), flush=True)
