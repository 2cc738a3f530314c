"""The dense attention kernels exactly as the 4B train step runs them (fp16, b = 24, 40 heads, s = 1088, dropout 0.1, stored keep
bits, fused QKV-bias column sums): forward and backward times.  COGVIEW_HIP_LIB selects the library (A/B in one call).  GPU box."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
s, drop = 1088, (0.1, 1, 2)
dt = torch.float16 if os.environ.get("MB_ATTN_DTYPE", "fp16") == "fp16" else torch.bfloat16
for H, bb in ((40, 24), (16, 30)):
    qkv = torch.randn(bb, s, 3 * H * 64, device="cuda", dtype=dt)
    q, k, v = [qkv[:, :, i * H * 64:(i + 1) * H * 64].view(bb, s, H, 64) for i in range(3)]
    do = torch.randn(bb, s, H, 64, device="cuda", dtype=dt)
    dqkv = torch.empty_like(qkv)
    outs = dict(dq=dqkv[:, :, :H * 64].view(bb, s, H, 64), dk=dqkv[:, :, H * 64:2 * H * 64].view(bb, s, H, 64), dv=dqkv[:, :, 2 * H * 64:].view(bb, s, H, 64))
    cs = torch.zeros(3 * H * 64, device="cuda", dtype=dt)
    o, lse, bits = ops.attention_fwd(q, k, v, dropout=drop, keep_bits=True)
    fl = ops.attention_executed_flops(bb, H, s, s)
    for rep in range(3):
        tf = timeit(lambda: ops.attention_fwd(q, k, v, dropout=drop, keep_bits=True), iters=10, warm=2)
        tb = timeit(lambda: ops.attention_bwd(do, q, k, v, o, lse, dropout=drop, keep_bits=bits, colsum_out=cs, **outs), iters=10, warm=2)
        print(json.dumps({"lib": os.path.basename(os.environ.get("COGVIEW_HIP_LIB") or "default"), "H": H, "b": bb, "rep": rep,
                          "fwd_us": round(tf * 1e6, 1), "bwd_us": round(tb * 1e6, 1),
                          "fwd_exec_TF": round(fl / tf / 1e12), "bwd_exec_TF": round(2.5 * fl / tb / 1e12)}), flush=True)
