import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from oracle import cogview_oracle as O
def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
for (b, H, s_q, s_k, sep) in [(1, 1, 64, 64, 0), (2, 2, 40, 40, 0), (1, 1, 128, 128, 0), (1, 3, 200, 200, 0)]:
    g = torch.Generator().manual_seed(1)
    q, k, v, do = [torch.randn(b, s, H, 64, generator=g).half() for s in (s_q, s_k, s_k, s_q)]
    qr, kr, vr = [t.float().permute(0, 2, 1, 3).contiguous().requires_grad_(True) for t in (q, k, v)]
    o_ref = O.standard_attention(qr, kr, vr, O.build_mask(s_q, s_k, sep))
    o_ref.backward(do.float().permute(0, 2, 1, 3))
    o, lse = ops.attention_fwd(q.cuda(), k.cuda(), v.cuda(), sep=sep)
    dq, dk, dv = ops.attention_bwd(do.cuda(), q.cuda(), k.cuda(), v.cuda(), o, lse, sep=sep)
    print((b, H, s_q, s_k, sep), "o", rel(o, o_ref.permute(0, 2, 1, 3)), "dq", rel(dq, qr.grad.permute(0, 2, 1, 3)),
          "dk", rel(dk, kr.grad.permute(0, 2, 1, 3)), "dv", rel(dv, vr.grad.permute(0, 2, 1, 3)))
    if b == 1 and H == 1 and s_q == 64:
        e = (o.float().cpu() - o_ref.permute(0, 2, 1, 3)).abs()[0, :, 0, :]
        print("fwd err by query rows(0..63) max:", [round(x, 3) for x in e.max(1)[0].tolist()][:16], "...")
        print("fwd err by d cols max:", [round(x, 3) for x in e.max(0)[0].tolist()])
