import os, sys
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/cogview_amd") else os.getcwd())
import torch
from cogview_amd import ops
rows, h, dt = 1000, 2560, torch.float16
g = torch.Generator().manual_seed(3)
rnd = lambda shape, s=1.0: (torch.randn(*shape, generator=g) * s)
a, w = rnd((rows, 64)).to(dt).cuda(), rnd((h, 64), 0.2).to(dt).cuda()
drop = (0.1, 31, 4)
slot, slot_y = ops.new_absmax_slot(a.device), ops.new_absmax_slot(a.device)
ao = ops.gemm(a, w, dropout=drop, absmax=slot)
gam3 = (torch.rand(h, generator=g) + 0.5).to(dt).cuda(); gam2 = (torch.rand(h, generator=g) + 0.5).to(dt).cuda()
bet = torch.zeros(h, dtype=dt, device="cuda")
x = rnd((rows, h)).cuda()
y, m3, r3 = ops.sandwich_ln_fwd(ao, gam3, bet, 1e-5, slot, residual=x, absmax_out=slot_y)
_, m2, r2 = ops.sandwich_ln_fwd(y, gam2, bet, 1e-5, slot_y)
dc, dout = rnd((rows, h)).to(dt).cuda(), rnd((rows, h)).cuda()
P = [torch.zeros(h, dtype=dt, device="cuda") for _ in range(10)]
dy_ref = ops.sandwich_ln_bwd(dc, y, gam2, m2, r2, add_in=dout, dgamma=P[0], dbeta=P[1])
dao_ref = ops.sandwich_ln_bwd(dy_ref, ao, gam3, m3, r3, dropout=drop, dgamma=P[2], dbeta=P[3], colsum=P[4], marked=True)
dy, dao = ops.sandwich_ln_bwd_pair(dc, y, gam2, m2, r2, dout, ao, gam3, m3, r3, dropout_p=0.1, dgamma2=P[5], dbeta2=P[6], dgamma3=P[7], dbeta3=P[8], colsum=P[9])
torch.cuda.synchronize()
ne = (dy != dy_ref)
print("dy mismatches", int(ne.sum()), "of", dy.numel(), "max abs diff", float((dy - dy_ref).abs().max()), "max rel", float(((dy - dy_ref).abs() / dy_ref.abs().clamp_min(1e-6)).max()))
if ne.any():
    idx = ne.nonzero()[:5]
    for i, j in idx.tolist(): print(i, j, float(dy[i, j]), float(dy_ref[i, j]))
    print("rows with mismatch:", ne.any(1).sum().item(), "cols with mismatch:", ne.any(0).sum().item())
ne2 = (dao.view(torch.int16) != dao_ref.view(torch.int16))
print("d_ao mismatches", int(ne2.sum()), "max abs diff", float((dao.float() - dao_ref.float()).abs().max()))
for k in range(5):
    print("param", k, float((P[k].float() - P[5 + k].float()).abs().max()), float(P[k].float().abs().max()))
rows_bad = ne.any(1).nonzero().flatten().tolist()
print("bad rows (first 40):", rows_bad[:40])
print("bad rows mod 2:", sum(r % 2 for r in rows_bad), "of", len(rows_bad), " min", min(rows_bad) if rows_bad else None, "max", max(rows_bad) if rows_bad else None)
print("per bad row mismatching columns (first 8 rows):", [int(ne[r].sum()) for r in rows_bad[:8]])
