import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cogview_amd import ops
for dtype in (torch.float16, torch.bfloat16):
    for rows in (7, 500, 501, 1000, 26112):
        g = torch.Generator().manual_seed(rows)
        h = 2560
        x = torch.randn(rows, h, generator=g).to(dtype).cuda()
        gam = (torch.rand(h, generator=g) + 0.5).to(dtype).cuda()
        bet = torch.zeros(h, dtype=dtype, device="cuda")
        stream = torch.randn(rows, h, generator=g).cuda()
        dy = torch.randn(rows, h, generator=g).cuda()
        _, mean, rstd = ops.sandwich_ln_fwd(x, gam, bet, 1e-5, ops.absmax(x), residual=stream)
        res = {}
        for lean in ("0", "1", "0b", "1b"):
            os.environ["COGV_LN_BWD_LEAN"] = lean[0]
            dg, db, cs = (torch.zeros(h, dtype=dtype, device="cuda") for _ in range(3))
            res[lean] = ops.sandwich_ln_bwd(dy, x, gam, mean, rstd, dropout=(0.1, 3, 4), dgamma=dg, dbeta=db, colsum=cs).float()
        d = (res["0"] - res["1"]).abs()
        bad = d > 0
        rws = bad.any(1).nonzero().flatten()
        print(dtype, rows, "mismatches", int(bad.sum()), "rows with mismatch", int(rws.numel()), rws[:8].tolist(), "max", float(d.max()),
              "rel", float(d.max() / res["0"].abs().max()), "self-consistent regular", bool(torch.equal(res["0"], res["0b"])), "lean", bool(torch.equal(res["1"], res["1b"])),
              "zeros equal", bool(torch.equal(res["0"] == 0, res["1"] == 0)))
        if bad.any():
            i, j = bad.nonzero()[0].tolist()
            print("   first", i, j, float(res["0"][i, j]), float(res["1"][i, j]))
