"""Sandwich-LN forward / backward micro-benchmark (GPU box): GB/s of algorithmic traffic, both hot-path widths."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from tools.microbench import timeit

for h, b in ((1024, 30), (2560, 24)):
    M = b * 1088
    dt = torch.bfloat16
    x = torch.randn(M, h, device="cuda", dtype=dt)
    dy = torch.randn(M, h, device="cuda", dtype=dt)
    add = torch.randn(M, h, device="cuda", dtype=dt)
    g_, b_ = torch.ones(h, device="cuda", dtype=dt), torch.zeros(h, device="cuda", dtype=dt)
    am = ops.absmax(x)
    y, mean, rstd = ops.sandwich_ln_fwd(x, g_, b_, 1e-5, am)
    dg, db, cs = torch.zeros_like(g_), torch.zeros_like(b_), torch.zeros_like(b_)
    row = {"h": h, "rows": M}
    t = timeit(lambda: ops.sandwich_ln_fwd(x, g_, b_, 1e-5, am))
    row["fwd_us"], row["fwd_GBs"] = round(t * 1e6, 1), round(2 * M * h * 2 / t / 1e9)
    t = timeit(lambda: ops.sandwich_ln_fwd(x, g_, b_, 1e-5, am, residual=add))
    row["fwd_res_us"], row["fwd_res_GBs"] = round(t * 1e6, 1), round(3 * M * h * 2 / t / 1e9)
    t = timeit(lambda: ops.sandwich_ln_bwd(dy, x, g_, mean, rstd, dgamma=dg, dbeta=db, accumulate=True))
    row["bwd_us"], row["bwd_GBs"] = round(t * 1e6, 1), round(3 * M * h * 2 / t / 1e9)
    t = timeit(lambda: ops.sandwich_ln_bwd(dy, x, g_, mean, rstd, add_in=add, dgamma=dg, dbeta=db, accumulate=True))
    row["bwd_add_us"], row["bwd_add_GBs"] = round(t * 1e6, 1), round(4 * M * h * 2 / t / 1e9)
    t = timeit(lambda: ops.sandwich_ln_bwd(dy, x, g_, mean, rstd, dropout=(0.1, 1, 2), dgamma=dg, dbeta=db, colsum=cs, accumulate=True))
    row["bwd_drop_colsum_us"], row["bwd_drop_colsum_GBs"] = round(t * 1e6, 1), round(3 * M * h * 2 / t / 1e9)
    print(json.dumps(row), flush=True)
