"""Sandwich-LN in the layer's four roles with the fp32 residual stream (round 3) next to the all-16-bit forms:
microseconds and GB/s of algorithmic traffic at both hot-path widths.  COGV_LN_BWD_ROWS=2|4 picks the rows in flight of
the wide STREAM_IN backward."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from tools.microbench import timeit

for h, b in ((1024, 30), (2560, 24)):
    M = b * 1088
    dt = torch.bfloat16
    x16 = torch.randn(M, h, device="cuda", dtype=dt)
    x32 = torch.randn(M, h, device="cuda")
    r32 = torch.randn(M, h, device="cuda")
    dy16 = torch.randn(M, h, device="cuda", dtype=dt)
    dy32 = torch.randn(M, h, device="cuda")
    g_, b_ = torch.ones(h, device="cuda", dtype=dt), torch.zeros(h, device="cuda", dtype=dt)
    am16, am32 = ops.absmax(x16), ops.absmax(x32)
    _, mean, rstd = ops.sandwich_ln_fwd(x32, g_, b_, 1e-5, am32)
    dg, db, cs = torch.zeros_like(g_), torch.zeros_like(b_), torch.zeros_like(b_)
    row = {"h": h, "rows": M}
    B = M * h
    def rec(name, fn, nbytes):
        t = timeit(fn)
        row[name + "_us"], row[name + "_GBs"] = round(t * 1e6, 1), round(nbytes / t / 1e9)
    slot = ops.new_absmax_slot(x16.device)
    rec("fwd_LN1_stream_in", lambda: ops.sandwich_ln_fwd(x32, g_, b_, 1e-5, am32), B * 6)
    rec("fwd_LN3_stream_out", lambda: ops.sandwich_ln_fwd(x16, g_, b_, 1e-5, am16, residual=r32, absmax_out=slot), B * 10)
    rec("fwd_all16", lambda: ops.sandwich_ln_fwd(x16, g_, b_, 1e-5, am16), B * 4)
    rec("fwd_all16_res", lambda: ops.sandwich_ln_fwd(x16, g_, b_, 1e-5, am16, residual=dy16, absmax_out=slot), B * 6)
    rec("bwd_LN1_stream_in_add", lambda: ops.sandwich_ln_bwd(dy16, x32, g_, mean, rstd, add_in=r32, dgamma=dg, dbeta=db, accumulate=True), B * 14)
    rec("bwd_LN4_stream_out_drop_colsum", lambda: ops.sandwich_ln_bwd(dy32, x16, g_, mean, rstd, dropout=(0.1, 1, 2), dgamma=dg, dbeta=db, colsum=cs, accumulate=True), B * 8)
    x16m = torch.where(torch.rand(M, h, device="cuda") < 0.1, torch.full_like(x16, -0.0), x16)      # marked zeros: 10 % dropped
    rec("bwd_LN4_stream_out_marked_colsum", lambda: ops.sandwich_ln_bwd(dy32, x16m, g_, mean, rstd, dropout=(0.1, 1, 2), dgamma=dg, dbeta=db, colsum=cs, accumulate=True, marked=True), B * 8)
    rec("bwd_LN4_stream_out_nodrop_colsum", lambda: ops.sandwich_ln_bwd(dy32, x16, g_, mean, rstd, dgamma=dg, dbeta=db, colsum=cs, accumulate=True), B * 8)
    rec("bwd_all16_add", lambda: ops.sandwich_ln_bwd(dy16, x16, g_, mean, rstd, add_in=dy16, dgamma=dg, dbeta=db, accumulate=True), B * 8)
    rec("bwd_all16_drop_colsum", lambda: ops.sandwich_ln_bwd(dy16, x16, g_, mean, rstd, dropout=(0.1, 1, 2), dgamma=dg, dbeta=db, colsum=cs, accumulate=True), B * 6)
    print(json.dumps(row), flush=True)
