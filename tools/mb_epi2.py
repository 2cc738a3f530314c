"""Backward GEMM epilogues at the 4B shapes (GPU box): dgrad of h->4h with the stored GeLU derivative + column sums,
plain dgrads, and the accumulating weight gradient."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cogview_amd import ops
from tools.microbench import timeit
dt = torch.bfloat16
_w = torch.randn(8192, 8192, device="cuda", dtype=dt)
for _ in range(20): _w @ _w
M = 26112
for N, K in ((10240, 2560), (2560, 10240), (2560, 7680)):
    dy = torch.randn(M, K, device="cuda", dtype=dt); w = torch.randn(K, N, device="cuda", dtype=dt) * 0.02
    aux = torch.randn(M, N, device="cuda", dtype=dt); cs = torch.zeros(N, device="cuda", dtype=dt)
    fl = 2.0 * M * N * K
    row = {"N": N, "K": K}
    for name, fn in (("plain", lambda: ops.gemm(dy, w, trans_b=True)),
                     ("mulaux_colsum", lambda: ops.gemm(dy, w, trans_b=True, mul_aux=aux, colsum_out=cs)),
                     ("dgelu_colsum", lambda: ops.gemm(dy, w, trans_b=True, dgelu_aux=aux, colsum_out=cs))):
        t = timeit(fn, iters=10, warm=2)
        row[name + "_us"] = round(t * 1e6); row[name + "_TF"] = round(fl / t / 1e12)
    print(json.dumps(row), flush=True)
