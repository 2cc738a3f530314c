"""Sparse attention training form (pivots + blocked window, slot space) against dense attention at the reference's
sparse configuration (4096 positions, query_window 128, key_window_times 6, 768 pivots; 40 heads), plus the dense
4B shape as a regression check of the shared kernels."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cogview_amd.mpu  # noqa: F401
from cogview_amd import ops
from cogview_amd import functional as F_
from tools.microbench import timeit

dt = torch.bfloat16
drop = (0.1, 1, 2)


def dense(b, H, s):
    qkv = torch.randn(b, s, 3 * H * 64, device="cuda", dtype=dt)
    q, k, v = [qkv[:, :, i * H * 64:(i + 1) * H * 64].view(b, s, H, 64) for i in range(3)]
    do = torch.randn(b, s, H, 64, device="cuda", dtype=dt)
    o, lse = ops.attention_fwd(q, k, v, dropout=drop)
    tf = timeit(lambda: ops.attention_fwd(q, k, v, dropout=drop), iters=10)
    tb = timeit(lambda: ops.attention_bwd(do, q, k, v, o, lse, dropout=drop), iters=10)
    pairs = b * H * s * (s + 1) / 2
    print(f"dense  b={b} H={H} s={s}: fwd {tf*1e6:8.1f} us ({4*64*pairs/tf/1e12:6.1f} TF)  bwd {tb*1e6:8.1f} us ({10*64*pairs/tb/1e12:6.1f} TF)", flush=True)
    return tf, tb


def sparse(b, H, s, w, times, n_piv):
    qkv = torch.randn(b, s, 3 * H * 64, device="cuda", dtype=dt)
    q, k, v = [qkv[:, :, i * H * 64:(i + 1) * H * 64].view(b, s, H, 64) for i in range(3)]
    do = torch.randn(b, s, H, 64, device="cuda", dtype=dt)
    piv = torch.stack([torch.randperm(s, device="cuda")[:n_piv] for _ in range(b)])
    tab, inv = F_.sparse_pivot_plan(piv, s, w, times)
    sp = (w, n_piv, math.log(s // n_piv))
    o, lse = ops.attention_fwd(q, k, v, dropout=drop, kv_index=tab, sparse=sp)
    tf = timeit(lambda: ops.attention_fwd(q, k, v, dropout=drop, kv_index=tab, sparse=sp), iters=10)
    tb = timeit(lambda: ops.sparse_attention_bwd(do, q, k, v, o, lse, tab, sp, inv, times, dropout=drop), iters=10)
    tp = timeit(lambda: F_.sparse_pivot_plan(piv, s, w, times), iters=10)
    print(f"sparse b={b} H={H} s={s} w={w}x{times} piv={n_piv}: fwd {tf*1e6:8.1f} us  bwd (dq + slot dk/dv + reduce) {tb*1e6:8.1f} us  plan {tp*1e6:.1f} us", flush=True)
    return tf, tb


for _ in range(2):
    dense(24, 40, 1088)
for _ in range(2):
    d = dense(4, 40, 4096)
    s_ = sparse(4, 40, 4096, 128, 6, 768)
    print(f"   sparse / dense time: fwd {s_[0]/d[0]:.2f}  bwd {s_[1]/d[1]:.2f}", flush=True)
