"""Golden vectors for the "next" row of SURVEY section 8f: CogView's sparse attention (pivot + window, joint softmax),
produced by the REFERENCE's own functions (mpu/sparse_transformer.py:629-750) on CPU.

    python oracle/gen_golden_sparse.py        (build container only; writes tests/golden/sparse_attention.npz)

Shims: the same as oracle/gen_golden.py (nothing in the sparse path needs more).  The pivot mask is built exactly as
the training branch of GPT2ParallelTransformer.forward does (mpu/sparse_transformer.py:491-496, 564-568).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import install_shims, npz  # noqa: E402


def main():
    mpu, st = install_shims()
    torch.manual_seed(77)
    b, nh, hn, w, times, n_piv = 2, 2, 64, 16, 3, 12
    s = 6 * w                                      # training form: s % w == 0
    g = torch.Generator().manual_seed(5)
    q, k, v = [torch.randn(b, nh, s, hn, generator=g) for _ in range(3)]
    n_txt = [5, 9]                                 # text tokens come first and are always pivots
    pivot_idx = []
    for i in range(b):
        img = torch.arange(n_txt[i], s)
        pick = img[torch.randperm(len(img), generator=g)[:n_piv - n_txt[i]]]
        pivot_idx.append(torch.cat((torch.arange(n_txt[i]), pick)))
    pivot_idx = torch.stack(pivot_idx)
    # rmask as in GPT2ParallelTransformer.forward (is_sparse == 1)
    gq = s // w
    tmp = torch.ones((gq - times + 1, w, w))
    tmp = torch.tril(1 - torch.block_diag(*tmp))
    rmask = torch.nn.functional.pad(tmp, (0, (times - 1) * w, (times - 1) * w, 0))
    pam = rmask.expand(b, s, s).gather(dim=-1, index=pivot_idx.unsqueeze(1).expand(b, s, n_piv))
    qg, kg, vg = [t.clone().requires_grad_(True) for t in (q, k, v)]
    ctx = st.sparse_attention(qg, kg, vg, pivot_idx, pam, query_window=w, key_window_times=times, attention_dropout=None)
    do = torch.randn(ctx.shape, generator=g)
    ctx.backward(do)
    # inference form: the last sq queries of a longer key sequence against pivots + trailing window
    sq, sk = 3, s - 7
    left = max(0, sk - times * w)
    window_idx = torch.arange(left, sk).expand(b, -1)
    pw_idx = torch.cat((pivot_idx.clamp(max=left - 1 if left > 0 else 0), window_idx), dim=-1)
    ci = st.sparse_attention_inference(q[:, :, sk - sq:sk], k[:, :, :sk], v[:, :, :sk], pw_idx)
    npz("sparse_attention.npz", cfg=[b, nh, s, hn, w, times, n_piv], q=q, k=k, v=v, pivot_idx=pivot_idx, rmask=rmask,
        pivot_attention_mask=pam, ctx=ctx, dout=do, dq=qg.grad, dk=kg.grad, dv=vg.grad,
        inf_cfg=[sq, sk], pw_idx=pw_idx, inf_ctx=ci)


if __name__ == "__main__":
    main()
