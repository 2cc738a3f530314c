"""CPU restatement of the `cogview_amd.ops` entry points a GPT train step calls -- TEST INFRASTRUCTURE ONLY.

`install(monkeypatch_like)` replaces those functions on the `cogview_amd.ops` module with plain torch-CPU code of the same
signatures and semantics (include/cogview_hip.h is the specification; fp32 arithmetic, results rounded to the storage type
where the kernels round).  It exists so that the HOST side of the path -- the `mpu` / `model` / `fp16` mirrors, the fused
layer Function, the gradient arena, the fused optimizer's control flow, and the REFERENCE's own driver functions calling them
(tests/test_reference_drivers_cpu.py) -- can execute end to end in a container without a GPU.  Nothing in the product
imports this file; the product path fails loudly without the HIP library (tests/test_abi.py).  Dropout IS emulated, with the
kernels' own counter-based generator as oracle/cogview_oracle.py restates it (dropout_keep_mask / attention_keep_mask: a mask is a
pure function of (seed, stream id, element index), so forward, backward and a recompute see the same one); the decode and the
sparse-training forms are not emulated (the gathered form of sparse generation is).
"""
import math

import torch

_HALF = (torch.float16, torch.bfloat16)


def _gelu(x):
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * x * (1.0 + 0.044715 * x * x)))


def _gelu_grad(x):
    u = 0.7978845608028654 * x * (1.0 + 0.044715 * x * x)
    t = torch.tanh(u)
    return 0.5 * (1.0 + t) + 0.5 * x * (1.0 - t * t) * 0.7978845608028654 * (1.0 + 3.0 * 0.044715 * x * x)


def _no_dropout(d):
    assert d is None or d[0] == 0.0, "tests/cpu_ops.py: no dropout in this form"


def _keep(shape, d):
    """Scaled keep mask (0 or 65536 / (65536 - thr16)) of the element-wise convention, element index = flat index; None for p = 0."""
    if d is None or float(d[0]) == 0.0:
        return None
    import numpy as np
    from oracle import cogview_oracle as O
    n = 1
    for v in shape:
        n *= int(v)
    return torch.from_numpy(np.ascontiguousarray(O.dropout_keep_mask(n, float(d[0]), int(d[1]), int(d[2])))).view(tuple(shape))


def new_absmax_slot(device):
    return torch.zeros(1, dtype=torch.float32)


def _publish(slot, t):
    if slot is not None:
        slot.copy_(torch.maximum(slot, t.detach().float().abs().max().view(1)))


def absmax(x, out=None):
    if out is None:
        out = new_absmax_slot(x.device)
    _publish(out, x)
    return out


def gemm(a, b, trans_a=False, trans_b=False, out=None, bias=None, gelu=False, gelu_aux=None, dgelu_aux=None, dropout=None,
         absmax=None, accumulate=False, splitk=None, out_dtype=None, variant=0, colsum_out=None, colsum_accumulate=True,
         gelu_daux=None, mul_aux=None, dropout_row0=0):
    A = a.float().t() if trans_a else a.float()
    B = b.float() if trans_b else b.float().t()
    c = A @ B
    if bias is not None:
        c = c + bias.float()
    if gelu:
        pre = c
        if gelu_aux is not None:
            gelu_aux.copy_(pre)
        if gelu_daux is not None:
            gelu_daux.copy_(_gelu_grad(pre))
        c = _gelu(pre)
    if dgelu_aux is not None:
        c = c * _gelu_grad(dgelu_aux.float())
    if mul_aux is not None:
        c = c * mul_aux.float()
    # epilogue order: ... -> dropout -> +C -> round (cogview_hip.h:49); dropout_row0: the rows are a chunk of a larger tensor
    keep = _keep((int(dropout_row0) + c.shape[0], c.shape[1]), dropout)
    if keep is not None:
        keep = keep[int(dropout_row0):]
    if keep is not None:
        c = c * keep
    if out is None:
        assert not accumulate
        out = torch.empty(c.shape, dtype=out_dtype or a.dtype)
    if accumulate:
        c = c + out.float()
    out.copy_(c)
    if keep is not None and not accumulate and out.dtype != torch.float32:
        # marked zeros (gemm_shared.cuh epilogue8): dropped -> -0.0, a kept value that rounds to a zero -> +0.0
        out.copy_(torch.where(keep == 0, torch.full_like(out, -0.0), torch.where(out == 0, torch.zeros_like(out), out)))
    _publish(absmax, out)
    if colsum_out is not None:
        colsum(out, out=colsum_out, accumulate=colsum_accumulate)
    return out


def gemm_grouped(problems, trans_a=True, trans_b=True, accumulate=True):
    assert 1 <= len(problems) <= 16
    for pr in problems:
        a, b, out = pr[:3]
        assert out.shape[0] >= 1 and a.stride(1) == 1 and b.stride(1) == 1 and out.stride(1) == 1
        gemm(a, b, trans_a=trans_a, trans_b=trans_b, out=out, accumulate=pr[3] if len(pr) > 3 else accumulate)


def colsum(dy, out=None, accumulate=False):
    s = dy.float().sum(0)
    if out is None:
        assert not accumulate
        out = torch.empty(dy.shape[1], dtype=dy.dtype)
    out.copy_(s + out.float() if accumulate else s)
    return out


def sandwich_ln_fwd(x, gamma, beta, eps, absmax_in, residual=None, absmax_out=None, save_stats=True):
    h = x.shape[-1]
    xf = x.reshape(-1, h).float()
    e = float(eps)
    if absmax_in is not None:
        c = float(absmax_in) * 0.125
        e = e * c * c
    mean = xf.mean(1)
    var = ((xf - mean[:, None]) ** 2).mean(1)
    rstd = 1.0 / torch.sqrt(var + e)
    y = (xf - mean[:, None]) * rstd[:, None] * gamma.float() + beta.float()
    if residual is not None and residual.dtype == torch.float32:
        y = residual.reshape(-1, h) + y                              # the fp32 stream: no rounding in between
    else:
        y = y.to(gamma.dtype)
        if residual is not None:
            y = (y.float() + residual.reshape(-1, h).float()).to(gamma.dtype)
    _publish(absmax_out, y)
    return y.view(x.shape), (mean if save_stats else None), (rstd if save_stats else None)


def _accum_param_grad(dst, val, accumulate):
    if dst is not None:
        dst.copy_(val + dst.float() if accumulate else val)


def sandwich_ln_bwd(dy, x, gamma, mean, rstd, add_in=None, dropout=None, dgamma=None, dbeta=None, colsum=None, accumulate=False,
                    marked=False):
    h = x.shape[-1]
    dyf, xf = dy.reshape(-1, h).float(), x.reshape(-1, h).float()
    xh = (xf - mean[:, None]) * rstd[:, None]
    g = dyf * gamma.float()
    dx = rstd[:, None] * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
    keep = _keep(dx.shape, dropout)                                  # dx = [add_in +] mask(LN'(dy)): the producing GEMM's dropout, replayed
    if marked and keep is not None:                                  # ... or read from x's marked zeros (-0.0 <=> dropped)
        dropped = (x.reshape(-1, h) == 0) & torch.signbit(x.reshape(-1, h))
        p_ = int(float(dropout[0]) * 65536.0 + 0.5)
        from_x = torch.where(dropped, torch.zeros((), dtype=torch.float32), torch.tensor(65536.0 / (65536.0 - p_), dtype=torch.float32))
        assert torch.equal(from_x, keep.float().view(from_x.shape)), "marked zeros of x disagree with the regenerated mask"
    if keep is not None:
        dx = dx * keep
    if add_in is not None:
        dx = dx + add_in.reshape(-1, h).float()
    dx = dx.to(x.dtype)                                              # fp32 for the stream forms' fp32 x, else the storage type
    _accum_param_grad(dgamma, (dyf * xh).sum(0), accumulate)
    _accum_param_grad(dbeta, dyf.sum(0), accumulate)
    _accum_param_grad(colsum, dx.float().sum(0), accumulate)
    return dx.view(x.shape)


def sandwich_ln_bwd_pair(dc, y, gamma2, mean2, rstd2, dout, ao, gamma3, mean3, rstd3, dropout_p=0.0, dgamma2=None, dbeta2=None,
                         dgamma3=None, dbeta3=None, colsum=None, accumulate=False):
    """cogv_sandwich_ln_bwd_pair as the two launches it replaces (the mask of the second one read from ao's marked zeros)."""
    dy = sandwich_ln_bwd(dc, y, gamma2, mean2, rstd2, add_in=dout, dgamma=dgamma2, dbeta=dbeta2, accumulate=accumulate)
    h = ao.shape[-1]
    dyf, xf = dy.reshape(-1, h).float(), ao.reshape(-1, h).float()
    xh = (xf - mean3[:, None]) * rstd3[:, None]
    g = dyf * gamma3.float()
    dx = rstd3[:, None] * (g - g.mean(1, keepdim=True) - xh * (g * xh).mean(1, keepdim=True))
    if float(dropout_p) > 0.0:
        thr = int(float(dropout_p) * 65536.0 + 0.5)
        dropped = (ao.reshape(-1, h) == 0) & torch.signbit(ao.reshape(-1, h))
        dx = torch.where(dropped, torch.zeros((), dtype=torch.float32), dx * (65536.0 / (65536.0 - thr)))
    dx = dx.to(ao.dtype)
    _accum_param_grad(dgamma3, (dyf * xh).sum(0), accumulate)
    _accum_param_grad(dbeta3, dyf.sum(0), accumulate)
    _accum_param_grad(colsum, dx.float().sum(0), accumulate)
    return dy, dx.view(ao.shape)


def _visible(s_q, s_k, sep):
    i = torch.arange(s_q)[:, None]
    j = torch.arange(s_k)[None, :]
    return (j <= i + (s_k - s_q)) | (j < int(sep) + (s_k - s_q))      # include/cogview_hip.h: the memory shifts the visible prefix too


def _attn_keep(q, k, d):
    if d is None or float(d[0]) == 0.0:
        return None
    import numpy as np
    from oracle import cogview_oracle as O
    b, s_q, H, _ = q.shape
    return torch.from_numpy(np.ascontiguousarray(O.attention_keep_mask(b, H, s_q, k.shape[1], float(d[0]), int(d[1]), int(d[2]))))


def _attn(q, k, v, sep, keep=None):
    # q [b, s_q, H, 64] -> scores [b, H, s_q, s_k]; reference order: Q / sqrt(d) first (mpu/sparse_transformer.py:653-659)
    qf, kf, vf = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))
    s = (qf / math.sqrt(q.shape[-1])) @ kf.transpose(-1, -2)
    vis = _visible(q.shape[1], k.shape[1], sep)
    s = torch.where(vis, s, torch.full_like(s, -10000.0))
    lse = torch.logsumexp(s, -1)
    p = torch.softmax(s, -1)
    if keep is not None:                                             # dropout on the probabilities (mpu/sparse_transformer.py:667-669)
        p = p * keep
    o = p @ vf
    return o.permute(0, 2, 1, 3), lse


def attention_fwd(q, k, v, sep=0, dropout=None, kv_index=None, sparse=None, keep_bits=False, mask=None):
    assert sparse is None and mask is None
    if kv_index is not None:
        _no_dropout(dropout)
        # gathered form (sparse_attention_inference): key slot j of row i is position kv_index[i, j]; the last s_q slots are the
        # queries themselves, left-to-right among them (include/cogview_hip.h, cogv_attention_fwd with kv_index)
        assert int(sep) == 0
        idx = kv_index.long()
        k = torch.stack([k[i, idx[i]] for i in range(k.shape[0])])
        v = torch.stack([v[i, idx[i]] for i in range(v.shape[0])])
    o, lse = _attn(q, k, v, sep, _attn_keep(q, k, dropout))
    o = o.contiguous().to(q.dtype)
    return (o, lse, None) if keep_bits else (o, lse)      # (no stored keep bits: the backward regenerates the mask from the same counters)


def attention_bwd(dout, q, k, v, o, lse, sep=0, dropout=None, dq=None, dk=None, dv=None, colsum_out=None, colsum_accumulate=True,
                  keep_bits=None, mask=None):
    qg, kg, vg = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    with torch.enable_grad():
        out, _ = _attn(qg, kg, vg, sep, _attn_keep(q, k, dropout))
        gq, gk, gv = torch.autograd.grad(out, (qg, kg, vg), dout.float())
    outs = []
    for dst, g in ((dq, gq), (dk, gk), (dv, gv)):
        if dst is None:
            dst = torch.empty(g.shape, dtype=q.dtype)
        dst.copy_(g)
        outs.append(dst)
    if colsum_out is not None:
        H = q.shape[2]
        for i, t in enumerate(outs):
            colsum(t.reshape(-1, H * 64), out=colsum_out[i * H * 64:(i + 1) * H * 64], accumulate=colsum_accumulate)
    return tuple(outs)


def embedding_fwd(ids, table, vocab_start, pos_ids=None, pos_table=None, dropout=None, absmax_out=None, x_in=None, out_f32=False):
    src = table if table is not None else x_in
    if ids is not None:
        local = ids - vocab_start
        ok = (local >= 0) & (local < table.shape[0])
        x = table.float()[local.clamp(0, table.shape[0] - 1)] * ok[..., None]
    else:
        x = x_in.float()
    if pos_table is not None:
        x = x + pos_table.float()[pos_ids.expand(x.shape[:-1]).clamp(0, pos_table.shape[0] - 1)]      # the kernel clamps position ids
    keep = _keep(x.shape, dropout)
    if keep is not None:
        x = x * keep
    out = x if out_f32 else x.to(src.dtype)
    _publish(absmax_out, out)
    return out.contiguous()


def embedding_bwd(dout, ids, dtable, vocab_start, pos_ids=None, dpos=None, dropout=None, dx=None):
    h = dout.shape[-1]
    d2 = dout.reshape(-1, h).float()
    keep = _keep(d2.shape, dropout)
    if keep is not None:
        d2 = d2 * keep
    if dtable is not None and ids is not None:
        local = (ids - vocab_start).reshape(-1)
        ok = (local >= 0) & (local < dtable.shape[0])
        acc = torch.zeros(dtable.shape, dtype=torch.float32).index_add_(0, local[ok], d2[ok])
        dtable.copy_(dtable.float() + acc)
    if dpos is not None:
        p = pos_ids.expand(dout.shape[:-1]).reshape(-1).clamp(0, dpos.shape[0] - 1)
        acc = torch.zeros(dpos.shape, dtype=torch.float32).index_add_(0, p, d2)
        dpos.copy_(dpos.float() + acc)
    if dx is not None:
        dx.copy_(d2.view(dout.shape))


def ce_fwd(logits2d, target1d, vocab_start, want_loss=True):
    lf = logits2d.float()
    rowmax = lf.max(1).values
    sumexp = torch.exp(lf - rowmax[:, None]).sum(1)
    local = target1d - vocab_start
    ok = (local >= 0) & (local < lf.shape[1])
    pred = lf.gather(1, local.clamp(0, lf.shape[1] - 1)[:, None])[:, 0] * ok
    loss = torch.log(sumexp) + rowmax - pred if want_loss else None
    return rowmax, sumexp, pred, loss


def ce_bwd(logits2d, target1d, vocab_start, gmax, gsum, grad, out=None):
    lf = logits2d.float()
    d = torch.exp(lf - gmax[:, None]) / gsum[:, None]
    local = target1d - vocab_start
    ok = (local >= 0) & (local < lf.shape[1])
    rows = torch.arange(lf.shape[0])[ok]
    d[rows, local[ok]] -= 1.0
    d = d * grad[:, None]
    if out is None:
        out = torch.empty_like(logits2d)
    out.copy_(d)
    return out


def grad_stats(flat_grads, chunk_start, chunk_len, chunk_norm, stats):
    g = flat_grads.double()
    tot, bad = 0.0, 0.0
    for s, n, cnt in zip(chunk_start.tolist(), chunk_len.tolist(), chunk_norm.tolist()):
        piece = g[s:s + n]
        if not bool(torch.isfinite(piece).all()):
            bad = 1.0
        if cnt:
            tot += float((piece * piece).sum())
    stats[0] += tot
    if bad:
        stats[1] = 1.0


def adamw_step(params, grads, master, exp_avg, exp_avg_sq, chunk_start, chunk_len, chunk_group, lrs, wds, beta1, beta2, eps, step,
               inv_loss_scale=1.0, max_grad_norm=0.0, stats=None, sumsq_override=None, bias_correction=True, adam_w_mode=True):
    if stats is not None and float(stats[1]) != 0.0:
        return
    gscale = float(inv_loss_scale)
    if max_grad_norm > 0.0 and stats is not None:
        ss = float(sumsq_override) if sumsq_override is not None else float(stats[0])
        coef = max_grad_norm / (math.sqrt(ss) * inv_loss_scale + 1.0e-6)
        if coef < 1.0:
            gscale *= coef
    bc1 = 1.0 - beta1 ** step if bias_correction else 1.0
    bc2 = 1.0 - beta2 ** step if bias_correction else 1.0
    for s, n, grp in zip(chunk_start.tolist(), chunk_len.tolist(), chunk_group.tolist()):
        lr, wd = float(lrs[grp & 7]), float(wds[grp & 7])
        sl = slice(s, s + n)
        g = grads[sl].float() * gscale
        w, m, v = master[sl], exp_avg[sl], exp_avg_sq[sl]
        if not adam_w_mode:
            g = g + wd * w
        m.mul_(beta1).add_(g, alpha=1.0 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        upd = (m / bc1) / (torch.sqrt(v) / math.sqrt(bc2) + eps)
        if adam_w_mode:
            upd = upd + wd * w
        w.sub_(lr * upd)
        params[sl].copy_(w)


def cast_flat(src_half, dst_f32):
    dst_f32.copy_(src_half)


def cast_flat_back(src_f32, dst_half):
    dst_half.copy_(src_f32)


def scale(x, s):
    return (x.float() * s).to(x.dtype)


def gelu_fwd(x):
    return _gelu(x.float()).to(x.dtype)


def gelu_bwd(dy, x):
    return (dy.float() * _gelu_grad(x.float())).to(x.dtype)


def dropout(x, p, seed, stream_id, absmax_out=None):
    keep = _keep(x.shape, (p, seed, stream_id))
    y = x.clone() if keep is None else (x.float() * keep).to(x.dtype)
    _publish(absmax_out, y)
    return y


def add(a, b, absmax_out=None):
    out = (a.float() + b.float()).to(torch.float32 if a.dtype == torch.float32 else a.dtype)
    _publish(absmax_out, out)
    return out


def gemm_reserve_cus(n):
    """No CUs on a CPU: the setting is recorded and handed back like the library does."""
    global _RESERVED
    prev = _RESERVED
    if int(n) >= 0:
        _RESERVED = int(n)
    return prev


_RESERVED = 0

NAMES = ("new_absmax_slot", "absmax", "gemm", "gemm_reserve_cus", "gemm_grouped", "colsum", "sandwich_ln_fwd", "sandwich_ln_bwd", "sandwich_ln_bwd_pair", "attention_fwd",
         "attention_bwd", "embedding_fwd", "embedding_bwd", "ce_fwd", "ce_bwd", "grad_stats", "adamw_step", "cast_flat",
         "cast_flat_back", "scale", "add", "gelu_fwd", "gelu_bwd", "dropout")


def install(setattr_fn=None):
    """Replace the entry points on the cogview_amd.ops module (setattr_fn: e.g. monkeypatch.setattr; default: plain setattr)."""
    from cogview_amd import ops
    g = globals()
    for n in NAMES:
        (setattr_fn or setattr)(ops, n, g[n])
    return ops
