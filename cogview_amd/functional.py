"""Autograd plumbing over the HIP ops.

Two tiers:
  * op-level Functions (linear, Sandwich-LN, attention, embedding, GeLU, dropout, cross entropy): compose
    freely -- this is what the individual `mpu` modules call, and it handles every shape the reference's modules
    accept (memories, "sep" masks, model parallelism);
  * `transformer_layer` -- one Function for a whole GPT2ParallelTransformerLayer (the 9-kernel forward /
    15-kernel backward chain with fused epilogues).  This is the training hot path.

Parameter gradients are ACCUMULATED IN PLACE into `param.grad` by the backward kernels (GEMM epilogue
`+= C`, column-sum kernels with accumulate) instead of being returned to autograd: with the parameters living
in one flat arena (cogview_amd/arena.py) `param.grad` is a view of one flat gradient buffer, so the
data-parallel all-reduce, the overflow check, the norm and the Adam step are each ONE kernel / collective over
that buffer (the reference does each per tensor: 388 tensors for the 24-layer model, 772 for 48 layers).
"""
import contextlib
import math
import os

import torch

from . import ops
from .mpu import mappings
from .mpu import random as mpu_random
from .mpu.initialize import get_model_parallel_group, mp_rank_or_0, mp_world_size_or_1


def grad_buffer(p):
    """The tensor backward kernels accumulate into (allocated zeroed on first use when no arena exists)."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


def grad_accumulate(*params):
    """The accumulate flag for a kernel that adds into the gradients of `params`: False (overwrite) only when all of them
    are arena gradients untouched since a lazy zero_grad (arena.ParamArena.take_fresh)."""
    params = [p for p in params if p is not None]
    arenas = [getattr(p, "_cogv_arena", (None,))[0] for p in params]
    if params and all(a is not None and a is arenas[0] for a in arenas):
        return not arenas[0].take_fresh(params)
    for p, a in zip(params, arenas):          # mixed / no arena: zero what is untouched, then accumulate
        if a is not None:
            a.ensure_zeroed(p)                # (take_fresh([p]) would hand back "overwrite" WITHOUT zeroing)
    return True


def _mp_allreduce(t):
    if mp_world_size_or_1() > 1:
        torch.distributed.all_reduce(t, group=get_model_parallel_group())
    return t


def _mp_allreduce_start(t):
    """Start the model-parallel sum of `t` (in place) without blocking the compute stream: RCCL runs it on its own
    stream behind everything enqueued so far; returns a handle for _mp_allreduce_finish (None when not model parallel)."""
    if mp_world_size_or_1() > 1:
        return torch.distributed.all_reduce(t, group=get_model_parallel_group(), async_op=True)
    return None


def _mp_allreduce_finish(work):
    if work is not None:
        work.wait()          # nccl: the compute stream waits for the collective; gloo (tests): the host waits


@contextlib.contextmanager
def _room_for_collective():
    """GEMMs launched inside run on all CUs but COGV_MP_RESERVE_CUS (default 16): an all-reduce in flight on RCCL's stream
    gets CUs for its channel workgroups instead of waiting for the persistent launch to end (ops.gemm_reserve_cus)."""
    n = int(os.environ.get("COGV_MP_RESERVE_CUS", "16"))
    if n <= 0 or mp_world_size_or_1() == 1:
        yield
        return
    prev = ops.gemm_reserve_cus(n)
    try:
        yield
    finally:
        ops.gemm_reserve_cus(prev)


def mp_row_chunks(rows):
    """Row chunks of a row-parallel Linear's output under model parallelism (COGV_MP_ROW_CHUNKS, default 4; 1 = the
    whole tensor in one GEMM + one all-reduce): [(row0, row1)], boundaries on multiples of 256 rows -- the GEMM's tile
    height, so the chunks together run exactly the tiles of the whole-tensor launch."""
    n = max(1, int(os.environ.get("COGV_MP_ROW_CHUNKS", "4")))
    tiles = (rows + 255) // 256
    n = min(n, tiles)
    cuts = [min(rows, 256 * ((tiles * i + n - 1) // n)) for i in range(n + 1)]
    return [(cuts[i], cuts[i + 1]) for i in range(n) if cuts[i + 1] > cuts[i]]


def _row_parallel_chunked(x2d, weight, bias, drop, absmax_slot):
    """Forward of a row-parallel Linear under model parallelism (mpu/layers.py:312-326: Y = reduce(X_i A_i) + b) with the
    exchange taken off the critical path as far as the data allows: the output is computed in row chunks, chunk i's
    all-reduce is started (RCCL's own stream) as soon as its GEMM is queued and runs under chunk i + 1's GEMM; the compute
    stream waits for all of them only before the one thing that needs the whole tensor -- Sandwich-LN's abs-max, a
    read-only pass over the reduced output.  Bias (rank 0 only) and hidden dropout ride in every chunk's GEMM epilogue as in
    the whole-tensor form (dropout is linear per element and its mask depends only on (seed, stream, element index), which
    `dropout_row0` keeps identical to the whole-tensor call): results are bit-identical to one GEMM + one all-reduce --
    same k-loop per element, same element-wise sum across ranks."""
    rows, n_out = x2d.shape[0], weight.shape[0]
    out = torch.empty((rows, n_out), dtype=x2d.dtype, device=x2d.device)
    works = []
    for i, (r0, r1) in enumerate(mp_row_chunks(rows)):
        if i == 0:                                   # nothing in flight yet: all CUs
            ops.gemm(x2d[r0:r1], weight, bias=bias, dropout=drop, dropout_row0=r0, out=out[r0:r1])
        else:
            with _room_for_collective():
                ops.gemm(x2d[r0:r1], weight, bias=bias, dropout=drop, dropout_row0=r0, out=out[r0:r1])
        works.append(_mp_allreduce_start(out[r0:r1]))
    for w in works:
        _mp_allreduce_finish(w)
    ops.absmax(out, absmax_slot)
    return out


def _drop(p, training, attention=False):
    """None or (p, seed, stream_id) drawn from the RNG tracker."""
    if not training or p <= 0.0:
        return None
    seed, sid = mpu_random.attention_dropout_stream() if attention else mpu_random.next_dropout_stream()
    return (p, seed, sid)


# =============================================================================================== op level
class _Linear(torch.autograd.Function):
    """y = x W^T (+ b); W is [out, in] (F.linear convention, mpu/layers.py:243,319)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x.reshape(-1, x.shape[-1])
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        y = ops.gemm(x2, weight, bias=bias)
        ctx.save_for_backward(x2, weight)
        ctx.bias = bias
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        dy2 = dy.reshape(-1, dy.shape[-1])
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dy2, weight, trans_b=True).view(ctx.xshape)
        if weight.requires_grad:
            ops.gemm(dy2, x2, trans_a=True, trans_b=True, out=grad_buffer(weight), accumulate=grad_accumulate(weight))
        if ctx.bias is not None and ctx.bias.requires_grad:
            ops.colsum(dy2, out=grad_buffer(ctx.bias), accumulate=grad_accumulate(ctx.bias))
        return dx, None, None


def linear(x, weight, bias=None):
    return _Linear.apply(x, weight, bias)


class _SandwichLN(torch.autograd.Function):
    """y = [residual +] SandwichLN(x).  The fp32 residual stream is recognised by dtype (ops.sandwich_ln_fwd): an fp32
    x yields a 16-bit y (LN1, LN2, final LN); an fp32 residual yields the fp32 stream residual + LN(x) (LN3, LN4)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, absmax, residual, absmax_out):
        xc = x if x.is_contiguous() else x.contiguous()
        if absmax is None:
            absmax = ops.absmax(xc)
        rc = None
        if residual is not None:
            rc = residual if residual.is_contiguous() else residual.contiguous()
        y, mean, rstd = ops.sandwich_ln_fwd(xc, weight, bias, eps, absmax, residual=rc, absmax_out=absmax_out)
        ctx.save_for_backward(xc, weight, mean, rstd)
        ctx.bias = bias
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, weight, mean, rstd = ctx.saved_tensors
        dyc = dy if dy.is_contiguous() else dy.contiguous()
        dg = grad_buffer(weight) if weight.requires_grad else None
        db = grad_buffer(ctx.bias) if ctx.bias.requires_grad else None
        dx = ops.sandwich_ln_bwd(dyc, xc, weight, mean, rstd, dgamma=dg, dbeta=db,
                                 accumulate=grad_accumulate(weight if dg is not None else None, ctx.bias if db is not None else None))
        return dx, None, None, None, None, (dyc if ctx.has_res else None), None


def sandwich_layer_norm(x, weight, bias, eps=1e-5, residual=None):
    """LayerNorm(x / (max|x| / 8)) -- mpu/sparse_transformer.py:40-44; with `residual` the sum residual + LN(x) of
    :329 / :340 (carrying its abs-max for the LayerNorm that follows)."""
    if residual is None:
        return _SandwichLN.apply(x, weight, bias, eps, getattr(x, "_cogv_absmax", None), None, None)
    slot = ops.new_absmax_slot(x.device)
    y = _SandwichLN.apply(x, weight, bias, eps, getattr(x, "_cogv_absmax", None), residual, slot)
    y._cogv_absmax = slot
    return y


class _Attention(torch.autograd.Function):
    """q [b,s_q,H,64], k/v [b,s_k,H,64] (strided views allowed) -> o [b,s_q,H,64]."""

    @staticmethod
    def forward(ctx, q, k, v, sep, dropout, mask=None):
        o, lse, bits = ops.attention_fwd(q, k, v, sep=sep, dropout=dropout, keep_bits=True, mask=mask)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.sep, ctx.dropout, ctx.keep_bits, ctx.mask = sep, dropout, bits, mask
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        doc = do if do.is_contiguous() else do.contiguous()
        dq, dk, dv = ops.attention_bwd(doc, q, k, v, o, lse, sep=ctx.sep, dropout=ctx.dropout, keep_bits=ctx.keep_bits,
                                       mask=ctx.mask)
        ctx.keep_bits = None
        return dq, dk, dv, None, None, None


def _as_bshd(t):
    """[b, np, s, hn] (usually a permuted view of [b, s, np, hn]) -> [b, s, np, hn] with packed heads."""
    u = t.permute(0, 2, 1, 3)
    if u.stride(3) != 1 or (u.shape[2] > 1 and u.stride(2) != u.shape[3]):
        u = u.contiguous()
    return u


def mask_to_sep(attention_mask, s_q, s_k):
    """Translate the reference's mask argument into the kernel's `sep`:  an int is the reference's "sep" form
    (mpu/sparse_transformer.py:477-489); a tensor that is the left-to-right mask (optionally with a visible prefix) --
    every mask the reference's training / generation paths build -- becomes its `sep` (causal block skipping in the
    kernels).  ANY OTHER tensor (per-sample masks, non-binary masks: mpu/sparse_transformer.py:661-663 multiplies by
    whatever it is given) returns None: the caller passes the tensor itself to the attention kernels' general-mask
    path (general_mask()).  Verified once per tensor object."""
    if isinstance(attention_mask, int):
        return attention_mask
    if attention_mask.numel() == 1:
        return int(attention_mask.item())
    key = (attention_mask._version, s_q, s_k)             # (the same tensor may be reused with another split of its elements)
    cached = getattr(attention_mask, "_cogv_sep", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    sep = None
    if attention_mask.numel() == s_q * s_k:
        m = attention_mask.reshape(s_q, s_k)
        off = s_k - s_q
        n0 = int(m[0].sum().item())
        cand = 0 if n0 <= off + 1 else n0 - off
        ref = torch.ones(s_q, s_k, device=m.device, dtype=m.dtype)
        ref[:, -s_q:] = torch.tril(ref[:, -s_q:])
        if cand > 0:
            ref[:, :cand + off] = 1
        if torch.equal(ref, m):
            sep = cand
    try:
        attention_mask._cogv_sep = (key, sep)
    except Exception:
        pass
    return sep


def general_mask(attention_mask, batch, s_q, s_k, dtype):
    """An arbitrary mask tensor in the form the kernels take: [B or 1, s_q, s_k], contiguous, storage type (the reference's
    masks are [1, 1, s_q, s_k] or [b, 1, s_q, s_k], broadcast over the heads).  Cached per tensor object, dtype and geometry.
    (The general-mask kernels read the mask element by element, uncoalesced: fine off the hot path, which takes the `sep` form.)"""
    key = (attention_mask._version, dtype, batch, s_q, s_k)
    cached = getattr(attention_mask, "_cogv_gmask", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    n = attention_mask.numel()
    if n == s_q * s_k:
        m = attention_mask.reshape(1, s_q, s_k)
    elif n == batch * s_q * s_k:
        m = attention_mask.reshape(batch, s_q, s_k)
    else:
        raise ValueError(f"attention mask of shape {tuple(attention_mask.shape)} does not broadcast to [{batch}, 1, {s_q}, {s_k}]")
    m = m.to(dtype).contiguous()
    try:
        attention_mask._cogv_gmask = (key, m)
    except Exception:
        pass
    return m


def standard_attention(query_layer, key_layer, value_layer, attention_mask, attention_dropout=None):
    """Drop-in for mpu/sparse_transformer.py:652-673; tensors are [b, np, s, hn] with hn = 64."""
    s_q, s_k = query_layer.shape[2], key_layer.shape[2]
    sep = mask_to_sep(attention_mask, s_q, s_k)
    drop = None
    if attention_dropout is not None:
        drop = _drop(attention_dropout.p, attention_dropout.training, attention=True)
    gm = None
    if sep is None:         # not a left-to-right mask: the tensor itself goes to the kernels (score * M - 10000 * (1 - M))
        gm, sep = general_mask(attention_mask, query_layer.shape[0], s_q, s_k, query_layer.dtype), 0
    o = _Attention.apply(_as_bshd(query_layer), _as_bshd(key_layer), _as_bshd(value_layer), sep, drop, gm)
    return o.permute(0, 2, 1, 3)


def sparse_slot_table(pivot_idx, s, w, times):
    """Index table of the sparse TRAINING form in slot space: [b, s // w, n_piv + times * w] int32.  Query block g
    attends its n_piv pivots (masked -- bit 31 -- unless the pivot lies before the block's window: the rmask rule of
    mpu/sparse_transformer.py:491-496) followed by the window keys (g - times + 1) * w ... (g + 1) * w - 1 (front
    padding masked)."""
    b, n_piv = pivot_idx.shape
    G = s // w
    dev = pivot_idx.device
    ks = (torch.arange(G, device=dev) - times + 1) * w                                   # first window key per block
    piv = pivot_idx.to(torch.int64).unsqueeze(1).expand(b, G, n_piv)
    piv = piv + (piv >= ks.view(1, G, 1)).to(torch.int64) * (1 << 31)
    win = ks.view(G, 1) + torch.arange(times * w, device=dev).view(1, -1)
    win = win.clamp(min=0) + (win < 0).to(torch.int64) * (1 << 31)
    tab = torch.cat((piv, win.unsqueeze(0).expand(b, G, times * w)), dim=-1)
    tab = torch.where(tab >= (1 << 31), tab - (1 << 32), tab)                            # as signed 32-bit patterns
    return tab.to(torch.int32).contiguous()


class _SparseAttention(torch.autograd.Function):
    """q, k, v [b,s,H,64]; tab = sparse_slot_table(...); inv [b,s] int32 pivot slot of each key or -1."""

    @staticmethod
    def forward(ctx, q, k, v, tab, inv, sparse, times, dropout):
        o, lse = ops.attention_fwd(q, k, v, sep=0, dropout=dropout, kv_index=tab, sparse=sparse)
        ctx.save_for_backward(q, k, v, o, lse, tab, inv)
        ctx.sparse, ctx.times, ctx.dropout = sparse, times, dropout
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse, tab, inv = ctx.saved_tensors
        doc = do if do.is_contiguous() else do.contiguous()
        dq, dk, dv = ops.sparse_attention_bwd(doc, q, k, v, o, lse, tab, ctx.sparse, inv, ctx.times, dropout=ctx.dropout)
        return dq, dk, dv, None, None, None, None, None


def sparse_pivot_plan(pivot_idx, s, w, times):
    """Everything the sparse training kernels derive from one pivot draw: (slot table, inverse pivot map).  A plan is
    shared by the layers that share the draw (one per checkpoint chunk in the reference, mpu/sparse_transformer.py:
    553-570)."""
    b, n_piv = pivot_idx.shape
    inv = torch.full((b, s), -1, dtype=torch.int32, device=pivot_idx.device)
    inv.scatter_(1, pivot_idx.to(torch.int64), torch.arange(n_piv, dtype=torch.int32, device=pivot_idx.device).expand(b, n_piv))
    return sparse_slot_table(pivot_idx, s, w, times), inv


def sparse_attention(q, k, v, pivot_idx, pivot_attention_mask, query_window=128, key_window_times=6,
                     attention_dropout=None):
    """Drop-in for mpu/sparse_transformer.py:675-725 (sparse attention, training form); [b, np, s, hn] tensors with
    hn = 64, s % query_window == 0, query_window % 128 == 0, distinct pivots per sample (the reference draws them with
    random.sample).  pivot_attention_mask is accepted for signature compatibility: the kernels rebuild it from the
    rule that defines it (rmask gathered at the pivots, :491-496 and :569 -- a pivot is visible to a query iff it lies
    before the query block's window), or take a precomputed plan (pivot_idx = the tuple sparse_pivot_plan returned)."""
    b, _np, s, _hn = q.shape
    w, times = int(query_window), int(key_window_times)
    if isinstance(pivot_idx, tuple):
        tab, inv = pivot_idx
    else:
        tab, inv = sparse_pivot_plan(pivot_idx, s, w, times)
    n_piv = tab.shape[2] - times * w
    drop = None
    if attention_dropout is not None:
        drop = _drop(attention_dropout.p, attention_dropout.training, attention=True)
    o = _SparseAttention.apply(_as_bshd(q), _as_bshd(k), _as_bshd(v), tab, inv,
                               (w, n_piv, math.log(s // n_piv)), times, drop)
    return o.permute(0, 2, 1, 3)


def sparse_attention_inference(q, k, v, pivot_and_window_idx, **kwargs):
    """Drop-in for mpu/sparse_transformer.py:727-750 (generation with sparse attention): the sq queries -- the last
    sq keys -- attend the gathered pivot + window keys, causally among themselves.  [b, np, s, hn] tensors, index
    [b, n_pivot_and_window]; inference only (no gradient).  The gather happens inside the attention kernel."""
    if torch.is_grad_enabled() and (q.requires_grad or k.requires_grad or v.requires_grad):
        raise NotImplementedError("sparse_attention_inference is the forward-only generation form (mpu/sparse_transformer.py:"
                                  "727-750); training uses sparse_attention (the slot-space form, with gradients)")
    idx = pivot_and_window_idx.to(torch.int32).contiguous()
    assert int(idx.max()) < k.shape[2] and int(idx.min()) >= 0
    o, _ = ops.attention_fwd(_as_bshd(q), _as_bshd(k), _as_bshd(v), sep=0, dropout=None, kv_index=idx)
    return o.permute(0, 2, 1, 3)


class _Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        xc = x if x.is_contiguous() else x.contiguous()
        ctx.save_for_backward(xc)
        return ops.gelu_fwd(xc)

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        return ops.gelu_bwd(dy if dy.is_contiguous() else dy.contiguous(), x)


def gelu(x):
    return _Gelu.apply(x)


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, drop):
        ctx.drop = drop
        return ops.dropout(x if x.is_contiguous() else x.contiguous(), *drop)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy if dy.is_contiguous() else dy.contiguous(), *ctx.drop), None


def dropout(x, p, training=True):
    drop = _drop(p, training)
    return x if drop is None else _Dropout.apply(x, drop)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.dtypes = (a.dtype, b.dtype)
        return ops.add(a if a.is_contiguous() else a.contiguous(), b if b.is_contiguous() else b.contiguous())

    @staticmethod
    def backward(ctx, dy):
        # the fp32 residual stream joined by a 16-bit branch: each input gets its gradient in ITS dtype (explicitly, not by
        # the autograd engine's silent cast)
        da, db = ctx.dtypes
        return (dy if dy.dtype == da else dy.to(da)), (dy if dy.dtype == db else dy.to(db))


def add(a, b):
    return _Add.apply(a, b)


class _Embedding(torch.autograd.Function):
    """VocabParallelEmbedding (mpu/layers.py:117-133) optionally fused with the position add + embedding
    dropout of GPT2ParallelTransformer.forward (mpu/sparse_transformer.py:522-524).
    Returns (out, absmax slot)."""

    @staticmethod
    def forward(ctx, ids, weight, vocab_start, pos_ids, pos_weight, drop, stream):
        mp = mp_world_size_or_1()
        slot = ops.new_absmax_slot(weight.device)
        if mp == 1:
            out = ops.embedding_fwd(ids, weight, vocab_start, pos_ids, pos_weight, dropout=drop, absmax_out=slot,
                                    out_f32=stream)
        else:
            word = ops.embedding_fwd(ids, weight, vocab_start)
            _mp_allreduce(word)
            if pos_weight is not None or drop is not None or stream:
                out = ops.embedding_fwd(None, None, 0, pos_ids, pos_weight, dropout=drop, absmax_out=slot, x_in=word,
                                        out_f32=stream)
            else:
                out = word
                ops.absmax(out, slot)
        ctx.save_for_backward(ids, pos_ids if pos_weight is not None else None)
        ctx.weight, ctx.pos_weight, ctx.vocab_start, ctx.drop = weight, pos_weight, vocab_start, drop
        ctx.mark_non_differentiable(slot)
        return out, slot

    @staticmethod
    def backward(ctx, dout, _):
        ids, pos_ids = ctx.saved_tensors
        w, pw = ctx.weight, ctx.pos_weight
        flush_weight_grads(final=True)            # a queued tied-logits gradient must land in the table's rows first
        dpos = grad_buffer(pw) if (pw is not None and pw.requires_grad) else None
        dtab = grad_buffer(w) if w.requires_grad else None
        for q in (pw if dpos is not None else None, w if dtab is not None else None):
            if q is not None and getattr(q, "_cogv_arena", None) is not None:
                q._cogv_arena[0].ensure_zeroed(q)         # the kernel adds its sums to the table: zero an untouched one first
        # SINGLE-STREAM ASSUMPTION: the segment-sum kernel read-modify-writes the table's gradient rows without atomics (that is
        # what makes it deterministic), so nothing else may touch dtab / dpos while it runs.  Backward runs on one stream (the
        # data-parallel exchange only reads a bucket after on_backward_done recorded its event); the tied logits GEMM's weight
        # gradient lands in the same rows EARLIER on that stream.  Hot ids (a pad token repeated ~1k times) are summed row by
        # row by one workgroup per 2048-column slab: correct, input-dependent tail of ~0.1 ms (embedding row).
        ops.embedding_bwd(dout, ids, dtab, ctx.vocab_start, pos_ids, dpos, dropout=ctx.drop)
        return None, None, None, None, None, None, None


def embedding(ids, weight, vocab_start, pos_ids=None, pos_weight=None, drop=None, stream=False):
    """stream=True: the result is the transformer's fp32 residual stream (GPT2ParallelTransformer.embed)."""
    out, slot = _Embedding.apply(ids, weight, vocab_start, pos_ids, pos_weight, drop, bool(stream))
    out._cogv_absmax = slot
    return out


class _VocabParallelCrossEntropy(torch.autograd.Function):
    """mpu/cross_entropy.py:25-104.  logits [..., V/p] in fp32 / fp16 / bf16 (read once, math in fp32);
    the three model-parallel all-reduces (MAX, SUM, SUM) act on [rows] vectors of shard statistics."""

    @staticmethod
    def forward(ctx, logits, target, inplace_backward):
        v = logits.shape[-1]
        l2 = logits.reshape(-1, v)
        if not l2.is_contiguous():
            l2 = l2.contiguous()
        t1 = target.reshape(-1).contiguous()
        mp, rank = mp_world_size_or_1(), mp_rank_or_0()
        vstart = rank * v
        rowmax, sumexp, pred, loss = ops.ce_fwd(l2, t1, vstart, want_loss=(mp == 1))
        if mp > 1:
            group = get_model_parallel_group()
            gmax = rowmax.clone()
            torch.distributed.all_reduce(gmax, op=torch.distributed.ReduceOp.MAX, group=group)
            packed = torch.stack((sumexp * torch.exp(rowmax - gmax), pred))
            torch.distributed.all_reduce(packed, group=group)
            gsum = packed[0].contiguous()
            loss = torch.log(gsum) + gmax - packed[1]
        else:
            gmax, gsum = rowmax, sumexp
        ctx.save_for_backward(l2, t1, gmax, gsum)
        ctx.vstart, ctx.shape, ctx.inplace = vstart, logits.shape, inplace_backward
        return loss.view(target.shape)

    @staticmethod
    def backward(ctx, g):
        l2, t1, gmax, gsum = ctx.saved_tensors
        gf = g.reshape(-1).float().contiguous()
        d = ops.ce_bwd(l2, t1, ctx.vstart, gmax, gsum, gf, out=l2 if ctx.inplace else None)
        return d.view(ctx.shape), None, None


def vocab_parallel_cross_entropy(vocab_parallel_logits, target, inplace_backward=False):
    return _VocabParallelCrossEntropy.apply(vocab_parallel_logits, target, inplace_backward)


class _Logits(torch.autograd.Function):
    """Tied output projection: logits = copy_to_mp(x) E^T (model/gpt2_modeling.py:115-118)."""

    @staticmethod
    def forward(ctx, x, emb_weight):
        x2 = x.reshape(-1, x.shape[-1])
        ctx.save_for_backward(x2, emb_weight)
        ctx.xshape = x.shape
        return ops.gemm(x2, emb_weight).view(*x.shape[:-1], emb_weight.shape[0])

    @staticmethod
    def backward(ctx, dl):
        x2, w = ctx.saved_tensors
        d2 = dl.reshape(-1, dl.shape[-1])
        dx = _mp_allreduce(ops.gemm(d2, w, trans_b=True)).view(ctx.xshape)
        if w.requires_grad:
            if mp_world_size_or_1() == 1 and WGRAD_QUEUE:
                # queued: its 256 x 256 tiles ride in the free slots of the layers' grouped weight-gradient launches
                # (_DeferredWeightGrads); d2 (the logits' gradient, 3 GB at 4B) stays alive until its last tile row is launched
                defer_weight_grad(d2, x2, w)
            else:
                ops.gemm(d2, x2, trans_a=True, trans_b=True, out=grad_buffer(w), accumulate=grad_accumulate(w))
        return dx, None


def tied_logits(x, emb_weight):
    return _Logits.apply(x, emb_weight)


# =============================================================================================== fused layer
class _LayerCtx:
    """Everything one layer's backward needs (activations + per-row LN statistics + dropout streams)."""
    __slots__ = ("x", "a", "qkv", "att", "lse", "ao", "y", "c", "u", "g", "mo", "st1", "st2", "st3", "st4",
                 "d_attn", "d_ao", "d_mo", "kbits")


def _layer_forward(layer, x, absmax_x, sep, drops, keep, kv_slot=None):
    """x [b,s,h] fp32 (the residual stream) -> (out fp32, absmax_out); `keep` is a _LayerCtx to fill (None: inference,
    nothing retained).  Everything that feeds a GEMM (a, qkv, att, ao, c, g, mo) is in the 16-bit storage type; the
    stream x -> y -> out stays fp32: LN1 / LN2 read it, LN3 / LN4 add their output to it in fp32.
    kv_slot (inference): the layer's key/value cache (mpu.transformer.KVCacheSlot): the new keys / values are appended
    and attention runs over the cache.
    Kernel chain (MP=1): LN1 | QKV GEMM+bias | attention | dense GEMM+bias+dropout+absmax | LN3+residual+absmax |
    LN2 | h->4h GEMM+bias+GeLU (+ stored gelu') | 4h->h GEMM+bias+dropout+absmax | LN4+residual+absmax."""
    att_m, mlp_m = layer.attention, layer.mlp
    b, s, h = x.shape
    rows = b * s
    mp, rank = mp_world_size_or_1(), mp_rank_or_0()
    eps = layer.input_layernorm.eps
    d_attn, d_ao, d_mo = drops
    dev = x.device
    npp = att_m.num_attention_heads_per_partition
    hp = npp * 64

    a, m1, r1 = ops.sandwich_ln_fwd(x, layer.input_layernorm.weight, layer.input_layernorm.bias, eps, absmax_x,
                                    save_stats=keep is not None)
    qkv = ops.gemm(a.view(rows, h), att_m.query_key_value.weight, bias=att_m.query_key_value.bias).view(b, s, 3 * hp)
    q = qkv[:, :, 0:hp].view(b, s, npp, 64)
    k = qkv[:, :, hp:2 * hp].view(b, s, npp, 64)
    v = qkv[:, :, 2 * hp:].view(b, s, npp, 64)
    kbits = None
    if kv_slot is None:
        # keep is not None: a backward pass will follow -- the forward kernel stores its dropout keep bits for it
        res = ops.attention_fwd(q, k, v, sep=sep, dropout=d_attn, keep_bits=keep is not None)
        att, lse, kbits = res[0], res[1], (res[2] if len(res) > 2 else None)
    elif s == 1 and getattr(kv_slot, "pos_index", None) is not None:
        # captured decode step (StaticKVSlot): keys split over workgroups, cache append fused, length read on the device
        att, lse = ops.attention_decode(qkv, kv_slot.cache, kv_slot.pos_index, npp).view(b, 1, npp, 64), None
        kv_slot.out = kv_slot.cache
    else:
        _, kc, vc = kv_slot.append(qkv[:, :, hp:2 * hp], qkv[:, :, 2 * hp:])
        kc, vc = kc.view(b, kc.shape[1], npp, 64), vc.view(b, vc.shape[1], npp, 64)
        table = getattr(kv_slot, "table", None)     # fixed-capacity cache of a captured decode step: gathered form
        if table is not None:
            att, lse = ops.attention_fwd(q, kc, vc, kv_index=table)
        else:
            att, lse = ops.attention_fwd(q, kc, vc, sep=sep, dropout=d_attn)

    slot_ao = ops.new_absmax_slot(dev)
    if mp == 1:
        ao = ops.gemm(att.view(rows, hp), att_m.dense.weight, bias=att_m.dense.bias, dropout=d_ao, absmax=slot_ao)
    else:
        # model parallel: hidden dropout is linear per element and its mask depends only on (seed, stream, element index)
        # of the DEFAULT state, which is the same on every rank of a model-parallel group -- so bias (once) + dropout
        # ride in each rank's GEMM epilogue and the all-reduce sums already-dropped partials:
        # mask*(A + B + bias) = mask*(A + bias) + mask*B.  What must follow the reduce is only the abs-max of the sum
        # (a read-only pass; the reference's order is dropout(all_reduce(.) + bias), mpu/layers.py:312-326)
        ao = _row_parallel_chunked(att.view(rows, hp), att_m.dense.weight, att_m.dense.bias if rank == 0 else None, d_ao, slot_ao)
    slot_y = ops.new_absmax_slot(dev)
    y, m3, r3 = ops.sandwich_ln_fwd(ao.view(b, s, h), layer.third_layernorm.weight, layer.third_layernorm.bias, eps,
                                    slot_ao, residual=x, absmax_out=slot_y, save_stats=keep is not None)
    c, m2, r2 = ops.sandwich_ln_fwd(y, layer.post_attention_layernorm.weight, layer.post_attention_layernorm.bias,
                                    eps, slot_y, save_stats=keep is not None)
    f4 = mlp_m.dense_h_to_4h.weight.shape[0]
    u = torch.empty((rows, f4), dtype=a.dtype, device=dev) if keep is not None else None
    # u receives gelu'(pre-activation): backward multiplies by it instead of re-evaluating the sigmoid
    g = ops.gemm(c.view(rows, h), mlp_m.dense_h_to_4h.weight, bias=mlp_m.dense_h_to_4h.bias, gelu=True, gelu_daux=u)
    slot_mo = ops.new_absmax_slot(dev)
    if mp == 1:
        mo = ops.gemm(g, mlp_m.dense_4h_to_h.weight, bias=mlp_m.dense_4h_to_h.bias, dropout=d_mo, absmax=slot_mo)
    else:
        mo = _row_parallel_chunked(g, mlp_m.dense_4h_to_h.weight, mlp_m.dense_4h_to_h.bias if rank == 0 else None, d_mo, slot_mo)
    slot_out = ops.new_absmax_slot(dev)
    out, m4, r4 = ops.sandwich_ln_fwd(mo.view(b, s, h), layer.fourth_layernorm.weight, layer.fourth_layernorm.bias,
                                      eps, slot_mo, residual=y, absmax_out=slot_out, save_stats=keep is not None)
    if keep is not None:
        keep.x, keep.a, keep.qkv, keep.att, keep.lse, keep.ao, keep.y, keep.c = x, a, qkv, att, lse, ao, y, c
        keep.u, keep.g, keep.mo = u, g, mo
        keep.st1, keep.st2, keep.st3, keep.st4 = (m1, r1), (m2, r2), (m3, r3), (m4, r4)
        keep.d_attn, keep.d_ao, keep.d_mo, keep.kbits = d_attn, d_ao, d_mo, kbits
    return out, slot_out


LN_BWD_PAIR = os.environ.get("COGV_LN_BWD_PAIR", "1") != "0"


def _layer_backward(layer, kp, dout, sep):
    """dout [b,s,h] fp32 (gradient of the residual stream) -> dx fp32; parameter gradients are accumulated into
    param.grad (16-bit)."""
    att_m, mlp_m = layer.attention, layer.mlp
    b, s, h = kp.x.shape
    rows = b * s
    npp = att_m.num_attention_heads_per_partition
    hp = npp * 64
    G = grad_buffer
    ln1, ln2, ln3, ln4 = (layer.input_layernorm, layer.post_attention_layernorm, layer.third_layernorm,
                          layer.fourth_layernorm)
    W1, b1, W2, b2 = mlp_m.dense_h_to_4h.weight, mlp_m.dense_h_to_4h.bias, mlp_m.dense_4h_to_h.weight, mlp_m.dense_4h_to_h.bias
    Wq, bq, Wo, bo = att_m.query_key_value.weight, att_m.query_key_value.bias, att_m.dense.weight, att_m.dense.bias
    dout = dout if dout.is_contiguous() else dout.contiguous()

    # out = y + LN4(mo):  d_mo = mask(LN4'(dout)); bias grad of 4h->h = column sums of d_mo
    # (kp.mo / kp.ao came out of the GEMM dropout epilogue: their dropped elements are -0.0, "marked zeros", and the two
    #  dropout-side LayerNorm backwards read the mask from them instead of re-hashing it.  Under model parallelism the tensors
    #  went through an all-reduce in between: there the mask is regenerated, as before.  COGV_LN_BWD_MARKED=0: always regenerate)
    marked = mp_world_size_or_1() == 1 and os.environ.get("COGV_LN_BWD_MARKED", "1") != "0"
    d_mo = ops.sandwich_ln_bwd(dout, kp.mo, ln4.weight, *kp.st4, dropout=kp.d_mo, dgamma=G(ln4.weight), marked=marked,
                               dbeta=G(ln4.bias), colsum=G(b2), accumulate=grad_accumulate(ln4.weight, ln4.bias, b2)).view(rows, h)
    # The four weight gradients dW = dY^T X are deferred to a grouped launch (flush_weight_grads): together their
    # 256x256 tiles fill whole rounds of the 256 CUs (each alone needs split-K slabs or idles half a round).
    # Model parallel: the two column-parallel dgrads (dc, da) are partial sums that must be all-reduced before the
    # LayerNorm backward that consumes them; the weight gradients of the GEMMs already back-propagated are independent
    # of that exchange, so they are launched while it runs (the reference's autograd serialises them, mpu/mappings.py:
    # 79-93 inside F.linear's backward).  Without model parallelism they stay deferred to the grouped launch.
    mp = mp_world_size_or_1()
    wgrads = []                                   # model parallel only: launched behind each exchange (below)

    def _wg(dy_, x_, w_):
        if mp == 1:
            defer_weight_grad(dy_, x_, w_, owner=layer)      # queued: the layer-group flushes launch whole rounds of tiles
        else:
            wgrads.append((dy_, x_, w_))
    du = ops.gemm(d_mo, W2, trans_b=True, mul_aux=kp.u, colsum_out=G(b1),       # dgrad x stored gelu' + bias grad of h->4h
                  colsum_accumulate=grad_accumulate(b1))
    _wg(d_mo, kp.g, W2)
    dc = ops.gemm(du, W1, trans_b=True)
    _wg(du, kp.c.view(rows, h), W1)
    if mp > 1:
        work = _mp_allreduce_start(dc)
        with _room_for_collective():
            _launch_weight_grads(wgrads)                                         # overlaps the exchange of dc
        del wgrads[:]
        _mp_allreduce_finish(work)
    # y feeds LN2 and the second residual:  dy = dout + LN2'(dc);  y = x + LN3(ao):  d_ao = mask(LN3'(dy)).  The two are
    # neighbours: one pass over the rows (dy is written for LN1' below but not read back) when the mask can be read from ao's
    # marked zeros -- or there is no dropout -- and the rows are wide; COGV_LN_BWD_PAIR=0 keeps the two launches
    p_ao = 0.0 if kp.d_ao is None else float(kp.d_ao[0])
    if LN_BWD_PAIR and ops.ln_bwd_pair_supported(h) and (marked or p_ao == 0.0):
        dy, d_ao = ops.sandwich_ln_bwd_pair(dc.view(b, s, h), kp.y, ln2.weight, *kp.st2, dout, kp.ao, ln3.weight, *kp.st3,
                                            dropout_p=p_ao, dgamma2=G(ln2.weight), dbeta2=G(ln2.bias), dgamma3=G(ln3.weight),
                                            dbeta3=G(ln3.bias), colsum=G(bo),
                                            accumulate=grad_accumulate(ln2.weight, ln2.bias, ln3.weight, ln3.bias, bo))
        d_ao = d_ao.view(rows, h)
    else:
        dy = ops.sandwich_ln_bwd(dc.view(b, s, h), kp.y, ln2.weight, *kp.st2, add_in=dout, dgamma=G(ln2.weight),
                                 dbeta=G(ln2.bias), accumulate=grad_accumulate(ln2.weight, ln2.bias))
        d_ao = ops.sandwich_ln_bwd(dy, kp.ao, ln3.weight, *kp.st3, dropout=kp.d_ao, dgamma=G(ln3.weight), marked=marked,
                                   dbeta=G(ln3.bias), colsum=G(bo), accumulate=grad_accumulate(ln3.weight, ln3.bias, bo)).view(rows, h)
    d_att = ops.gemm(d_ao, Wo, trans_b=True).view(b, s, npp, 64)
    _wg(d_ao, kp.att.view(rows, hp), Wo)
    qkv = kp.qkv
    q = qkv[:, :, 0:hp].view(b, s, npp, 64)
    k = qkv[:, :, hp:2 * hp].view(b, s, npp, 64)
    v = qkv[:, :, 2 * hp:].view(b, s, npp, 64)
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(d_att, q, k, v, kp.att, kp.lse, sep=sep, dropout=kp.d_attn,
                      dq=dqkv[:, :, 0:hp].view(b, s, npp, 64), dk=dqkv[:, :, hp:2 * hp].view(b, s, npp, 64),
                      dv=dqkv[:, :, 2 * hp:].view(b, s, npp, 64), colsum_out=G(bq), colsum_accumulate=grad_accumulate(bq),
                      keep_bits=getattr(kp, "kbits", None))
    dqkv2 = dqkv.view(rows, 3 * hp)
    da = ops.gemm(dqkv2, Wq, trans_b=True)
    _wg(dqkv2, kp.a.view(rows, h), Wq)
    if mp > 1:
        work = _mp_allreduce_start(da)
        with _room_for_collective():
            _launch_weight_grads(wgrads)                                         # overlaps the exchange of da
        _mp_allreduce_finish(work)
    dx = ops.sandwich_ln_bwd(da.view(b, s, h), kp.x, ln1.weight, *kp.st1, add_in=dy, dgamma=G(ln1.weight),
                             dbeta=G(ln1.bias), accumulate=grad_accumulate(ln1.weight, ln1.bias))
    return dx


class _WgradEntry:
    """One deferred weight gradient dW (+)= dY^T X, launched in chunks of whole 256-row tile rows of dW."""
    __slots__ = ("dy", "x", "w", "owner", "t0", "trows", "tiles_n", "acc")

    def __init__(self, dy, x, w, owner):
        self.dy, self.x, self.w, self.owner = dy, x, w, owner
        self.t0, self.acc = 0, None                                   # next tile row to launch; accumulate flag (first chunk decides)
        self.trows, self.tiles_n = (w.shape[0] + 255) // 256, (w.shape[1] + 255) // 256

    def tiles_left(self):
        return (self.trows - self.t0) * self.tiles_n


def plan_wgrad_chunks(entries, budget):
    """Which tile rows of which pending problems go into the next launch.  entries: [(t0, trows, tiles_n, out_rows)] in queue
    order; budget: tiles the launch may hold (None: everything).  Returns [(index, t0, t1)], oldest problem first, whole tile
    rows only, at most `budget` tiles in all.  A problem whose last tile row is partial (out_rows % 256 != 0) is never cut so
    that this row would be left on its own: the grouped kernel takes problems of at least 256 rows."""
    out, left = [], budget
    for i, (t0, trows, tiles_n, out_rows) in enumerate(entries):
        if t0 >= trows:
            continue
        n = trows - t0
        if left is not None:
            n = min(n, left // tiles_n)
            if n <= 0:
                continue
        t1 = t0 + n
        if t1 == trows - 1 and out_rows % 256 != 0:                   # would strand the partial last row
            t1 -= 1
        if t1 <= t0 or (t1 - t0 == 1 and t1 == trows and out_rows % 256 != 0 and trows > 1):
            continue
        out.append((i, t0, t1))
        if left is not None:
            left -= (t1 - t0) * tiles_n
    return out


class _DeferredWeightGrads:
    """Weight-gradient GEMMs waiting for a grouped launch: a QUEUE of problems cut into launches of whole rounds.

    A launch of the persistent 256 x 256-tile kernel costs ceil(tiles / 256) rounds of the 256 CUs.  One 4B layer's four weight
    gradients are 1200 tiles = 4.69 rounds (5 paid), the tied-logits gradient 2280 tiles = 8.9 (9 paid): 249 rounds per step for
    233.9 rounds of work.  Round 5: every flush (one per layer group, as before) launches a MULTIPLE of 256 tiles taken from the
    head of the queue -- at most the rounds the group itself would have paid (1280 tiles at 4B, so that a launch stays as short
    as it was: longer uninterrupted GEMMs clock lower, see below) -- and what does not fit waits for the next flush; problems are
    cut along tile rows of dW (a column slice of dY, a row slice of dW: plain pointer offsets, every tile is computed by the same
    kernel in the same order -> bit-identical results).  The tied-logits gradient (functional._Logits.backward) joins the queue
    first and fills the 80 free tile slots of the first 29 layer launches (a layer's own problems are taken before it); after that
    the launches alternate between 5 and 4 full rounds, a layer's last tile rows riding in the next layer's launch.  234 rounds
    instead of 249 at 4B: -9.5 ms per step.  The flush of layer index 0 (the last one backward visits), the embedding's backward
    and the end of the autograd pass launch whatever is left.
    The data-parallel callback of a layer runs once all of ITS problems have been launched, in backward order.
    COGV_WGRAD_QUEUE=0: every flush launches everything it has (the round-4 behaviour).

    Measured at 4B (round 3): one layer per launch at 1328 TFLOP/s (generation-4 GEMM); four layers per launch (18.75 rounds,
    1.3 % tail) ran at 1267 TFLOP/s -- a 13 ms uninterrupted GEMM sits at the sustained power limit, while 3 ms launches
    separated by the lighter LN / attention kernels clock higher.  Hence one layer per flush whenever a layer fills the chip
    (wgrad_group_layers; COGV_WGRAD_GROUP_LAYERS overrides it for experiments)."""
    __slots__ = ("entries", "callbacks", "new_tiles", "root", "hooked")

    def __init__(self):
        # root: graph-task id of the (outermost) backward pass the pending problems belong to; hooked: ids of the passes -- that
        # one and the passes nested in it (mpu.checkpoint's recompute) -- whose end already carries the flush callback
        self.entries, self.callbacks, self.new_tiles, self.root, self.hooked = [], [], 0, -1, set()

    def add(self, dy, x, w, owner=None):
        e = _WgradEntry(dy, x, w, owner)
        self.entries.append(e)
        if owner is not None:                     # the flushed layer group's own tiles set the size of its launch
            self.new_tiles += e.trows * e.tiles_n

    # kept for callers that read the list of pending problems
    @property
    def problems(self):
        return self.entries


import os as _os
WGRAD_GROUP_LAYERS = int(_os.environ.get("COGV_WGRAD_GROUP_LAYERS", "0"))       # 0: by tile count (wgrad_group_layers)
WGRAD_QUEUE = _os.environ.get("COGV_WGRAD_QUEUE", "1") != "0"
WGRAD_ROUND_TILES = 256                                                          # tile slots of one round: one 256 x 256 tile per CU
_WGRADS = _DeferredWeightGrads()
_WGRAD_GROUP_CACHE = {}


def wgrad_group_layers(layer):
    """Layers whose weight gradients share one grouped launch.  A layer's four problems are sum(ceil(out/256) *
    ceil(in/256)) tiles of 256 x 256: 1200 at the 4B width (4.7 rounds of the 256 CUs: one layer per launch measured
    best, see _DeferredWeightGrads), 192 at the 336M width -- less than one round, which forces split-K slabs and a
    reduce pass; four layers together are 768 tiles = 3.0 rounds without either (measured at 336M: weight-gradient
    launches 1274 -> 1512 TFLOP/s, step 95.8 -> 94.4 ms).  Rule: the smallest group of 1, 2 or 4 layers (16 problems per
    launch is the kernel's limit) that fills at least 2.5 rounds."""
    if WGRAD_GROUP_LAYERS > 0:
        return WGRAD_GROUP_LAYERS
    w = layer.mlp.dense_h_to_4h.weight
    key = (w.shape, layer.attention.query_key_value.weight.shape)
    g = _WGRAD_GROUP_CACHE.get(key)
    if g is None:
        tiles = 0
        for lin in (layer.mlp.dense_h_to_4h, layer.mlp.dense_4h_to_h, layer.attention.dense, layer.attention.query_key_value):
            o, i = lin.weight.shape
            tiles += ((o + 255) // 256) * ((i + 255) // 256)
        g = 1
        while g < 4 and g * tiles < 640:
            g *= 2
        _WGRAD_GROUP_CACHE[key] = g
    return g


def _launch_weight_grads(probs):
    """probs: [(dY, X, weight parameter)] -> grouped dW (+)= dY^T X launches of at most 16 problems; whether a problem
    accumulates or overwrites is decided here, per parameter (grad_accumulate)."""
    for i in range(0, len(probs), 16):
        ops.gemm_grouped([(a, b, grad_buffer(w), grad_accumulate(w)) for a, b, w in probs[i:i + 16]], trans_a=True, trans_b=True)


def _launch_wgrad_chunks(chunks):
    """chunks: [(entry, t0, t1)] -> grouped launches of at most 16 sub-problems (tile rows [t0, t1) of each entry's dW)."""
    subs = []
    for e, t0, t1 in chunks:
        if e.acc is None:
            e.acc = grad_accumulate(e.w)          # decided once per problem: its chunks write disjoint rows
        r0, r1 = 256 * t0, min(256 * t1, e.w.shape[0])
        g = grad_buffer(e.w)
        whole = r0 == 0 and r1 == e.w.shape[0]
        subs.append((e.dy if whole else e.dy[:, r0:r1], e.x, g if whole else g[r0:r1], e.acc))
        e.t0 = t1
    for i in range(0, len(subs), 16):
        ops.gemm_grouped(subs[i:i + 16], trans_a=True, trans_b=True)


def flush_weight_grads(final=True):
    """Launch pending weight gradients.  final=True: all of them (the last layer of a backward pass, the embedding's backward,
    the end of the autograd pass, tests); final=False (a layer group finished, more follow): whole rounds only -- see
    _DeferredWeightGrads."""
    q = _WGRADS
    if not q.entries and not q.callbacks:
        return
    budget = None
    if not final and WGRAD_QUEUE:
        pending = sum(e.tiles_left() for e in q.entries)
        cap = -(-q.new_tiles // WGRAD_ROUND_TILES) * WGRAD_ROUND_TILES
        budget = min(cap, (pending // WGRAD_ROUND_TILES) * WGRAD_ROUND_TILES)
    q.new_tiles = 0
    if budget is None or budget > 0:
        # the layers' own problems first (oldest first: their activations are released and their data-parallel buckets start as
        # early as possible), then problems without an owner (the tied-logits gradient) as filler for the remaining tile slots
        order = [e for e in q.entries if e.owner is not None] + [e for e in q.entries if e.owner is None]
        plan = plan_wgrad_chunks([(e.t0, e.trows, e.tiles_n, e.w.shape[0]) for e in order], budget)
        _launch_wgrad_chunks([(order[i], t0, t1) for i, t0, t1 in plan])
    q.entries = [e for e in q.entries if e.t0 < e.trows]
    # data-parallel callbacks, in backward order, up to the first layer that still has a problem in the queue
    busy = {id(e.owner) for e in q.entries if e.owner is not None}
    while q.callbacks and id(q.callbacks[0][1]) not in busy:
        cb, layer = q.callbacks.pop(0)
        cb(layer)


def _flush_at_end_of_backward():
    flush_weight_grads(final=True)


# graph-task ids of the backward passes that are, right now, running a NESTED backward pass on this thread (mpu.checkpoint's
# recompute: CheckpointFunction.backward calls torch.autograd.backward inside the outer pass).  The problems a nested pass defers
# belong to the outermost pass: its queue must survive them.
_NESTED_OUTER = []


@contextlib.contextmanager
def nested_backward():
    """Around a torch.autograd.backward() issued from inside a backward pass (mpu/random.py CheckpointFunction.backward)."""
    _NESTED_OUTER.append(torch._C._current_graph_task_id())
    try:
        yield
    finally:
        _NESTED_OUTER.pop()


def defer_weight_grad(dy, x, w, owner=None):
    """Queue dW (+)= dY^T X for the next grouped launches.  Called inside an autograd backward pass: a callback at the end of
    that pass launches whatever no layer flush has taken (nothing in a GPT2Model step: layer index 0 flushes everything)."""
    q = _WGRADS
    task = torch._C._current_graph_task_id()      # the autograd pass this call belongs to (-1: none)
    root = _NESTED_OUTER[0] if _NESTED_OUTER else task
    if root >= 0 and q.root != root:
        # the first problem of a new backward pass: drop whatever a pass that died half way (an exception between two flushes:
        # the engine runs no final callbacks then) left behind -- stale tensors, gradients nobody consumed.  A pass NESTED in the
        # current one is not a new pass: round 5's first form compared the nested pass' own id and threw the outer pass'
        # pending problems away (the tied-logits gradient under mpu.checkpoint'ed layers)
        q.entries, q.callbacks, q.new_tiles = [], [], 0
        q.root, q.hooked = root, set()
    if task >= 0 and task not in q.hooked:
        torch.autograd.Variable._execution_engine.queue_callback(_flush_at_end_of_backward)      # at the end of THIS pass
        q.hooked.add(task)
    q.add(dy, x, w, owner)                        # (outside a backward pass -- a test driving _layer_backward by hand -- the caller flushes)


class _TransformerLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, absmax_x, layer, sep, drops, recompute, on_backward_done):
        xc = x if x.is_contiguous() else x.contiguous()
        keep = None if recompute else _LayerCtx()
        out, slot = _layer_forward(layer, xc, absmax_x, sep, drops, keep)
        ctx.layer, ctx.sep, ctx.drops, ctx.recompute, ctx.keep = layer, sep, drops, recompute, keep
        ctx.done_cb = on_backward_done
        if recompute:
            # (with mpu.partition_activations_in_checkpoint(True): this model-parallel rank's slice of the layer input)
            piece, ctx.x_shape = mpu_random.partition_activation(xc)
            ctx.save_for_backward(piece, absmax_x)
        ctx.mark_non_differentiable(slot)
        return out, slot

    @staticmethod
    def backward(ctx, dout, _):
        keep = ctx.keep
        if ctx.recompute:
            # activation checkpointing (mpu/random.py:273-372): only the layer input was kept; the dropout
            # streams are part of `drops`, so the recomputed forward replays identical masks
            xc, absmax_x = ctx.saved_tensors
            xc = mpu_random.gather_activation(xc, ctx.x_shape)
            keep = _LayerCtx()
            _layer_forward(ctx.layer, xc, absmax_x, ctx.sep, ctx.drops, keep)
        dx = _layer_backward(ctx.layer, keep, dout, ctx.sep)
        ctx.keep = None
        if ctx.done_cb is not None:
            _WGRADS.callbacks.append((ctx.done_cb, ctx.layer))
        idx = getattr(ctx.layer, "_cogv_index", 0)
        if idx % wgrad_group_layers(ctx.layer) == 0:
            flush_weight_grads(final=(idx == 0))
        return dx, None, None, None, None, None, None


# decode_chain: combine the decode attention's key splits in the prologue of the attention-output GEMV (one launch per layer
# less).  Measured at 4B, 1024-position memory (tools/mb_decode.py, same call): EAGER step 4.16 -> 3.73 ms, CAPTURED graph
# 3.40 -> 3.62 ms (inside a graph a launch costs less than 320 workgroups recombining the partials) -- so the default is
# "fused unless the stream is being captured"; COGV_DECODE_FUSE_COMBINE=0 / 1 forces either form.
_DECODE_FUSE_ENV = _os.environ.get("COGV_DECODE_FUSE_COMBINE")


_DECODE_CHAIN_MAX_ROWS = min(8, max(0, int(_os.environ.get("COGV_DECODE_CHAIN_MAX_ROWS", "4"))))


def _decode_fuse_combine():
    if _DECODE_FUSE_ENV is not None:
        return _DECODE_FUSE_ENV != "0"
    return not torch.cuda.is_current_stream_capturing()


def decode_chain_supported(tr, batch):
    """The fused decode chain (decode_chain) covers the dense, single-partition model in a 16-bit type at one token per
    row: what a captured decode step runs."""
    h = tr.layers[0].input_layernorm.weight.shape[0]
    # COGV_DECODE_CHAIN_MAX_ROWS (default 4): above it the step runs layer by layer (Sandwich-LN launches + plain matrix-core
    # products).  At 5 .. 8 rows the fused LayerNorm prologue is what bounds the chain (every workgroup re-derives x_in for all
    # rows); measured at batch 8, captured 4B step: 6.34 ms per step with the chain, 5.07 layer by layer
    # (profiles/r05_decode_chain_rows_ab.log).
    return (mp_world_size_or_1() == 1 and batch <= _DECODE_CHAIN_MAX_ROWS and h % 512 == 0 and h <= 4096
            and tr.layers[0].input_layernorm.weight.dtype in (torch.float16, torch.bfloat16))


def decode_chain(tr, h0, absmax0, slots, emb_weight):
    """One decode step through all layers with FIVE launches per layer (round 3: the split-combine of the decode attention
    rides in the dense GEMV's prologue): QKV GEMV with [previous layer's
    LN4 + residual, LN1] as prologue | decode attention, key splits (cache append fused) | dense GEMV with the combine as prologue | h->4h GEMV with [LN3 +
    residual, LN2] as prologue and GeLU epilogue | 4h->h GEMV; the last LN4 + residual and the final LayerNorm are the
    prologue of the tied-logits GEMV.  Same arithmetic and rounding points as the layer-by-layer path
    (mpu/sparse_transformer.py:314-342, 612; model/gpt2_modeling.py:115-118).  h0 [b, 1, h]; slots: StaticKVSlot per
    layer.  Returns logits [b, 1, V]."""
    b, s, h = h0.shape
    assert s == 1
    dev = h0.device
    z, z_absmax, post, res = h0.view(b, h), absmax0, None, None
    for layer, slot in zip(tr.layers, slots):
        att_m, mlp_m = layer.attention, layer.mlp
        eps = layer.input_layernorm.eps
        npp = att_m.num_attention_heads_per_partition
        hp = npp * 64
        qkv, x = ops.gemv_ln(z, att_m.query_key_value.weight, att_m.query_key_value.bias, layer.input_layernorm.weight,
                             layer.input_layernorm.bias, eps, z_absmax, post, res, want_t=post is not None)
        if x is None:
            x = z
        # (round 4) the Sandwich scale of the two branch outputs, max|ao| and max|mo|, is taken by the CONSUMING launch's LayerNorm
        # prologue from the vector it holds anyway (z_absmax=None) instead of being published by the producing matrix-vector
        # launch through an atomic: same value, and the producers' workgroups -- which all finish together -- end without a
        # burst of same-address atomics
        if hp % 512 == 0 and _decode_fuse_combine():
            # the key splits' partials are combined in the prologue of the attention-output GEMV (one launch less)
            parts = ops.attention_decode(qkv.view(b, 1, 3 * hp), slot.cache, slot.pos_index, npp, combine=False)
            ao = ops.gemv_attn(parts, b, npp, slot.cache.shape[1], att_m.dense.weight, bias=att_m.dense.bias)
        else:
            att = ops.attention_decode(qkv.view(b, 1, 3 * hp), slot.cache, slot.pos_index, npp)
            ao = ops.gemm(att.view(b, hp), att_m.dense.weight, bias=att_m.dense.bias)
        slot.out = slot.cache
        g, y = ops.gemv_ln(ao, mlp_m.dense_h_to_4h.weight, mlp_m.dense_h_to_4h.bias, layer.post_attention_layernorm.weight,
                           layer.post_attention_layernorm.bias, eps, None,
                           (layer.third_layernorm.weight, layer.third_layernorm.bias), x, want_t=True, gelu=True)
        mo = ops.gemm(g, mlp_m.dense_4h_to_h.weight, bias=mlp_m.dense_4h_to_h.bias)
        z, z_absmax, post, res = mo, None, (layer.fourth_layernorm.weight, layer.fourth_layernorm.bias), y
    fl = tr.final_layernorm
    logits, _ = ops.gemv_ln(z, emb_weight, None, fl.weight, fl.bias, fl.eps, z_absmax, post, res)
    return logits.view(b, 1, emb_weight.shape[0])


def transformer_layer_kv(layer, x, absmax_x, sep, kv_slot):
    """The fused layer chain for incremental decoding (no gradient): 9 kernels + the cache append instead of the ~20 of
    the op-by-op composition."""
    xc = x if x.is_contiguous() else x.contiguous()
    if absmax_x is None:
        absmax_x = ops.absmax(xc)
    out, slot = _layer_forward(layer, xc, absmax_x, sep, (None, None, None), None, kv_slot=kv_slot)
    out._cogv_absmax = slot
    return out


def transformer_layer(layer, x, absmax_x, sep, training, recompute=False, on_backward_done=None):
    """Fused GPT2ParallelTransformerLayer.forward (mpu/sparse_transformer.py:314-342), dense attention."""
    p_attn = layer.attention.attention_dropout.p
    p_out = layer.attention.output_dropout.p
    p_mlp = layer.mlp.dropout.p
    drops = (_drop(p_attn, training, attention=True), _drop(p_out, training), _drop(p_mlp, training))
    if absmax_x is None:
        absmax_x = ops.absmax(x if x.is_contiguous() else x.contiguous())
    if torch.is_grad_enabled() and x.requires_grad:
        out, slot = _TransformerLayer.apply(x, absmax_x, layer, sep, drops, recompute, on_backward_done)
    else:
        out, slot = _layer_forward(layer, x if x.is_contiguous() else x.contiguous(), absmax_x, sep, drops, None)
    out._cogv_absmax = slot
    return out
