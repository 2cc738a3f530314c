"""CPU oracle for the CogView hot path -- TEST INFRASTRUCTURE ONLY.

A plain restatement (torch CPU tensors in float32/float64, autograd for the backward pass) of the reference
algorithm, function by function, each citing the reference file:line it follows.  Nothing under
`cogview_amd/` imports this module; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
and only as the checker / the timed CPU baseline -- never as the product path.

Pinning: the reference ships no golden vectors or tests for this path, so the oracle
is pinned against the reference ITSELF: oracle/gen_golden.py imports /root/reference (with shims for the
un-vendored apex / deepspeed / torch._six imports), runs it on seeded inputs and stores inputs + outputs
under tests/golden/*.npz; tests/test_oracle_golden.py checks every function here against those vectors.
Third-party arithmetic not present in /root/reference (apex FusedLayerNorm == torch.nn.LayerNorm,
apex FusedAdam(adam_w_mode=True) == decoupled-weight-decay Adam) is restated from its published definition.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------------- primitives
def sandwich_layernorm(x, weight, bias, eps=1e-5):
    """mpu/sparse_transformer.py:40-44: FusedLayerNorm(x / (x.abs().max().detach() / 8))."""
    c = x.detach().abs().max() / 8
    return F.layer_norm(x / c, (x.shape[-1],), weight, bias, eps)


def gelu(x):
    """mpu/sparse_transformer.py:172-176 (OpenAI tanh form)."""
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * x * (1.0 + 0.044715 * x * x)))


def linear(x, w, b=None):
    """F.linear as used at mpu/layers.py:243,319 and model/gpt2_modeling.py:117 (weight is [out, in])."""
    y = x @ w.t()
    return y if b is None else y + b


def build_mask(s_q, s_k, sep=0, dtype=torch.float32):
    """Left-to-right mask: pretrain_gpt2.py:219-221 (sep=0) / mpu/sparse_transformer.py:482-488 (sep form)."""
    m = torch.ones(s_q, s_k, dtype=dtype)
    m[:, -s_q:] = torch.tril(m[:, -s_q:])
    if sep > 0:
        m[:, :sep + (s_k - s_q)] = 1
    return m.view(1, 1, s_q, s_k)


def standard_attention(q, k, v, mask, drop_mask=None):
    """mpu/sparse_transformer.py:652-673.  q,k,v: [b, np, s, hn].  drop_mask (optional) is the already
    scaled keep mask (0 or 1/(1-p)) applied to the probabilities (line 667-669)."""
    scores = torch.matmul(q / math.sqrt(q.shape[-1]), k.transpose(-1, -2))
    scores = scores * mask - 10000.0 * (1.0 - mask)
    probs = torch.softmax(scores, dim=-1)
    if drop_mask is not None:
        probs = probs * drop_mask
    return torch.matmul(probs, v)


def sparse_rmask(s, w, times):
    """Mask used to pick the visible pivots in sparse training (mpu/sparse_transformer.py:491-496): position q may
    use key j as a PIVOT iff j lies before the window of q's block, i.e. j < (q // w - times + 1) * w
    (keys inside the window are reached through the window branch instead)."""
    qi = torch.arange(s).unsqueeze(1)
    kj = torch.arange(s).unsqueeze(0)
    return (kj < (qi // w - times + 1) * w).to(torch.float32)


def sparse_attention(q, k, v, pivot_idx, pivot_attention_mask, w=128, times=6):
    """CogView sparse attention, training form (mpu/sparse_transformer.py:675-725), restated per query:
        keys of query i  =  the n_piv pivot keys (gathered; masked; every pivot score gets + log(s // n_piv))
                          U the w*times window slots ending with i's block: slot c of block g is key
                            (g - times + 1) * w + c (zero padding in front of the sequence).
        window slot c of the query with in-block offset r is visible iff c <= r + w*(times-1); masked slots score
        -10000; for the first times-1 blocks the padding slots take ANOTHER -10000 (:710-711 subtracts it from the
        already masked value, so those slots score -20000).
      joint softmax over the n_piv + w*times scores, context = probs @ [pivot values ; window values].
    q, k, v [b, nh, s, hn] with s % w == 0; pivot_idx [b, n_piv]; pivot_attention_mask [b, s, n_piv]."""
    b, nh, s, hn = q.shape
    n_piv = pivot_idx.shape[1]
    scale = 1.0 / math.sqrt(hn)
    idx = pivot_idx.view(b, 1, n_piv, 1).expand(b, nh, n_piv, hn)
    pk, pv = torch.gather(k, 2, idx), torch.gather(v, 2, idx)
    pm = pivot_attention_mask.unsqueeze(1)
    sp = torch.einsum("bhqd,bhpd->bhqp", q, pk) * (pm * scale) - 10000.0 * (1.0 - pm) + math.log(s // n_piv)
    # window: slot c of query i is key (i // w - times + 1) * w + c; negative keys are padding (zero vectors)
    qi = torch.arange(s)
    c = torch.arange(w * times)
    key = ((qi // w - times + 1) * w).unsqueeze(1) + c.unsqueeze(0)             # [s, w*times]
    pad = key < 0
    kw = k[:, :, key.clamp(min=0)] * (~pad).view(1, 1, s, w * times, 1)         # [b, nh, s, w*times, hn]
    vw = v[:, :, key.clamp(min=0)] * (~pad).view(1, 1, s, w * times, 1)
    vis = (c.unsqueeze(0) <= (qi % w).unsqueeze(1) + w * (times - 1)).to(q.dtype)
    sw = torch.einsum("bhqd,bhqcd->bhqc", q, kw) * (vis * scale) - 10000.0 * (1.0 - vis)
    sw = sw - 10000.0 * pad.to(q.dtype)
    probs = torch.softmax(torch.cat((sp, sw), dim=-1), dim=-1)
    return torch.einsum("bhqp,bhpd->bhqd", probs[..., :n_piv], pv) + torch.einsum("bhqc,bhqcd->bhqd", probs[..., n_piv:], vw)


def sparse_attention_inference(q, k, v, pivot_and_window_idx):
    """mpu/sparse_transformer.py:727-750: the sq queries are the last sq keys; they attend the gathered
    (pivot + window) keys, causally among themselves (the last sq gathered keys are the queries' own positions)."""
    b, nh, sq, hn = q.shape
    n = pivot_and_window_idx.shape[1]
    idx = pivot_and_window_idx.view(b, 1, n, 1).expand(b, nh, n, hn)
    gk, gv = torch.gather(k, 2, idx), torch.gather(v, 2, idx)
    sc = torch.einsum("bhqd,bhpd->bhqp", q / math.sqrt(hn), gk)
    if sq > 1:
        causal = torch.triu(torch.full((sq, sq), -10000.0, dtype=q.dtype), diagonal=1)
        sc = torch.cat((sc[..., :-sq], sc[..., -sq:] + causal), dim=-1)
    return torch.einsum("bhqp,bhpd->bhqd", torch.softmax(sc, dim=-1), gv)


def self_attention(x, ltor_mask, p, prefix, n_heads, attn_drop_mask=None, out_drop_mask=None, mem=None):
    """GPT2ParallelSelfAttention.forward, mpu/sparse_transformer.py:123-169 (model-parallel size 1)."""
    b, s, h = x.shape
    src = x if mem is None else torch.cat((mem, x), 1)
    mixed = linear(src, p[prefix + "query_key_value.weight"], p[prefix + "query_key_value.bias"])
    q, k, v = torch.split(mixed, h, dim=-1)
    if mem is not None:
        q = q[:, -s:]
    hn = h // n_heads
    tr = lambda t: t.reshape(b, t.shape[1], n_heads, hn).permute(0, 2, 1, 3)
    ctx = standard_attention(tr(q), tr(k), tr(v), ltor_mask, attn_drop_mask)
    ctx = ctx.permute(0, 2, 1, 3).reshape(b, s, h)
    out = linear(ctx, p[prefix + "dense.weight"], p[prefix + "dense.bias"])
    if out_drop_mask is not None:
        out = out * out_drop_mask
    return out


def mlp(x, p, prefix, out_drop_mask=None):
    """GPT2ParallelMLP.forward, mpu/sparse_transformer.py:226-234."""
    u = linear(x, p[prefix + "dense_h_to_4h.weight"], p[prefix + "dense_h_to_4h.bias"])
    out = linear(gelu(u), p[prefix + "dense_4h_to_h.weight"], p[prefix + "dense_4h_to_h.bias"])
    if out_drop_mask is not None:
        out = out * out_drop_mask
    return out


def transformer_layer(x, ltor_mask, p, prefix, n_heads, eps=1e-5, drop=None, mem=None):
    """GPT2ParallelTransformerLayer.forward, mpu/sparse_transformer.py:314-342 (Sandwich-LN: 4 LayerNorms).
    drop: optional dict with keys 'attn', 'attn_out', 'mlp_out' holding scaled keep masks."""
    drop = drop or {}
    ln = lambda t, name: sandwich_layernorm(t, p[prefix + name + ".weight"], p[prefix + name + ".bias"], eps)
    a = ln(x, "input_layernorm")
    mem_n = ln(mem, "input_layernorm") if mem is not None else None
    att = self_attention(a, ltor_mask, p, prefix + "attention.", n_heads, drop.get("attn"), drop.get("attn_out"), mem_n)
    att = ln(att, "third_layernorm")
    y = x + att
    c = ln(y, "post_attention_layernorm")
    m = mlp(c, p, prefix + "mlp.", drop.get("mlp_out"))
    m = ln(m, "fourth_layernorm")
    return y + m


def gpt2_forward(ids, position_ids, ltor_mask, p, n_layers, n_heads, eps=1e-5, drops=None, emb_drop_mask=None):
    """GPT2Model.forward (model/gpt2_modeling.py:106-123) + GPT2ParallelTransformer.forward
    (mpu/sparse_transformer.py:471-613), dense attention, no mems.  Returns logits [b, s, V]."""
    x = F.embedding(ids, p["word_embeddings.weight"])                       # mpu/layers.py:117-133
    x = x + F.embedding(position_ids, p["transformer.position_embeddings.weight"])   # :522-523
    if emb_drop_mask is not None:
        x = x * emb_drop_mask                                                # :524
    for l in range(n_layers):
        x = transformer_layer(x, ltor_mask, p, f"transformer.layers.{l}.", n_heads, eps,
                              None if drops is None else drops[l])
    x = sandwich_layernorm(x, p["transformer.final_layernorm.weight"], p["transformer.final_layernorm.bias"], eps)
    return linear(x, p["word_embeddings.weight"])                            # tied logits, gpt2_modeling.py:117


def vocab_parallel_cross_entropy(logits, target):
    """mpu/cross_entropy.py:25-78 for one shard: loss = log(sum(exp(l - max))) - (l[target] - max)."""
    l = logits.float()
    m = l.max(dim=-1, keepdim=True)[0]
    l = l - m
    return torch.log(l.exp().sum(-1)) - l.gather(-1, target.unsqueeze(-1)).squeeze(-1)


def vocab_parallel_cross_entropy_sharded(logit_shards, target):
    """The same with the vocabulary split over len(logit_shards) ranks; the three all-reduces
    (MAX, SUM, SUM at mpu/cross_entropy.py:34,42,70) are played by explicit reductions over the list."""
    gmax = torch.stack([s.float().max(-1)[0] for s in logit_shards]).max(0)[0]
    gsum, pred, start = 0, 0, 0
    for s in logit_shards:
        sh = s.float() - gmax.unsqueeze(-1)
        gsum = gsum + sh.exp().sum(-1)
        v = s.shape[-1]
        inside = (target >= start) & (target < start + v)
        t = (target - start).clamp(0, v - 1)
        pred = pred + torch.where(inside, sh.gather(-1, t.unsqueeze(-1)).squeeze(-1), torch.zeros_like(gmax))
        start += v
    return torch.log(gsum) - pred


def lm_loss(logits, labels, loss_mask, txt_mask=None, txt_loss_scale=1.0):
    """pretrain_gpt2.py:310-325: masked mean of the per-token CE, text positions weighted."""
    losses = vocab_parallel_cross_entropy(logits.contiguous().float(), labels)
    lm = loss_mask.clone().float()
    if txt_mask is not None:
        lm[txt_mask] *= txt_loss_scale
    lm = lm.view(-1)
    return torch.sum(losses.view(-1) * lm) / lm.sum()


IMG_TXT_SEP = 8192      # data_utils/unified_tokenizer.py:32-67: image codes are ids [0, 8192)


def forward_step_losses(logits, tokens, labels, loss_mask, txt_loss_scale=1.0):
    """pretrain_gpt2.py:304-331 on one rank: (loss, img_loss, txt_loss).  Image / text positions are told apart by the INPUT
    token id (:305-306), text positions additionally need a non-zero mask; the logged partial losses are sums of the already
    WEIGHTED per-token losses over the boolean sets divided by the set sizes (pads inside the image set count in the
    denominator), the text one divided by the scale again (:330-331)."""
    img = tokens < IMG_TXT_SEP
    txt = (~img) & (loss_mask > 0)
    losses = vocab_parallel_cross_entropy(logits.contiguous().float(), labels)
    lm = loss_mask.clone().float()
    lm[txt] *= txt_loss_scale
    lm = lm.view(-1)
    losses = losses.view(-1) * lm
    loss = torch.sum(losses) / lm.sum()
    img, txt = img.view(-1), txt.view(-1)
    img_loss = losses[img].detach().sum() / max(int(img.sum()), 1)
    txt_loss = losses[txt].detach().sum() / max(int(txt.sum()), 1) / txt_loss_scale
    return loss, img_loss, txt_loss


# --------------------------------------------------------------------------------------------- optimizer
def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """apex FusedAdam(adam_w_mode=True, bias_correction=True) as called at pretrain_gpt2.py:139-140
    ("decoupled weight decay", comment at :128).  Published definition (apex multi_tensor_adam.cu, ADAM_MODE_1):
        m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; update = (m/bc1) / (sqrt(v/bc2) + eps) + wd*p ; p -= lr*update
    Operates in place on float tensors; returns None."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1, bc2 = 1 - beta1 ** step, 1 - beta2 ** step
    update = (m / bc1) / ((v / bc2).sqrt() + eps) + weight_decay * p
    p.add_(update, alpha=-lr)


def clip_grad_norm(grads, max_norm):
    """mpu/grads.py:28-74 (norm_type 2, model-parallel size 1).  In place; returns the total norm."""
    total = math.sqrt(sum(float(g.double().norm(2)) ** 2 for g in grads))
    coef = max_norm / (total + 1e-6)
    if coef < 1:
        for g in grads:
            g.mul_(coef)
    return total


class DynamicLossScaler:
    """fp16/loss_scaler.py:63-172 (state machine only; the overflow flag comes from the caller)."""

    def __init__(self, init_scale=2 ** 32, scale_factor=2., scale_window=1000, min_scale=1, delayed_shift=1,
                 consecutive_hysteresis=False):
        self.cur_scale, self.cur_iter, self.last_overflow_iter = init_scale, 0, -1
        self.scale_factor, self.scale_window, self.min_scale = scale_factor, scale_window, min_scale
        self.delayed_shift = self.cur_hysteresis = delayed_shift
        self.consecutive_hysteresis = consecutive_hysteresis

    def update_scale(self, overflow):
        if overflow:
            if self.delayed_shift == 1 or self.cur_hysteresis == 1:
                self.cur_scale = max(self.cur_scale / self.scale_factor, self.min_scale)
            else:
                self.cur_hysteresis -= 1
            self.last_overflow_iter = self.cur_iter
        else:
            if self.consecutive_hysteresis:
                self.cur_hysteresis = self.delayed_shift
            if (self.cur_iter - self.last_overflow_iter) % self.scale_window == 0:
                if not self.consecutive_hysteresis:
                    self.cur_hysteresis = self.delayed_shift
                self.cur_scale *= self.scale_factor
        self.cur_iter += 1


# --------------------------------------------------------------------------------------------- VQ-VAE
def vqvae_encode(img, p):
    """vqvae/vqvae_zc.py:121-129,159-164 (Encoder, stride 6, simple) + :41-54 (Quantize.forward_, eval):
    three 4x4 stride-2 convs + ReLU, ReLU, 1x1 conv, NHWC, nearest code by
    dist = |x|^2 - 2 x E + |E|^2 and argmax(-dist).  Returns ids [b, h/8, w/8]."""
    x = F.relu(F.conv2d(img, p["enc_b.blocks.0.weight"], p["enc_b.blocks.0.bias"], stride=2, padding=1))
    x = F.relu(F.conv2d(x, p["enc_b.blocks.2.weight"], p["enc_b.blocks.2.bias"], stride=2, padding=1))
    x = F.conv2d(x, p["enc_b.blocks.4.weight"], p["enc_b.blocks.4.bias"], stride=2, padding=1)
    x = F.conv2d(F.relu(x), p["enc_b.blocks.6.weight"], p["enc_b.blocks.6.bias"])
    x = x.permute(0, 2, 3, 1)
    flat = x.reshape(-1, x.shape[-1])
    e = p["quantize_t.embed"]
    dist = flat.pow(2).sum(1, keepdim=True) - 2 * flat @ e + e.pow(2).sum(0, keepdim=True)
    ind = (-dist).max(1)[1]
    return ind.view(*x.shape[:-1]), x, dist


def vqvae_decode(ids, p):
    """vqvae/vqvae_zc.py:95-96 (embed_code), :264-269 (decode_code), :172-192 (Decoder, stride 4, simple):
    embedding lookup, NCHW, three 4x4 stride-2 transposed convs with ReLU, 1x1 conv."""
    q = F.embedding(ids, p["quantize_t.embed"].t()).permute(0, 3, 1, 2)
    x = F.relu(F.conv_transpose2d(q, p["dec.blocks.0.weight"], p["dec.blocks.0.bias"], stride=2, padding=1))
    x = F.relu(F.conv_transpose2d(x, p["dec.blocks.2.weight"], p["dec.blocks.2.bias"], stride=2, padding=1))
    x = F.relu(F.conv_transpose2d(x, p["dec.blocks.4.weight"], p["dec.blocks.4.bias"], stride=2, padding=1))
    return F.conv2d(x, p["dec.blocks.6.weight"], p["dec.blocks.6.bias"])


def code2img_denorm(out):
    """vqvae/api.py:43 de-normalisation."""
    std = torch.tensor([0.30379, 0.32279, 0.32800]).view(1, -1, 1, 1)
    mean = torch.tensor([0.79093, 0.76271, 0.75340]).view(1, -1, 1, 1)
    return out * std + mean


# --------------------------------------------------------------------------------------------- dropout RNG
# NumPy restatement of the device generator in cogview_amd/csrc/common.cuh (pcg32 / xorshift32 / rng_key),
# so that tests can apply bit-identical dropout masks on the oracle side.
_U = np.uint32


def _pcg32(x):
    x = np.asarray(x, dtype=np.uint32)
    with np.errstate(over="ignore"):
        state = x * _U(747796405) + _U(2891336453)
        word = ((state >> ((state >> _U(28)) + _U(4))) ^ state) * _U(277803737)
    return (word >> _U(22)) ^ word


def _xorshift32(x):
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x << _U(13)
    x ^= x >> _U(17)
    x ^= x << _U(5)
    return x


def rng_key(seed, stream):
    seed, stream = int(seed) & (2 ** 64 - 1), int(stream) & (2 ** 64 - 1)
    k = _pcg32(_U(((stream >> 32) + 0x9E3779B9) & 0xFFFFFFFF))
    k = _pcg32(_U(stream & 0xFFFFFFFF) ^ k)
    k = _pcg32(_U(seed >> 32) ^ k)
    k = _pcg32(_U(seed & 0xFFFFFFFF) ^ k)
    return _U(k)


def _words(key, ctr, nwords):
    ctr = np.asarray(ctr, dtype=np.uint64)
    lo = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (ctr >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        w0 = _pcg32((lo ^ key) + hi * _U(0x85EBCA6B))
    ws = [w0, _xorshift32(w0 ^ _U(0x68E31DA4))]
    while len(ws) < nwords:
        ws.append(_xorshift32(ws[-1]))
    return ws[:nwords]


def _thr16(p):
    return int(p * 65536.0 + 0.5)


def dropout_keep_mask(n_elements, p, seed, stream):
    """Scaled keep mask (float32 numpy, length n_elements) of the element-wise convention:
    group G = e >> 3, bits = (word[(e&7)>>1] >> 16*(e&1)) & 0xffff, keep iff bits >= round(p*65536)."""
    thr = _thr16(p)
    e = np.arange(n_elements, dtype=np.uint64)
    ws = _words(rng_key(seed, stream), e >> np.uint64(3), 4)
    j = (e & np.uint64(7)).astype(np.int64)
    w = np.choose(j >> 1, ws)
    bits = (w >> ((j & 1) * 16).astype(np.uint32)) & _U(0xFFFF)
    keep = bits >= thr
    return keep.astype(np.float32) * np.float32(65536.0 / (65536.0 - thr))


def attention_keep_mask(b, heads, s_q, s_k, p, seed, stream):
    """Scaled keep mask [b, heads, s_q, s_k] of the attention convention (cogview_amd/csrc/attention.hip, "Attention
    dropout bits"): one 64-bit draw per (attention row, 4 consecutive keys); the row is hashed once with PCG, the key
    group enters as a Weyl step and each of the two words gets one multiply-xorshift round:
        rk = pcg32((lo32(row) ^ key) + hi32(row) * 0x85EBCA6B);  x = rk + (key // 4) * 0x9E3779B9
        x ^= x >> 15; x *= 0x2C1B3C6D; x ^= x >> 12                  (word 0)
        y = (x ^ 0x68E31DA4) * 0x297A2D39; y ^= y >> 15              (word 1)
    element i = key & 3 takes bits 16 (i & 1) .. +15 of word i >> 1; keep iff bits >= round(p * 65536)."""
    thr = _thr16(p)
    key = rng_key(seed, stream)
    rows = np.arange(b * heads * s_q, dtype=np.uint64).reshape(-1, 1)
    keys = np.arange(s_k, dtype=np.uint64).reshape(1, -1)
    lo = (rows & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    hi = (rows >> np.uint64(32)).astype(np.uint32)
    with np.errstate(over="ignore"):
        rk = _pcg32((lo ^ key) + hi * _U(0x85EBCA6B))
        x = rk + (keys >> np.uint64(2)).astype(np.uint32) * _U(0x9E3779B9)
        x = x ^ (x >> _U(15))
        x = x * _U(0x2C1B3C6D)
        x = x ^ (x >> _U(12))
        y = (x ^ _U(0x68E31DA4)) * _U(0x297A2D39)
        y = y ^ (y >> _U(15))
    i = (keys & np.uint64(3)).astype(np.int64) + np.zeros(x.shape, dtype=np.int64)
    w = np.where((i >> 1) == 0, x, y)
    bits = (w >> ((i & 1) * 16).astype(np.uint32)) & _U(0xFFFF)
    keep = (bits >= thr).astype(np.float32) * np.float32(65536.0 / (65536.0 - thr))
    return keep.reshape(b, heads, s_q, s_k)
