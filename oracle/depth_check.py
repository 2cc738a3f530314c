"""Depth parity checker -- TEST INFRASTRUCTURE ONLY (used by tests/test_depth_parity_gpu.py and by bench.py's
cpu_baseline leg as the CHECKER of the model it just timed; never on the product path).

Runs the CPU oracle (oracle/cogview_oracle.py, fp32) through ALL layers of a GPT2Model on one sequence and reports
the relative L2 error of the residual stream after chosen layers and of the logits, against the tensors the HIP path
produced with the same (storage-rounded) weights.  Reference dataflow restated: layer loop
mpu/sparse_transformer.py:571-613, final LayerNorm :612, tied logits model/gpt2_modeling.py:106-123.
"""
import time

import torch
import torch.nn.functional as F

from . import cogview_oracle as O

REPORT_LAYERS = (1, 2, 4, 8, 16, 24, 32, 48)


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@torch.no_grad()
def oracle_streams(ids, params, n_layers, n_heads, eps=1e-5, keep=None):
    """fp32 oracle forward of one batch: returns (logits [b,s,V], {n: residual stream after n layers}, seconds).
    `params`: name -> fp32 CPU tensor (GPT2Model.state_dict() naming); `keep`: layer counts whose stream to return
    (0 = the embedding output)."""
    t0 = time.perf_counter()
    b, s = ids.shape
    pos = torch.arange(s).unsqueeze(0).expand(b, -1)
    mask = O.build_mask(s, s)
    x = F.embedding(ids, params["word_embeddings.weight"]) + \
        F.embedding(pos, params["transformer.position_embeddings.weight"])
    keep = set(range(n_layers + 1)) if keep is None else set(keep)
    streams = {0: x.clone()} if 0 in keep else {}
    for l in range(n_layers):
        x = O.transformer_layer(x, mask, params, f"transformer.layers.{l}.", n_heads, eps)
        if l + 1 in keep:
            streams[l + 1] = x.clone()
    xf = O.sandwich_layernorm(x, params["transformer.final_layernorm.weight"], params["transformer.final_layernorm.bias"], eps)
    logits = O.linear(xf, params["word_embeddings.weight"])
    return logits, streams, time.perf_counter() - t0


@torch.no_grad()
def hip_streams(module, ids):
    """The HIP forward of `module` (a cogview_amd GPT2Model in 16-bit storage, model-parallel size 1, eval mode) on
    `ids` [b, s] (CUDA): (logits, [stream after 0..L layers]).  The per-layer streams come out through the reference's
    own `*mems` return (layer inputs + final output, mpu/sparse_transformer.py:526-546,615-626), so the model must have
    been built with max_memory_length >= s; when it was not, the memory length is raised for this call."""
    tr = module.transformer
    old = tr.max_memory_length
    b, s = ids.shape
    pos = torch.arange(s, device=ids.device).unsqueeze(0).expand(b, -1)
    tr.max_memory_length = max(old, s)
    try:
        logits, *mems = module(ids, pos, 0, None, None, 0)
    finally:
        tr.max_memory_length = old
    return logits, mems


@torch.no_grad()
def hip_logits_fp32_out(module, stream_out):
    """The tied-logits product of the final LayerNorm's output written in fp32 (cogv_gemm_desc.out_f32) instead of the storage
    type: the logits without their last rounding.  stream_out: the residual stream after the last layer (hip_streams' last mem)."""
    from cogview_amd import ops
    xf = module.transformer.final_layernorm(stream_out)
    w = module.word_embeddings.weight
    return ops.gemm(xf.reshape(-1, xf.shape[-1]), w, out_dtype=torch.float32).view(*xf.shape[:-1], w.shape[0])


def storage_rounded_params(module):
    """The module's parameters exactly as stored (16-bit), widened to fp32 on the CPU: the oracle then sees the same
    weights as the kernels, which isolates arithmetic error from weight rounding."""
    return {n: p.detach().float().cpu() for n, p in module.state_dict().items()}


def depth_report(module, ids, n_layers, n_heads, report_layers=REPORT_LAYERS):
    """-> dict(logits=rel-L2, stream={n: rel-L2}, oracle_seconds=..., tokens=...).  `ids` on CUDA."""
    was_training = module.training
    module.eval()
    try:
        logits, mems = hip_streams(module, ids)
    finally:
        module.train(was_training)
    keep = sorted({n for n in report_layers if n <= n_layers} | {0, n_layers})
    # the same logits WITHOUT their final rounding to the 16-bit storage type: the tied-logits product of the final LayerNorm's
    # output written in fp32 (cogv_gemm_desc.out_f32) -- separates the arithmetic error of the path from the last rounding
    logits32 = hip_logits_fp32_out(module, mems[n_layers])
    ref_logits, ref_streams, secs = oracle_streams(ids.cpu(), storage_rounded_params(module), n_layers, n_heads, keep=keep)
    return {"logits": rel_l2(logits, ref_logits), "logits_fp32_out": rel_l2(logits32, ref_logits),
            "stream": {n: rel_l2(mems[n], ref_streams[n]) for n in keep},
            "oracle_seconds": secs, "tokens": int(ids.numel())}


def host_mem_available_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) / 2 ** 20
    except OSError:
        pass
    return 0.0


def oracle_loss_and_grads(tokens, labels, loss_mask, params, n_layers, n_heads, eps=1e-5, recompute=None, keep=None):
    """fp32 oracle forward + backward of the whole model on one batch (dropout off): -> (loss, {name: gradient}, seconds).
    `params`: name -> fp32 CPU tensor (storage-rounded weights, oracle_streams() naming); they are not modified.
    Restates the reference's reverse pass by autograd through the oracle's forward -- mpu/random.py:332-372 (per-layer
    recompute) and fp16/fp16.py:494-567 only change WHEN tensors are produced, not their values, so with `recompute` each
    layer is re-run in backward exactly as the reference's --checkpoint-activations does (torch.utils.checkpoint): the
    48-layer / 2560-wide model then needs ~35 GB of host memory (weights + gradients) instead of ~80 GB.
    recompute=None: decide from /proc/meminfo (plain autograd when the activations fit with room to spare).
    keep (optional, layer counts): additionally returns (logits, {n: residual stream after n layers}) of the SAME forward pass as
    a fourth / fifth value -- one oracle pass then serves the logits / stream check and the gradient check of a configuration."""
    from torch.utils.checkpoint import checkpoint
    t0 = time.perf_counter()
    pr = {n: p.detach().clone().requires_grad_(True) for n, p in params.items()}
    b, s = tokens.shape
    if recompute is None:
        h = pr["transformer.final_layernorm.weight"].numel()
        act_gb = n_layers * b * s * (60.0 * h + 12.0 * n_heads * s) * 4 / 2 ** 30
        recompute = host_mem_available_gb() < 3.0 * act_gb + 64.0
    pos = torch.arange(s).unsqueeze(0).expand(b, -1)
    mask = O.build_mask(s, s)
    x = F.embedding(tokens, pr["word_embeddings.weight"]) + F.embedding(pos, pr["transformer.position_embeddings.weight"])
    keep_set = set(keep) if keep is not None else set()
    streams = {0: x.detach().clone()} if 0 in keep_set else {}
    for l in range(n_layers):
        pre = f"transformer.layers.{l}."
        if recompute:
            x = checkpoint(lambda t, pre=pre: O.transformer_layer(t, mask, pr, pre, n_heads, eps), x, use_reentrant=False)
        else:
            x = O.transformer_layer(x, mask, pr, pre, n_heads, eps)
        if l + 1 in keep_set:
            streams[l + 1] = x.detach().clone()
    xf = O.sandwich_layernorm(x, pr["transformer.final_layernorm.weight"], pr["transformer.final_layernorm.bias"], eps)
    logits = O.linear(xf, pr["word_embeddings.weight"])
    loss = O.lm_loss(logits, labels, loss_mask)
    loss.backward()
    out = (loss.detach(), {n: p.grad for n, p in pr.items()}, time.perf_counter() - t0)
    return out if keep is None else out + (logits.detach(), streams)
