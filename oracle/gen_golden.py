"""Generate tests/golden/*.npz by running the REFERENCE ITSELF (read-only import from /root/reference).

Run in the build container only (the GPU box has no /root/reference):   python oracle/gen_golden.py
The reference imports apex / deepspeed / torch._six at module import; they are not installed here and not
vendored there, so they are shimmed exactly as SURVEY.md section 7.1 lists:
    torch._six.inf                       -> float('inf')
    apex FusedLayerNorm                  -> torch.nn.LayerNorm          (published semantics of the apex op)
    deepspeed.checkpointing.is_configured-> False                       (the non-DeepSpeed branch is the path)
    mpu.sparse_transformer.get_cuda_rng_tracker -> CPU no-op context   (dropout is 0 in every fixture)
    torch.cuda.FloatTensor               -> torch.FloatTensor           (mpu/grads.py:52,65 scalar staging)
Everything else is the reference's own code.  Fixtures are small (a few MB) and committed.
"""
import contextlib
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def install_shims():
    sys.path.insert(0, REF)
    six = types.ModuleType("torch._six")
    six.inf = float("inf")
    sys.modules["torch._six"] = six
    torch._six = six
    apex = types.ModuleType("apex")
    norm = types.ModuleType("apex.normalization")
    fln = types.ModuleType("apex.normalization.fused_layer_norm")
    fln.FusedLayerNorm = torch.nn.LayerNorm
    sys.modules.update({"apex": apex, "apex.normalization": norm, "apex.normalization.fused_layer_norm": fln})
    ds = types.ModuleType("deepspeed")
    ck = types.ModuleType("deepspeed.checkpointing")
    ck.is_configured = lambda: False
    ds.checkpointing = ck
    sys.modules.update({"deepspeed": ds, "deepspeed.checkpointing": ck})
    torch.cuda.FloatTensor = torch.FloatTensor
    import torch.distributed as dist
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29533", world_size=1, rank=0)
    import mpu
    mpu.initialize_model_parallel(1)
    import mpu.sparse_transformer as st

    class _NoTracker:
        @contextlib.contextmanager
        def fork(self, name=None):
            yield

    st.get_cuda_rng_tracker = lambda: _NoTracker()
    return mpu, st


def npz(name, **arrs):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                                 for k, v in arrs.items()})
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    mpu, st = install_shims()
    from model.gpt2_modeling import GPT2Model

    # ---------------------------------------------------------------- primitives
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(3, 20, 128, generator=g) * 3.0
    ln = st.LayerNorm(128, eps=1e-5)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(128, generator=g) + 0.5)
        ln.bias.copy_(torch.randn(128, generator=g) * 0.1)
    xg = x.clone().requires_grad_(True)
    y = ln(xg)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    u = torch.randn(4, 33, generator=g) * 2
    q = torch.randn(2, 2, 40, 64, generator=g)
    k = torch.randn(2, 2, 40, 64, generator=g)
    v = torch.randn(2, 2, 40, 64, generator=g)
    mask = torch.tril(torch.ones(1, 1, 40, 40))
    att = st.standard_attention(q, k, v, mask, None)
    # "sep" mask with memory: s_q = 24 queries against s_k = 40 keys, sep = 5 (sparse_transformer.py:482-488)
    msep = torch.ones(1, 24, 40)
    msep[0, :, -24:] = torch.tril(msep[0, :, -24:])
    msep[0, :, :5 + 16] = 1
    att_sep = st.standard_attention(q[:, :, -24:], k, v, msep.unsqueeze(1), None)
    logits = torch.randn(2, 7, 96, generator=g) * 4
    tgt = torch.randint(0, 96, (2, 7), generator=g)
    lg = logits.clone().requires_grad_(True)
    ce = mpu.vocab_parallel_cross_entropy(lg, tgt)
    wce = torch.rand(2, 7, generator=g)
    (ce * wce).sum().backward()
    npz("primitives.npz", ln_x=x, ln_w=ln.weight, ln_b=ln.bias, ln_y=y, ln_dy=dy, ln_dx=xg.grad, ln_dw=ln.weight.grad,
        ln_db=ln.bias.grad, gelu_x=u, gelu_y=st.gelu(u), att_q=q, att_k=k, att_v=v, att_out=att, att_sep_out=att_sep,
        ce_logits=logits, ce_target=tgt, ce_loss=ce, ce_w=wce, ce_dlogits=lg.grad)

    # ---------------------------------------------------------------- GPT2Model forward + backward (dropout 0)
    torch.manual_seed(1234)
    L_, V_, H_, NH_, P_, S_, B_ = 2, 256, 128, 2, 48, 40, 2
    model = GPT2Model(L_, V_, H_, NH_, 0.0, 0.0, 0.0, P_, 0, False)
    # the reference initialises biases to 0 and LN to (1, 0); perturb them so that the fixture exercises them
    with torch.no_grad():
        for n, p_ in model.named_parameters():
            if n.endswith("bias"):
                p_.add_(torch.randn_like(p_) * 0.02)
            if "layernorm.weight" in n:
                p_.add_(torch.randn_like(p_) * 0.05)
    ids = torch.randint(0, V_, (B_, S_ + 1), generator=g)
    tokens, labels = ids[:, :-1].contiguous(), ids[:, 1:].contiguous()
    pos = torch.arange(S_).unsqueeze(0).expand(B_, -1)
    m = torch.tril(torch.ones(1, 1, S_, S_))
    logits_, = model(tokens, pos, m, None, None, 0)
    losses = mpu.vocab_parallel_cross_entropy(logits_.contiguous().float(), labels)
    loss_mask = torch.ones(B_, S_)
    loss_mask[1, -5:] = 0
    lm = loss_mask.view(-1)
    loss = torch.sum(losses.view(-1) * lm) / lm.sum()                       # pretrain_gpt2.py:324-325
    loss.backward()
    sd = {"param." + n: p_ for n, p_ in model.state_dict().items()}
    gd = {"grad." + n: p_.grad.clone() for n, p_ in model.named_parameters()}   # clip below is in place
    # clip_grad_norm on a copy of the grads (mpu/grads.py)
    params = [p_ for p_ in model.parameters()]
    for p_ in params:
        p_.model_parallel = getattr(p_, "model_parallel", False)
    norm_before = mpu.clip_grad_norm(params, 0.05)
    # keep only two clipped tensors (the scaling is uniform) to keep the fixture small
    cd = {"clipped." + n: p_.grad for n, p_ in model.named_parameters()
          if n in ("transformer.final_layernorm.weight", "transformer.layers.1.mlp.dense_4h_to_h.bias")}
    npz("gpt2_small.npz", cfg=np.array([L_, V_, H_, NH_, P_, S_, B_]), tokens=tokens, labels=labels,
        loss_mask=loss_mask, logits=logits_, loss=loss, grad_norm=np.float64(norm_before), **sd, **gd, **cd)

    # ---------------------------------------------------------------- DynamicLossScaler trajectory
    from fp16.loss_scaler import DynamicLossScaler
    traj = {}
    for tag, kw in {"default": dict(init_scale=2 ** 16, scale_window=4),
                    "hyst": dict(init_scale=2 ** 20, scale_window=3, min_scale=256, delayed_shift=2)}.items():
        sc = DynamicLossScaler(**kw)
        pattern = [0, 0, 1, 1, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0]
        scales = []
        for o in pattern:
            sc.update_scale(bool(o))
            scales.append(sc.cur_scale)
        traj[tag + "_pattern"], traj[tag + "_scales"] = np.array(pattern), np.array(scales, dtype=np.float64)
    npz("loss_scaler.npz", **traj)

    # ---------------------------------------------------------------- VQ-VAE (same architecture, small channels)
    from vqvae.vqvae_zc import VQVAE
    torch.manual_seed(0)
    vq = VQVAE(channel=32, n_res_block=0, n_res_channel=32, embed_dim=16, n_embed=64, stride=6).eval()
    img = torch.randn(2, 3, 64, 64, generator=g)
    with torch.no_grad():
        _, _, ids_ = vq.encode(img)
        dec = vq.decode_code(ids_)
        dec_dn = dec * torch.tensor([0.30379, 0.32279, 0.32800]).view(1, -1, 1, 1) + \
            torch.tensor([0.79093, 0.76271, 0.75340]).view(1, -1, 1, 1)     # vqvae/api.py:43
    vsd = {"param." + n: p_ for n, p_ in vq.state_dict().items()}
    npz("vqvae_small.npz", img=img, ids=ids_, dec=dec, dec_denorm=dec_dn, **vsd)


if __name__ == "__main__":
    main()
