"""Parity AT DEPTH (round-2 verdict, item 1): the HIP forward of the configurations BASELINE.json names -- cfg 2
(24 layers / 1024 hidden / 16 heads) and cfg 3/4 (48 layers / 2560 hidden / 40 heads, vocabulary 58240), one sequence of
1088 positions -- against the fp32 CPU oracle run through ALL layers on the same storage-rounded weights (seed 1234,
ids randint(0, 58219); LayerNorm affines and biases perturbed so they are not the trivial 1 / 0 of a fresh model).

Bars (relative L2 of the logits against the fp32 oracle): fp16 < 1e-3 -- BASELINE.json's north-star number, asserted at
the depth it is quoted for; bf16 < 8e-3 (8 significant bits: 2^-9 per rounding point).  The residual stream after 1, 2,
4, 8, 16, 24, 32, 48 layers is compared too and the growth printed: with the stream held in fp32 the error no longer
accumulates with depth (round 2, 16-bit stream: 5.6e-4 after one layer -> 1.8e-3 after 48 in fp16, 1.4e-2 in bf16).
Both configurations also run the oracle's BACKWARD pass: loss and every parameter gradient through all 24 / 48 layers,
and cfg 2 split over two model-parallel ranks compares every gradient SHARD with the matching slice of the unsharded gradient.
Reference: layer loop mpu/sparse_transformer.py:571-613, logits model/gpt2_modeling.py:106-123.
"""
import os

import pytest
import torch

from oracle import cogview_oracle as O
from oracle import depth_check as D

pytestmark = pytest.mark.gpu

CFG = {"cogview-small-336M": (24, 1024, 16), "cogview-base-4B": (48, 2560, 40)}
VOCAB, N_IDS, S = 58240, 58219, 1088
LOGIT_TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}
STREAM_TOL = {torch.float16: 8e-4, torch.bfloat16: 6e-3}
GRAD_TOL = {torch.float16: 2e-3, torch.bfloat16: 1.5e-2}      # <= 2x the measured worst tensor (round 4: 9.7e-4 / 7.7e-3 at 48 layers,
                                                                 # 1.1e-3 for the model-parallel shards)


def _build(cfg, dtype):
    from cogview_amd.fp16 import FP16_Module
    from cogview_amd.model import GPT2Model
    L, h, heads = CFG[cfg]
    torch.manual_seed(1234)
    m = GPT2Model(L, VOCAB, h, heads, 0.1, 0.1, 0.1, S + 1, 0, False)
    with torch.no_grad():
        for _, p in m.named_parameters():
            if p.dim() == 1:
                p.add_(0.05 * torch.randn_like(p))
    return FP16_Module(m.cuda(), dtype=dtype, keep_half_outputs=True), L, heads


def _ids():
    return torch.randint(0, N_IDS, (1, S), generator=torch.Generator().manual_seed(1234))


def _row(seed=99):
    row = torch.randint(0, N_IDS, (1, S + 1), generator=torch.Generator().manual_seed(seed))
    return row[:, :-1].contiguous(), row[:, 1:].contiguous()


# One oracle pass per (configuration, dtype) serves both tests below (round 5: the GPU suite ran 653 s of the driver's 1200-s
# limit, more than half of it the CPU oracle at 48 layers).  Where gradients are checked, the oracle's forward + backward of ONE
# sequence also yields the logits and the residual streams of that sequence; the 48-layer bf16 configuration gets the
# (forward-only) logits / stream check -- its gradients were measured at 48 layers in round 4 (7.7e-3,
# profiles/r04_depth_gradient_parity_48L_mp2.log) and are asserted here at 24 layers, fp16's at 24 AND 48.
WITH_GRADIENTS = {("cogview-small-336M", torch.float16), ("cogview-small-336M", torch.bfloat16), ("cogview-base-4B", torch.float16)}
_PASS = {}


def _oracle_pass(cfg, dtype):
    key = (cfg, dtype)
    if key in _PASS:
        return _PASS[key]
    from cogview_amd import training
    model, L, heads = _build(cfg, dtype)
    model.eval()
    tokens, labels = _row()
    out = {"L": L}
    keep = sorted({n for n in D.REPORT_LAYERS if n <= L} | {0, L})
    logits, mems = D.hip_streams(model.module, tokens.cuda())
    logits32 = D.hip_logits_fp32_out(model.module, mems[L])         # the same logits without their last rounding to 16 bits
    params = D.storage_rounded_params(model.module)
    if key in WITH_GRADIENTS:
        lmask = torch.ones(1, S)
        pos = torch.arange(S).unsqueeze(0)
        batch = (tokens.cuda(), labels.cuda(), lmask.cuda(), 0, pos.cuda())
        loss, _, _, _ = training.forward_step(batch, model, log=False)
        # fp16 needs the loss scale the training step runs with (fp16/loss_scaler.py): d(loss)/d(logit) ~ 1e-5 / 1088 is
        # below fp16's smallest subnormal.  A power of two, so unscaling is exact.
        scale = 2.0 ** 14 if dtype == torch.float16 else 1.0
        (loss * scale).backward()
        l_ref, g_ref, secs, ref_logits, ref_streams = D.oracle_loss_and_grads(tokens, labels, lmask, params, L, heads, keep=keep)
        worst, worst_n, by_layer = 0.0, "", {}
        for n, p in model.module.named_parameters():
            e = D.rel_l2(p.grad.float() / scale, g_ref[n])
            if n.startswith("transformer.layers."):
                li = int(n.split(".")[2])
                by_layer[li] = max(by_layer.get(li, 0.0), e)
            if e > worst:
                worst, worst_n = e, n
        out.update(loss=loss.item(), loss_ref=l_ref.item(), worst=worst, worst_n=worst_n, by_layer=by_layer)
        del g_ref
    else:
        ref_logits, ref_streams, secs = D.oracle_streams(tokens, params, L, heads, keep=keep)
    out.update(logits=D.rel_l2(logits, ref_logits), logits32=D.rel_l2(logits32, ref_logits),
               stream={n: D.rel_l2(mems[n], ref_streams[n]) for n in keep}, secs=secs)
    del model, logits, logits32, mems, ref_logits, ref_streams, params
    torch.cuda.empty_cache()
    _PASS[key] = out
    return out


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cfg", ["cogview-small-336M", "cogview-base-4B"])
def test_logits_and_residual_stream_at_full_depth(cfg, dtype):
    rep = _oracle_pass(cfg, dtype)
    print(f"\n[{cfg} {dtype}] logits rel-L2 {rep['logits']:.3e} (written in fp32: {rep['logits32']:.3e}); residual stream after n layers: " +
          " ".join(f"{n}:{e:.2e}" for n, e in rep["stream"].items()) + f" (oracle {rep['secs']:.0f}s)")
    assert rep["logits"] < LOGIT_TOL[dtype], rep
    assert max(rep["stream"].values()) < STREAM_TOL[dtype], rep
    assert rep["stream"][0] < 1e-6          # the embedding sum is exact in the fp32 stream


@pytest.mark.parametrize("cfg,dtype", sorted(WITH_GRADIENTS, key=str))
def test_gradients_at_full_depth_vs_oracle(cfg, dtype):
    """Loss and EVERY parameter gradient of cfg 2 (24 layers, fp16 and bf16) and of cfg 3/4 -- the 48-layer / 2560-wide model the
    metric is quoted on, in its headline type fp16 -- against the oracle's autograd through all layers (fp32, the same
    storage-rounded weights, one sequence of 1088 positions, dropout off: parity protocol of SURVEY section 8c).  The reference's
    backward runs through all 48 layers (mpu/random.py:332-372, fp16/fp16.py:494-567): a dgrad wrong by a constant in layer 40
    fails here."""
    r = _oracle_pass(cfg, dtype)
    by_layer = r["by_layer"]
    print(f"\n[{cfg} {dtype}] loss {r['loss']:.5f} (oracle {r['loss_ref']:.5f}, {r['secs']:.0f}s); worst gradient rel-L2 {r['worst']:.2e} "
          f"({r['worst_n']}); worst per layer: " + " ".join(f"{li}:{by_layer[li]:.1e}" for li in (0, 1, 3, 7, 15, 23, 31, 39, 47) if li in by_layer))
    assert abs(r["loss"] - r["loss_ref"]) < 2e-3 * abs(r["loss_ref"]), (r["loss"], r["loss_ref"])
    assert r["worst"] < GRAD_TOL[dtype], (r["worst"], r["worst_n"])


# ------------------------------------------------------------------------------------------------ cfg 3: model parallel = 2
GRAD_SCALE = 2.0 ** 14               # fp16: the loss scale of the training step (a power of two, exact to undo)
MP_VOCAB = 58368                     # 58219 padded to a multiple of 128 x 2 (arguments.py --make-vocab-size-divisible-by)


def _perturb_replicated(module, seed=77):
    """Non-trivial LayerNorm affines and row-parallel biases (replicated over the model-parallel group): the same
    perturbation on every rank and in the unsharded reference, keyed by the parameter's position."""
    with torch.no_grad():
        for k, (n, p) in enumerate(module.named_parameters()):
            replicated = p.dim() == 1 and ("layernorm" in n or n.endswith("attention.dense.bias") or n.endswith("dense_4h_to_h.bias"))
            if replicated:
                p.add_(0.05 * torch.randn(p.shape, generator=torch.Generator().manual_seed(seed + k)).to(p.device, p.dtype))


def _mp2_worker(rank, world, port, cfg, out_dir, ret, grads=False):
    import sys
    import traceback
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
        torch.cuda.set_device(0)
        from cogview_amd import mpu
        from cogview_amd.fp16 import FP16_Module
        from cogview_amd.model import GPT2Model
        mpu.initialize_model_parallel(world)
        L, h, heads = CFG[cfg]
        torch.manual_seed(1234)                            # every rank draws the FULL master weights, keeps its shard
        m = GPT2Model(L, MP_VOCAB, h, heads, 0.1, 0.1, 0.1, S + 1, 0, False)
        _perturb_replicated(m)
        model = FP16_Module(m.cuda(), dtype=torch.float16, keep_half_outputs=True).eval()
        pos = torch.arange(S, device="cuda").unsqueeze(0)
        if grads:
            # loss through the vocab-parallel cross entropy (mpu/cross_entropy.py:25-104) and the whole reverse pass:
            # dgrad all-reduces of the column-parallel layers, identity backward of the row-parallel ones
            from cogview_amd import training
            tokens, labels = _row()
            batch = (tokens.cuda(), labels.cuda(), torch.ones(1, S, device="cuda"), 0, pos)
            loss, _, _, _ = training.forward_step(batch, model, log=False)
            (loss * GRAD_SCALE).backward()
            torch.save({"loss": loss.item(),
                        "grads": {n: (p.grad.detach().float() / GRAD_SCALE).cpu() for n, p in model.module.named_parameters()}},
                       os.path.join(out_dir, f"grads_{rank}.pt"))
        else:
            with torch.no_grad():
                logits, = model(_ids().cuda(), pos, 0, None, None, 0)
            assert logits.shape == (1, S, MP_VOCAB // world)
            torch.save(logits.float().cpu(), os.path.join(out_dir, f"logits_{rank}.pt"))
        ret[rank] = "ok"
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        ret[rank] = traceback.format_exc()


@pytest.mark.parametrize("cfg", ["cogview-small-336M", "cogview-base-4B"])
def test_model_parallel_2_logits_at_full_depth(cfg, tmp_path):
    """BASELINE configs[2] (the 4B model split column / row-wise over two adjacent ranks, vocabulary 58368 = 2 x 29184) at
    FULL depth: two model-parallel processes share the GPU (gloo carries the CUDA all-reduces of mpu/mappings.py:22-31 and
    the vocab-parallel embedding), their logit shards concatenated along the vocabulary against the fp32 CPU oracle on the
    UNSHARDED weights (the same seed: every rank draws the full master weights and keeps its shard, mpu/layers.py:42-74).
    fp16, bar 1e-3."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_mp2_worker, args=(r, 2, port, cfg, str(tmp_path), ret)) for r in range(2)]
        for p in procs:
            p.start()
        # meanwhile: the unsharded reference weights (rounded to fp16 as the shards are) and the oracle forward
        from cogview_amd.model import GPT2Model
        L, h, heads = CFG[cfg]
        torch.manual_seed(1234)
        full = GPT2Model(L, MP_VOCAB, h, heads, 0.1, 0.1, 0.1, S + 1, 0, False)
        _perturb_replicated(full)
        params = {n: p.detach().to(torch.float16).float() for n, p in full.state_dict().items()}
        del full
        ref, _, secs = D.oracle_streams(_ids(), params, L, heads, keep=())
        for p in procs:
            p.join(900)
        for r in range(2):
            assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"
    got = torch.cat([torch.load(os.path.join(str(tmp_path), f"logits_{r}.pt")) for r in range(2)], dim=-1)
    e = D.rel_l2(got, ref)
    print(f"\n[{cfg} fp16, model parallel 2] logits rel-L2 vs the oracle on the unsharded weights {e:.3e} (oracle {secs:.0f}s)")
    assert e < LOGIT_TOL[torch.float16], e


def _mp_slice(name, t, rank, world):
    """The shard of the full tensor `t` model-parallel rank `rank` owns (mpu/layers.py:42-74: column-parallel weights and
    biases split along dim 0 -- QKV with stride 3 --, row-parallel weights along dim 1, vocabulary rows of the embedding);
    None: replicated."""
    if name.endswith("word_embeddings.weight") or "dense_h_to_4h" in name:
        return t.chunk(world, 0)[rank]
    if "query_key_value" in name:
        slabs = t.chunk(3 * world, 0)
        return torch.cat([slabs[rank], slabs[rank + world], slabs[rank + 2 * world]], 0)
    if name.endswith("attention.dense.weight") or name.endswith("dense_4h_to_h.weight"):
        return t.chunk(world, 1)[rank]
    return None


def test_model_parallel_2_gradients_at_full_depth(tmp_path):
    """cfg 2 (24 layers / 1024 hidden) split over TWO model-parallel ranks, fp16: loss and every gradient SHARD of both
    ranks against the matching slice of the oracle's gradient on the UNSHARDED weights (autograd through all layers);
    replicated parameters (LayerNorms, row-parallel biases, position table) against the whole tensor on every rank."""
    import socket
    import torch.multiprocessing as mp
    cfg = "cogview-small-336M"
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_mp2_worker, args=(r, 2, port, cfg, str(tmp_path), ret, True)) for r in range(2)]
        for p in procs:
            p.start()
        from cogview_amd.model import GPT2Model
        L, h, heads = CFG[cfg]
        torch.manual_seed(1234)
        full = GPT2Model(L, MP_VOCAB, h, heads, 0.1, 0.1, 0.1, S + 1, 0, False)
        _perturb_replicated(full)
        params = {n: p.detach().to(torch.float16).float() for n, p in full.named_parameters()}
        del full
        tokens, labels = _row()
        l_ref, g_ref, secs = D.oracle_loss_and_grads(tokens, labels, torch.ones(1, S), params, L, heads)
        for p in procs:
            p.join(900)
        for r in range(2):
            assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"
    worst, worst_n = 0.0, ""
    for r in range(2):
        got = torch.load(os.path.join(str(tmp_path), f"grads_{r}.pt"))
        assert abs(got["loss"] - l_ref.item()) < 2e-3 * abs(l_ref.item()), (r, got["loss"], l_ref.item())
        assert set(got["grads"]) == set(g_ref)
        for n, g in got["grads"].items():
            want = _mp_slice(n, g_ref[n], r, 2)
            want = g_ref[n] if want is None else want
            assert g.shape == want.shape, (n, g.shape, want.shape)
            e = D.rel_l2(g, want)
            if e > worst:
                worst, worst_n = e, f"{n} (rank {r})"
    print(f"\n[{cfg} fp16, model parallel 2] loss {got['loss']:.5f} (oracle {l_ref.item():.5f}, {secs:.0f}s); worst gradient-shard "
          f"rel-L2 vs the slice of the unsharded oracle gradient {worst:.2e} ({worst_n})")
    assert worst < GRAD_TOL[torch.float16], (worst, worst_n)
