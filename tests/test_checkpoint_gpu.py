"""Checkpoint round trips WITH optimizer state on the GPU (reference utils.py:188-234,289-380,
fp16/fp16.py:336-397).

  * resume: train 2 steps -> save_checkpoint(optimizer=opt) -> fresh model / optimizer -> load_checkpoint -> step 3 must
    be BIT-identical to the uninterrupted run: 16-bit weights, fp32 masters, Adam m / v, loss scale and the dropout
    counters (dropout is on, so a wrong RNG restore changes the masks and the weights).  No tensor is exempt: since
    round 3 the embedding backward sums in fp32 in token order and rounds once (no 16-bit atomics), so the word- and
    position-embedding tables are bit-identical too.
  * fine-tune from a release file (weights only): the fp32 masters must be refreshed from the loaded weights -- without
    that the first step writes the random initialisation back (reference utils.py:300-301).
"""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "gpt2_small.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _make(g, dtype, drop, seed, load_golden=True):
    from cogview_amd import mpu
    from cogview_amd.fp16 import FP16_Module, FP16_Optimizer
    from cogview_amd.model import GPT2Model, gpt2_get_params_for_weight_decay_optimization
    from cogview_amd.optim import FusedAdam
    L_, V_, H_, NH_, P_, S_, B_ = [int(v) for v in g["cfg"]]
    torch.manual_seed(seed)
    mpu.model_parallel_cuda_manual_seed(seed)
    m = GPT2Model(L_, V_, H_, NH_, drop, drop, drop, P_, 0, False)
    if load_golden:
        m.load_state_dict({k[6:]: v for k, v in g.items() if k.startswith("param.")})
    model = FP16_Module(m.cuda(), dtype=dtype, keep_half_outputs=True)
    groups = gpt2_get_params_for_weight_decay_optimization(model.module)
    for grp in groups:
        for p in grp['params']:
            if not hasattr(p, 'model_parallel'):
                p.model_parallel = False
    opt = FP16_Optimizer(FusedAdam(groups, lr=1e-3, weight_decay=0.01), dynamic_loss_scale=True,
                         dynamic_loss_args={'init_scale': 2 ** 10, 'scale_window': 2, 'min_scale': 1, 'delayed_shift': 1})
    assert opt._arena is not None
    model.train()
    return model, opt


def _batch(g):
    L_, V_, H_, NH_, P_, S_, B_ = [int(v) for v in g["cfg"]]
    pos = torch.arange(S_).unsqueeze(0).expand(B_, -1)
    return (g["tokens"].cuda(), g["labels"].cuda(), g["loss_mask"].cuda(), 0, pos.cuda())


def _args(path, **kw):
    a = types.SimpleNamespace(save=path, load=path, deepspeed=False, no_save_optim=False, no_save_rng=False,
                              no_load_optim=False, no_load_rng=False, finetune=False)
    for k, v in kw.items():
        setattr(a, k, v)
    return a


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_resume_with_optimizer_state_is_bit_identical(golden_dir, tmp_path, dtype):
    from cogview_amd import training, utils
    g = _golden(golden_dir)
    batch = _batch(g)
    # uninterrupted: 3 steps (scale_window 2 -> the loss scale doubles after step 2: the scaler state matters)
    model, opt = _make(g, dtype, drop=0.1, seed=1234)
    for _ in range(2):
        training.train_step(batch, model, opt, clip_grad=1.0)
    utils.save_checkpoint(2, model, opt, None, _args(str(tmp_path)))
    loss3, _ = training.train_step(batch, model, opt, clip_grad=1.0)
    want = {"w": model.module._cogv_arena.data.clone(), "master": opt._master_flat.clone(), "m": opt._m_flat.clone(),
            "v": opt._v_flat.clone(), "scale": opt.loss_scale, "loss": loss3.item()}
    assert opt._step_count == 3
    # resumed: a fresh model from a DIFFERENT seed and random weights, everything must come from the file
    model2, opt2 = _make(g, dtype, drop=0.1, seed=99, load_golden=False)
    it = utils.load_checkpoint(model2, opt2, None, _args(str(tmp_path)))
    assert it == 2 and opt2._step_count == 2
    sd = torch.load(utils.get_checkpoint_name(str(tmp_path), 2), map_location="cpu", weights_only=False)
    assert {"optimizer", "rng_tracker_states", "cogv_default_dropout_state", "module", "iteration"} <= set(sd)
    assert sd["optimizer"]["optimizer_state_dict"]["param_groups"][0]["step"] == 2
    loss3b, _ = training.train_step(batch, model2, opt2, clip_grad=1.0)
    assert loss3b.item() == want["loss"]
    assert opt2.loss_scale == want["scale"] and opt2._step_count == 3
    assert torch.equal(model2.module._cogv_arena.data, want["w"])
    assert torch.equal(opt2._master_flat, want["master"])
    assert torch.equal(opt2._m_flat, want["m"]) and torch.equal(opt2._v_flat, want["v"])


def test_load_without_cogv_step_count_falls_back_to_param_group_step(golden_dir, tmp_path):
    """A state dict written by the generic path / the reference (no 'cogv_step_count'): the Adam step comes from the
    param groups, so bias correction does not restart."""
    from cogview_amd import training
    g = _golden(golden_dir)
    model, opt = _make(g, torch.float16, drop=0.0, seed=1)
    for _ in range(2):
        training.train_step(_batch(g), model, opt, clip_grad=1.0)
    sd = opt.state_dict()
    sd.pop("cogv_step_count")
    model2, opt2 = _make(g, torch.float16, drop=0.0, seed=1)
    opt2.load_state_dict(sd)
    assert opt2._step_count == 2
    assert torch.equal(opt2._m_flat, opt._m_flat) and torch.equal(opt2._master_flat, opt._master_flat)


@pytest.mark.parametrize("how", ["release", "finetune", "no_load_optim"])
def test_weights_only_load_refreshes_the_fp32_masters(golden_dir, tmp_path, how):
    from cogview_amd import training, utils
    g = _golden(golden_dir)
    model, opt = _make(g, torch.float16, drop=0.0, seed=5)
    a = _args(str(tmp_path))
    utils.save_checkpoint(7, model, opt, None, a)
    if how == "release":                       # what the released cogview-base files look like: weights only
        os.rename(os.path.join(str(tmp_path), "7"), os.path.join(str(tmp_path), "release"))
        sd = torch.load(utils.get_checkpoint_name(str(tmp_path), 0, release=True), map_location="cpu", weights_only=False)
        torch.save({"module": sd["module"]}, utils.get_checkpoint_name(str(tmp_path), 0, release=True))
        with open(utils.get_checkpoint_tracker_filename(str(tmp_path)), "w") as f:
            f.write("release")
    loaded = model.module._cogv_arena.data.clone()
    model2, opt2 = _make(g, torch.float16, drop=0.0, seed=77, load_golden=False)       # random weights and masters
    assert not torch.equal(model2.module._cogv_arena.data, loaded)
    it = utils.load_checkpoint(model2, opt2, None, _args(str(tmp_path), finetune=how == "finetune",
                                                          no_load_optim=how == "no_load_optim"))
    assert it == (7 if how == "no_load_optim" else 0)
    assert torch.equal(model2.module._cogv_arena.data, loaded)
    assert torch.equal(opt2._master_flat.to(torch.float16), loaded), "fp32 masters were not refreshed from the loaded weights"
    training.train_step(_batch(g), model2, opt2, clip_grad=1.0)
    after = model2.module._cogv_arena.data.float()
    rel = ((after - loaded.float()).norm() / loaded.float().norm()).item()
    assert rel < 0.1, f"one fine-tuning step moved the weights by {rel}: the loaded weights were thrown away"


def test_fp32_data_parallel_gradients_stay_in_the_arena(golden_dir):
    """fp32 mode (no FP16_Optimizer): FusedAdam.zero_grad must keep param.grad as views of the flat gradient buffer the
    data-parallel wrapper reduces (apex's set_grad_none default would detach them and the replicas would diverge)."""
    import torch.distributed as dist
    from cogview_amd import mpu, training
    from cogview_amd.model import DistributedDataParallel, GPT2Model, gpt2_get_params_for_weight_decay_optimization
    from cogview_amd.optim import FusedAdam
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("gloo", world_size=1, rank=0)
    if not mpu.model_parallel_is_initialized():
        mpu.initialize_model_parallel(1)
    g = _golden(golden_dir)
    L_, V_, H_, NH_, P_, S_, B_ = [int(v) for v in g["cfg"]]
    from cogview_amd.fp16 import FP16_Module
    m = GPT2Model(L_, V_, H_, NH_, 0.0, 0.0, 0.0, P_, 0, False)
    m.load_state_dict({k[6:]: v for k, v in g.items() if k.startswith("param.")})
    model = FP16_Module(m.cuda(), dtype=torch.bfloat16, keep_half_outputs=True)     # 16-bit kernels, bare optimizer
    ddp = DistributedDataParallel(model, force_collectives=True)
    opt = FusedAdam(gpt2_get_params_for_weight_decay_optimization(model.module), lr=1e-4)
    arena = ddp.arena
    for _ in range(2):
        loss, _, _, _ = training.forward_step(_batch(g), ddp, log=False)
        training.backward_step(opt, ddp, loss, clip_grad=0.0, fp16=False)
        esz = arena.grad.element_size()
        for p, off in zip(arena.params, arena.offsets):
            assert p.grad is not None and p.grad.data_ptr() == arena.grad.data_ptr() + off * esz
        assert float(arena.grad.float().abs().sum()) > 0
        opt.step()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_lazy_zero_grad_gives_the_gradients_of_a_memset(golden_dir, dtype):
    """arena.zero_grad(lazy=True) skips the memset and lets the first backward kernel of every gradient overwrite it
    (training.backward_step runs that way).  The gradients must equal the memset path bit for bit -- from a POISONED
    buffer, over one and over two accumulated backward passes; gradients no kernel reaches become zeros when the
    gradients are consumed.  Every tensor, the embedding tables included (deterministic since round 3)."""
    from cogview_amd import training
    g = _golden(golden_dir)
    batch = _batch(g)
    model, opt = _make(g, dtype, drop=0.0, seed=1234)
    arena = model.module._cogv_arena
    first = arena.params[0].numel()

    def run(lazy, passes):
        if lazy:
            arena.grad.fill_(7.0)                                  # stale garbage everywhere
            opt.zero_grad(lazy=True)
        else:
            opt.zero_grad()
        for _ in range(passes):
            loss, *_ = training.forward_step(batch, model, 1.0)
            opt.backward(loss, update_master_grads=False)
        if lazy:
            assert arena._fresh == set()                           # every gradient was written by a kernel
            opt.finish_lazy_zero_grad()
            assert arena._fresh is None
        return arena.grad.clone()

    for passes in (1, 2):
        want, got = run(False, passes), run(True, passes)
        for prm, off in zip(arena.params, arena.offsets):
            w, g_ = want[off:off + prm.numel()], got[off:off + prm.numel()]
            assert torch.equal(w, g_), (passes, off)
    # no backward pass at all: whoever consumes the gradients finds zeros, not the stale values
    arena.grad.fill_(7.0)
    opt.zero_grad(lazy=True)
    assert float(arena.params[3].grad.float().abs().max()) == 7.0
    opt.finish_lazy_zero_grad()
    assert all(float(p.grad.float().abs().max()) == 0.0 for p in arena.params)
    # the whole step through training.train_step (lazy inside) == the same step with the memset path
    ref_model, ref_opt = _make(g, dtype, drop=0.0, seed=1234)
    ref_opt.zero_grad()
    loss, *_ = training.forward_step(batch, ref_model, 1.0)
    ref_opt.backward(loss, update_master_grads=False)
    ref_opt.update_master_grads()
    ref_opt.step()
    model2, opt2 = _make(g, dtype, drop=0.0, seed=1234)
    model2.module._cogv_arena.grad.fill_(3.0)
    training.train_step(batch, model2, opt2, clip_grad=0.0)
    a, b = ref_opt._master_flat, opt2._master_flat
    assert torch.equal(a, b)
