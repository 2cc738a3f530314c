"""The HOST side of the train step on a CPU restatement of the kernels (tests/cpu_ops.py): GPT2Model -> fused layer Functions
-> tied logits -> fused CE -> backward (weight-gradient queue, in-place gradient accumulation) against the fixture the
REFERENCE produced (tests/golden/gpt2_small.npz: logits, loss, every gradient).  What this pins without a GPU is the
plumbing around the kernels: which operand goes where, accumulate / overwrite decisions, the order of the launches, the
queue's cutting of weight gradients along tile rows.  The kernels themselves are checked on the GPU (-m gpu)."""
import os

import numpy as np
import pytest
import torch

from tests import cpu_ops


def rel(a, b):
    return ((a.double() - b.double()).norm() / (b.double().norm() + 1e-30)).item()


@pytest.fixture()
def cpu_kernels(monkeypatch):
    import torch.distributed as dist
    from cogview_amd import mpu
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % (29700 + os.getpid() % 200), world_size=1, rank=0)
    if not mpu.model_parallel_is_initialized():
        mpu.initialize_model_parallel(1)
    cpu_ops.install(monkeypatch.setattr)
    yield


@pytest.mark.parametrize("round_tiles", [256, 1])
def test_forward_backward_on_cpu_kernels_matches_the_reference_fixture(cpu_kernels, golden_dir, monkeypatch, round_tiles):
    from cogview_amd import functional as F_
    from cogview_amd import mpu
    from cogview_amd.model import GPT2Model
    z = np.load(os.path.join(golden_dir, "gpt2_small.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    L_, V_, H_, NH_, P_, S_, B_ = [int(v) for v in g["cfg"]]
    # round_tiles = 1: every flush of the weight-gradient queue may launch any whole number of (one-tile) rounds, so the
    # layers' problems and the tied-logits gradient are cut and mixed across launches exactly as at production sizes
    monkeypatch.setattr(F_, "WGRAD_ROUND_TILES", round_tiles)
    monkeypatch.setattr(F_, "_WGRADS", F_._DeferredWeightGrads())
    launches = []
    real = F_.ops.gemm_grouped
    monkeypatch.setattr(F_.ops, "gemm_grouped", lambda probs, **kw: (launches.append(len(probs)), real(probs, **kw))[1])
    m = GPT2Model(L_, V_, H_, NH_, 0.0, 0.0, 0.0, P_, 0, False)
    m.load_state_dict({k[6:]: v for k, v in g.items() if k.startswith("param.")})
    m = m.half()
    pos = torch.arange(S_).unsqueeze(0).expand(B_, -1)
    logits, = m(g["tokens"], pos, 0, None, None, 0)
    assert rel(logits.float(), g["logits"]) < 1.5e-3
    lm = g["loss_mask"].view(-1)
    loss = (mpu.vocab_parallel_cross_entropy(logits.contiguous().float(), g["labels"]).view(-1) * lm).sum() / lm.sum()
    assert abs(loss.item() - float(g["loss"])) < 1e-3 * float(g["loss"])
    loss.backward()
    assert not F_._WGRADS.entries and not F_._WGRADS.callbacks and launches
    worst = max(rel(p.grad.float(), g["grad." + n]) for n, p in m.named_parameters())
    assert worst < 5e-3, worst


def test_train_step_on_cpu_kernels_updates_through_the_flat_arena(cpu_kernels, golden_dir, monkeypatch):
    """training.train_step with FP16_Module / FP16_Optimizer(FusedAdam) on the flat arena, three steps: no skipped step at a
    scale fp16 carries, the loss of the fixture on step 1, a falling loss afterwards, master weights and 16-bit weights in
    step, and the deferred forward-NaN flag leaving optimizer.overflow False on a clean step."""
    from cogview_amd import training
    from cogview_amd.fp16 import FP16_Module, FP16_Optimizer
    from cogview_amd.model import GPT2Model, gpt2_get_params_for_weight_decay_optimization
    from cogview_amd.optim import FusedAdam
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    z = np.load(os.path.join(golden_dir, "gpt2_small.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    L_, V_, H_, NH_, P_, S_, B_ = [int(v) for v in g["cfg"]]
    m = GPT2Model(L_, V_, H_, NH_, 0.0, 0.0, 0.0, P_, 0, False)
    m.load_state_dict({k[6:]: v for k, v in g.items() if k.startswith("param.")})
    model = FP16_Module(m, dtype=torch.float16, keep_half_outputs=True)
    groups = gpt2_get_params_for_weight_decay_optimization(model.module)
    for grp in groups:
        for p in grp["params"]:
            p.model_parallel = getattr(p, "model_parallel", False)
    opt = FP16_Optimizer(FusedAdam(groups, lr=1e-3, weight_decay=0.01), dynamic_loss_scale=True, dynamic_loss_args={"init_scale": 2 ** 10})
    assert opt._arena is not None
    pos = torch.arange(S_).unsqueeze(0).expand(B_, -1)
    batch = (g["tokens"], g["labels"], g["loss_mask"], 0, pos)
    losses = []
    for _ in range(3):
        loss, skipped = training.train_step(batch, model, opt, clip_grad=1.0, check_forward_nan=True)
        assert skipped == 0 and opt.overflow is False
        losses.append(loss.item())
    assert abs(losses[0] - float(g["loss"])) < 1e-3 * float(g["loss"])
    assert losses[2] < losses[1] < losses[0]
    assert torch.equal(opt._master_flat.half(), opt._arena.data)


def test_deferred_forward_nan_guard_leaves_the_optimizer_as_the_reference_does(cpu_kernels, golden_dir, monkeypatch):
    """training.train_step(check_forward_nan=True) on the fused optimizer: the forward's NaN flag reaches the host with the
    gradient statistics, AFTER backward was enqueued (pretrain_gpt2.py:414-416 returns before backward).  A non-finite forward
    must leave what the reference's early return leaves: parameters, fp32 masters and Adam moments untouched, loss scale and its
    counters unchanged, `optimizer.overflow` as it was (round-4 advisor finding: it used to stay True), no scheduler step, the
    return value (img_loss + txt_loss, 1) -- and the next clean step must run normally."""
    from cogview_amd import training
    from cogview_amd.fp16 import FP16_Module, FP16_Optimizer
    from cogview_amd.model import GPT2Model, gpt2_get_params_for_weight_decay_optimization
    from cogview_amd.optim import FusedAdam
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    z = np.load(os.path.join(golden_dir, "gpt2_small.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    L_, V_, H_, NH_, P_, S_, B_ = [int(v) for v in g["cfg"]]
    m = GPT2Model(L_, V_, H_, NH_, 0.0, 0.0, 0.0, P_, 0, False)
    m.load_state_dict({k[6:]: v for k, v in g.items() if k.startswith("param.")})
    model = FP16_Module(m, dtype=torch.float16, keep_half_outputs=True)
    groups = gpt2_get_params_for_weight_decay_optimization(model.module)
    for grp in groups:
        for p in grp["params"]:
            p.model_parallel = getattr(p, "model_parallel", False)
    opt = FP16_Optimizer(FusedAdam(groups, lr=1e-3, weight_decay=0.01), dynamic_loss_scale=True, dynamic_loss_args={"init_scale": 2 ** 10})

    class Sched:
        n = 0

        def step(self):
            Sched.n += 1

    pos = torch.arange(S_).unsqueeze(0).expand(B_, -1)
    batch = (g["tokens"], g["labels"], g["loss_mask"], 0, pos)
    loss, skipped = training.train_step(batch, model, opt, lr_scheduler=Sched(), clip_grad=1.0, check_forward_nan=True)
    assert skipped == 0 and Sched.n == 1 and opt.overflow is False
    w = model.module.transformer.final_layernorm.weight
    good = w.data.clone()
    snap = (opt._arena.data.clone(), opt._master_flat.clone(), opt._m_flat.clone(), opt._v_flat.clone(), opt.loss_scale,
            opt.loss_scaler.cur_iter, opt.loss_scaler.last_overflow_iter, opt._step_count)
    w.data[0] = float("nan")                                         # the forward goes non-finite
    snap_params = opt._arena.data.clone()
    tot, skipped = training.train_step(batch, model, opt, lr_scheduler=Sched(), clip_grad=1.0, check_forward_nan=True)
    assert skipped == 1 and not bool(torch.isfinite(tot).all()) and Sched.n == 1
    assert opt.overflow is False and opt._stats_valid is False
    assert torch.equal(opt._arena.data.view(torch.int16), snap_params.view(torch.int16))          # bit patterns: NaN != NaN
    assert torch.equal(opt._master_flat, snap[1]) and torch.equal(opt._m_flat, snap[2]) and torch.equal(opt._v_flat, snap[3])
    assert (opt.loss_scale, opt.loss_scaler.cur_iter, opt.loss_scaler.last_overflow_iter, opt._step_count) == snap[4:]
    w.data.copy_(good)                                               # repaired: the next step is an ordinary one
    loss2, skipped = training.train_step(batch, model, opt, lr_scheduler=Sched(), clip_grad=1.0, check_forward_nan=True)
    assert skipped == 0 and Sched.n == 2 and bool(torch.isfinite(loss2)) and loss2.item() < loss.item()


def test_a_backward_pass_that_dies_half_way_leaves_nothing_in_the_weight_gradient_queue(cpu_kernels, golden_dir, monkeypatch):
    """An exception between two flushes of functional._DeferredWeightGrads (here: the attention backward of the last layer
    raises) leaves queued problems whose tensors belong to a dead pass.  The next backward pass must start from an empty queue
    and produce the fixture's gradients."""
    from cogview_amd import functional as F_
    from cogview_amd import mpu
    from cogview_amd.model import GPT2Model
    z = np.load(os.path.join(golden_dir, "gpt2_small.npz"))
    g = {k: torch.from_numpy(z[k]) for k in z.files}
    L_, V_, H_, NH_, P_, S_, B_ = [int(v) for v in g["cfg"]]
    monkeypatch.setattr(F_, "_WGRADS", F_._DeferredWeightGrads())
    m = GPT2Model(L_, V_, H_, NH_, 0.0, 0.0, 0.0, P_, 0, False)
    m.load_state_dict({k[6:]: v for k, v in g.items() if k.startswith("param.")})
    m = m.half()
    pos = torch.arange(S_).unsqueeze(0).expand(B_, -1)
    lm = g["loss_mask"].view(-1)

    def loss_of():
        logits, = m(g["tokens"], pos, 0, None, None, 0)
        return (mpu.vocab_parallel_cross_entropy(logits.contiguous().float(), g["labels"]).view(-1) * lm).sum() / lm.sum()

    real = F_.ops.attention_bwd

    def boom(*a, **k):
        raise RuntimeError("injected")

    monkeypatch.setattr(F_.ops, "attention_bwd", boom)
    with pytest.raises(RuntimeError, match="injected"):
        loss_of().backward()
    assert F_._WGRADS.entries, "the dead pass left its problems behind (that is the situation under test)"
    monkeypatch.setattr(F_.ops, "attention_bwd", real)
    for p in m.parameters():
        p.grad = None
    loss_of().backward()
    assert not F_._WGRADS.entries and not F_._WGRADS.callbacks
    worst = max(rel(p.grad.float(), g["grad." + n]) for n, p in m.named_parameters())
    assert worst < 5e-3, worst


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_generic_optimizer_path_warns_and_follows_the_reference_step(cpu_kernels, monkeypatch, dtype):
    """FP16_Optimizer around loose parameters and torch.optim.SGD (fp16/fp16.py:322-453 step by step): a RuntimeWarning names why
    the fused flat path was not taken; overflow pass on the 16-bit model gradients and norm on the fp32 masters both through
    ops.grad_stats; clean step = fp32 arithmetic written out; overflowing step skipped with the scale halved.  The same case
    runs through the C ABI in tests/test_generic_optimizer_gpu.py."""
    from tests.optimizer_cases import run_generic_path_case
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    run_generic_path_case(dtype, "cpu")


def test_cpu_emulation_keeps_the_signatures_of_the_entry_points_it_replaces():
    """tests/cpu_ops.py stands in for cogview_amd.ops in every CPU test of the host path: each stand-in must accept exactly the
    parameters (names, order, defaults) of the entry point it replaces, or those tests would exercise calls the product never makes."""
    import inspect
    from cogview_amd import ops
    for name in cpu_ops.NAMES:
        real, fake = inspect.signature(getattr(ops, name)), inspect.signature(getattr(cpu_ops, name))
        want = [(p.name, p.default, p.kind) for p in real.parameters.values()]
        got = [(p.name, p.default, p.kind) for p in fake.parameters.values()]
        assert got == want, (name, got, want)


def test_layers_under_mpu_checkpoint_keep_the_outer_pass_weight_gradients(cpu_kernels):
    """mpu.checkpoint (mpu/random.py:273-372) around stand-alone layers, as the reference's GPT2Transformer uses it: the recompute
    runs a NESTED backward pass inside the outer one.  The weight gradients the nested pass defers join the outer pass' queue --
    the first form of the round-5 queue took the nested pass for a new one and dropped what the outer pass had queued (the
    tied-logits gradient: the word-embedding gradient lost its larger half).  Bit-identical gradients with and without
    checkpointing (the emulated kernels are deterministic)."""
    from tests.queue_cases import run_layers_under_checkpoint_case
    run_layers_under_checkpoint_case("cpu")
