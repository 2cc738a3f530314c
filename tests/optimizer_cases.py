"""Shared body of the generic-path optimizer tests (GPU: tests/test_generic_optimizer_gpu.py through the C ABI; CPU:
tests/test_host_train_step_cpu.py on the emulated ops).  See the GPU test's docstring for what is checked."""
import pytest
import torch


def loose_params(gen, dtype, dev):
    shapes = [(48, 64), (64,), (33, 7), (5,)]
    return [torch.nn.Parameter((torch.randn(s, generator=gen) * 0.3).to(dtype).to(dev)) for s in shapes]


def run_generic_path_case(dtype, dev):
    from cogview_amd import ops
    from cogview_amd.fp16 import FP16_Optimizer
    gen = torch.Generator().manual_seed(21)
    params = loose_params(gen, dtype, dev)
    for p in params:
        p.model_parallel = False
    start = [p.detach().float().cpu().clone() for p in params]
    inner = torch.optim.SGD(params, lr=0.5, momentum=0.0)
    with pytest.warns(RuntimeWarning, match="per-tensor path"):
        opt = FP16_Optimizer(inner, dynamic_loss_scale=True,
                             dynamic_loss_args={"init_scale": 2.0 ** 10, "scale_window": 1000, "delayed_shift": 1})
    assert opt._arena is None

    seen = []
    real = ops.grad_stats
    ops.grad_stats = lambda *a, **k: (seen.append(a[0].dtype), real(*a, **k))[1]
    try:
        # a clean step: loss = sum_i <p_i, c_i>  ->  dL/dp_i = c_i
        coef = [torch.randn(p.shape, generator=gen).to(dtype).to(dev) for p in params]
        opt.zero_grad()
        loss = sum((p.float() * c.float()).sum() for p, c in zip(params, coef))
        opt.backward(loss)
        assert not opt.overflow
        ref_norm = sum(float((c.double() ** 2).sum()) for c in coef) ** 0.5
        got_norm = opt.clip_master_grads(ref_norm / 2)
        assert got_norm == pytest.approx(ref_norm, rel=2e-3)                 # the model gradients are 16-bit, scaled by 2^10
        opt.step()
        for p, p0, c in zip(params, start, coef):
            want = (p0 - 0.5 * 0.5 * c.float().cpu()).to(dtype)
            tol = 2.0 ** (-9 if dtype == torch.float16 else -6)
            assert ((p.detach().cpu().float() - want.float()).abs() <= tol * want.float().abs() + 1e-6).all()
        assert opt.loss_scale == 2.0 ** 10
        assert dtype in seen and torch.float32 in seen                     # overflow pass on the model grads, norm on the masters

        # an overflowing step: skipped, the parameters untouched, the scale halved
        before = [p.detach().clone() for p in params]
        opt.zero_grad()
        loss = sum((p.float() * c.float()).sum() for p, c in zip(params, coef)) * float("inf")
        opt.backward(loss)
        assert opt.overflow
        assert opt.clip_master_grads(1.0) == -1
        opt.step()
        assert all(torch.equal(p.detach(), b) for p, b in zip(params, before))
        assert opt.loss_scale == 2.0 ** 9
    finally:
        ops.grad_stats = real
