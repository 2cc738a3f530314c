"""BASELINE.json configs[0] on the HIP path: 4 layers / 256 hidden / 4 heads, vocabulary 58240, 4 rows of 256 tokens ->
s = 255 positions, M = 1020 rows -- the configuration the REFERENCE runs end to end on a CPU (pretrain_gpt2.py:406-448,
BASELINE.md section 2), against what the reference itself computed there (tests/golden/gpt2_cfg1.npz, written by
oracle/gen_golden_cfg1.py).  No dimension is a multiple of 8 / 64 / 256: ragged attention blocks, GEMM tails, LayerNorm /
cross-entropy row tails.  The weights are not in the fixture: the mirror's constructors draw them under seed 1234 and
tests/test_oracle_golden.py::test_cfg1_init_is_bit_identical_to_the_reference pins them to the reference's, bit for bit.

Tolerances (relative to the reference's fp32 run on UNROUNDED weights -- here they are rounded to 16 bits, which is most of
the error): loss 2e-3 (fp16) / 5e-3 (bf16); whole-tensor |logits| 1e-3 / 8e-3 (measured 4e-7 / 2e-6); the worst single logits
ROW rel-L2 1.5e-3 / 1.2e-2 (measured 9.5e-4 / 7.4e-3); per-tensor gradient norm 2e-3 / 1e-2 (2.0e-4 / 3.1e-3); gradient tensors
rel-L2 3e-3 / 2e-2 (1.0e-3 / 8.2e-3); global gradient norm 1e-3 / 3e-3 (9e-5 / 7.4e-4).
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOSS_TOL = {torch.float16: 2e-3, torch.bfloat16: 5e-3}
LOGIT_TOL = {torch.float16: 1e-3, torch.bfloat16: 8e-3}
ROW_TOL = {torch.float16: 1.5e-3, torch.bfloat16: 1.2e-2}
NORM_TOL = {torch.float16: 2e-3, torch.bfloat16: 1e-2}
GRAD_TOL = {torch.float16: 3e-3, torch.bfloat16: 2e-2}
GNORM_TOL = {torch.float16: 1e-3, torch.bfloat16: 3e-3}


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _model(dtype):
    from cogview_amd.fp16 import FP16_Module
    from cogview_amd.model import GPT2Model
    torch.manual_seed(1234)
    return FP16_Module(GPT2Model(4, 58240, 256, 4, 0.0, 0.0, 0.0, 256, 0, False).cuda(), dtype=dtype, keep_half_outputs=True)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_cfg1_forward_backward_vs_the_reference_run(golden_dir, dtype):
    from cogview_amd import mpu, training
    z = np.load(os.path.join(golden_dir, "gpt2_cfg1.npz"))
    rows = torch.from_numpy(z["rows"]).cuda()
    model = _model(dtype)
    batch = training.get_batch(rows, torch.ones(rows.shape, device="cuda"))      # pretrain_gpt2.py:273-275 split
    tokens, pos = batch[0], batch[4]
    assert tokens.shape == (4, 255)
    logits, = model(tokens, pos, 0, None, None, 0)
    worst_row = 0.0
    for k, (b, t) in enumerate(z["logit_rows"].tolist()):
        worst_row = max(worst_row, rel(logits[b, t], torch.from_numpy(z["logits"][k])))
    e_norm = abs(logits.double().norm().item() - float(z["logits_norm"])) / float(z["logits_norm"])
    loss, _, _, _ = training.forward_step(batch, model, log=False)
    scale = 2.0 ** 12 if dtype == torch.float16 else 1.0        # a power of two: unscaling is exact
    (loss * scale).backward()
    names = [str(n) for n in z["grad_names"]]
    params = dict(model.module.named_parameters())
    assert names == list(params)
    worst_n, worst_name = 0.0, ""
    for n, ref in zip(names, z["grad_norms"]):
        e = abs(params[n].grad.double().norm().item() / scale - ref) / ref
        if e > worst_n:
            worst_n, worst_name = e, n
    worst_g = max(rel(params[k[5:]].grad.float() / scale, torch.from_numpy(z[k])) for k in z.files if k.startswith("grad."))
    for p in params.values():
        if not hasattr(p, "model_parallel"):
            p.model_parallel = False
    gnorm = float(mpu.clip_grad_norm(list(params.values()), 1e12)) / scale       # max_norm huge: nothing is scaled
    e_gn = abs(gnorm - float(z["grad_norm"])) / float(z["grad_norm"])
    print(f"\n[cfg1 {dtype}] loss {loss.item():.5f} (reference {float(z['loss']):.5f}); logits rows rel-L2 {worst_row:.2e}, "
          f"|logits| {e_norm:.1e}; per-tensor grad norms worst {worst_n:.2e} ({worst_name}); grad tensors {worst_g:.2e}; "
          f"global norm {gnorm:.5f} vs {float(z['grad_norm']):.5f} ({e_gn:.1e})")
    assert abs(loss.item() - float(z["loss"])) < LOSS_TOL[dtype] * float(z["loss"])
    assert worst_row < ROW_TOL[dtype] and e_norm < LOGIT_TOL[dtype]
    assert worst_n < NORM_TOL[dtype], (worst_n, worst_name)
    assert worst_g < GRAD_TOL[dtype]
    assert e_gn < GNORM_TOL[dtype]


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_forward_step_with_txt_loss_scale_5_vs_the_reference_run(golden_dir, dtype):
    """SURVEY 8(a) G14 off its default on the HIP path: `--txt-loss-scale 5` (scripts/pretrain_single_node.sh:40) on a mixed
    text / image / pad batch (ragged rows of 128 tokens -> s = 127) against the REFERENCE's own forward_step
    (pretrain_gpt2.py:292-341; tests/golden/forward_step_txtscale.npz, oracle/gen_golden_txtscale.py): loss, the two logged
    partial losses, per-tensor gradient norms, five gradient tensors, the global norm.  Same tolerances as the cfg 1 run."""
    from cogview_amd import mpu, training
    z = np.load(os.path.join(golden_dir, "forward_step_txtscale.npz"))
    ts = float(z["txt_loss_scale"])
    rows, mask = torch.from_numpy(z["rows"]).cuda(), torch.from_numpy(z["loss_mask"]).cuda()
    model = _model(dtype)
    batch = training.get_batch(rows, mask)
    assert batch[0].shape == (4, 127) and int((batch[2] == 0).sum()) > 0
    loss, _, img_loss, txt_loss = training.forward_step(batch, model, txt_loss_scale=ts, log=True)
    plain, _, _, _ = training.forward_step(batch, model, txt_loss_scale=1.0, log=False)
    scale = 2.0 ** 12 if dtype == torch.float16 else 1.0
    (loss * scale).backward()
    names = [str(n) for n in z["grad_names"]]
    params = dict(model.module.named_parameters())
    assert names == list(params)
    worst_n, worst_name = 0.0, ""
    for n, ref in zip(names, z["grad_norms"]):
        e = abs(params[n].grad.double().norm().item() / scale - ref) / ref
        if e > worst_n:
            worst_n, worst_name = e, n
    worst_g = max(rel(params[k[5:]].grad.float() / scale, torch.from_numpy(z[k])) for k in z.files if k.startswith("grad."))
    for p in params.values():
        if not hasattr(p, "model_parallel"):
            p.model_parallel = False
    gnorm = float(mpu.clip_grad_norm(list(params.values()), 1e12)) / scale
    e_gn = abs(gnorm - float(z["grad_norm"])) / float(z["grad_norm"])
    print(f"\n[txt_loss_scale 5 {dtype}] loss {loss.item():.5f} / img {img_loss.item():.5f} / txt {txt_loss.item():.5f} "
          f"(reference {float(z['loss']):.5f} / {float(z['img_loss']):.5f} / {float(z['txt_loss']):.5f}; unweighted {plain.item():.5f}); "
          f"per-tensor grad norms worst {worst_n:.2e} ({worst_name}); grad tensors {worst_g:.2e}; global norm {e_gn:.1e}")
    for got, key in ((loss, "loss"), (img_loss, "img_loss"), (txt_loss, "txt_loss")):
        assert abs(got.item() - float(z[key])) < LOSS_TOL[dtype] * float(z[key]), key
    assert abs(plain.item() - float(z["loss"])) > 1e-3            # the weighting is visible on this batch
    assert worst_n < NORM_TOL[dtype], (worst_n, worst_name)
    assert worst_g < GRAD_TOL[dtype]
    assert e_gn < GNORM_TOL[dtype]


def test_cfg1_train_steps_run_and_loss_falls(golden_dir):
    """Five optimizer steps of the reference's loop on the cfg 1 batch (dropout 0.1 as arguments.py:30,40 default, dynamic
    loss scale, clip 1.0): no step skipped after the scale settles, finite loss, and the loss on the fixed batch falls."""
    from cogview_amd import mpu, training
    from cogview_amd.fp16 import FP16_Module, FP16_Optimizer
    from cogview_amd.model import GPT2Model, gpt2_get_params_for_weight_decay_optimization
    from cogview_amd.optim import FusedAdam
    z = np.load(os.path.join(golden_dir, "gpt2_cfg1.npz"))
    rows = torch.from_numpy(z["rows"]).cuda()
    torch.manual_seed(1234)
    mpu.model_parallel_cuda_manual_seed(1234)
    model = FP16_Module(GPT2Model(4, 58240, 256, 4, 0.1, 0.1, 0.1, 256, 0, False).cuda(), dtype=torch.float16,
                        keep_half_outputs=True).train()
    groups = gpt2_get_params_for_weight_decay_optimization(model.module)
    for grp in groups:
        for p in grp["params"]:
            if not hasattr(p, "model_parallel"):
                p.model_parallel = False
    opt = FP16_Optimizer(FusedAdam(groups, lr=1e-3, weight_decay=0.01), dynamic_loss_scale=True,
                         dynamic_loss_args={"init_scale": 2 ** 14})
    batch = training.get_batch(rows, torch.ones(rows.shape, device="cuda"))
    losses, skipped = [], 0
    for _ in range(5):
        loss, sk = training.train_step(batch, model, opt, clip_grad=1.0)
        losses.append(loss.item())
        skipped += int(sk)
    print(f"\n[cfg1 fp16 train] losses {' '.join(f'{v:.4f}' for v in losses)}; skipped {skipped}; scale {opt.loss_scale}")
    assert all(v == v for v in losses) and skipped <= 1
    assert abs(losses[0] - float(z["loss"])) < 2e-2 * float(z["loss"])          # dropout on: same ballpark as ln(V)
    assert losses[-1] < losses[0] - 0.1
