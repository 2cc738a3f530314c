"""FP16_Optimizer outside the fused flat path (fp16/fp16.py:322-453 step by step): loose 16-bit parameters and an inner optimizer
that is not this package's FusedAdam.  The wrapper must say so (a RuntimeWarning naming the reason), keep the reference's
semantics -- fp32 masters, overflow check on the model gradients, skip + scale halving, unscale, clip on the masters, inner step,
master -> model copy -- and run its norm / overflow reductions on cogv_grad_stats (fp32 instantiation for the masters), not on
torch reductions.  Checked against the same arithmetic written out in fp32 torch on the CPU."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_generic_path_warns_and_follows_the_reference_step(dtype):
    assert torch.cuda.is_available(), "GPU tests need an MI355X; run with -m 'not gpu' elsewhere"
    from tests.optimizer_cases import run_generic_path_case
    run_generic_path_case(dtype, "cuda")


def test_fused_path_does_not_warn():
    assert torch.cuda.is_available(), "GPU tests need an MI355X; run with -m 'not gpu' elsewhere"
    from cogview_amd.fp16 import FP16_Module, FP16_Optimizer
    from cogview_amd.optim import FusedAdam
    lin = torch.nn.Linear(64, 32).cuda()
    mod = FP16_Module(lin)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        opt = FP16_Optimizer(FusedAdam(mod.parameters(), lr=1e-3), dynamic_loss_scale=True)
    assert opt._arena is not None
