"""cogview_amd.generation (the mirror of the reference's generation/sampling.py + magnify.py) on the CPU-emulated ops, held to
what the reference's own generation code produced on the reference's own fp32 model (tests/golden/generate_samples.npz,
oracle/gen_golden_generate.py).  No /root/reference needed: runs wherever the CPU suite runs."""
import types

import pytest
import torch

from tests import cpu_ops
from tests.generation_cases import ToyIds, run_generation_golden_case, run_sparse_generation_golden_case


@pytest.fixture()
def cpu_kernels(monkeypatch):
    import os
    import torch.distributed as dist
    from cogview_amd import mpu
    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % (29900 + os.getpid() % 90), world_size=1, rank=0)
    if not mpu.model_parallel_is_initialized():
        mpu.initialize_model_parallel(1)
    cpu_ops.install(monkeypatch.setattr)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    yield


@pytest.mark.parametrize("kv_cache", [False, True])
def test_generation_loop_reproduces_the_reference_tokens_and_scores(cpu_kernels, golden_dir, kv_cache):
    """filling_sequence: context pass under a [1, 1, s, s] mask, then one token at a time with `attention_mask = 0` over the
    memories (the reference's layer inputs, or this package's key/value caches), two beams expanded from one;
    inverse_prompt_score: one 1037-token forward, image codes excluded."""
    run_generation_golden_case(golden_dir, "cpu", kv_cache=kv_cache)


def test_sparse_generation_reproduces_the_reference_tokens(cpu_kernels, golden_dir):
    """is_sparse = 2: 64 image codes generated past a 32-position trailing window, pivots redrawn by every layer of every pass."""
    run_sparse_generation_golden_case(golden_dir, "cpu")


class _PositionalOracle:
    """A stand-in 'model' whose answer at every position is a pure function of (token, position id): with top_k = 1 the filled-in
    sequence then records exactly which token was fed at which position id -- the bookkeeping `magnify` and `filling_sequence`
    exist for (the [ROI2] position offset, lines given by earlier windows, beams shrinking) -- independent of any arithmetic."""

    def __init__(self, n_img, vocab):
        self.n_img, self.vocab, self.calls = n_img, vocab, 0

    def __call__(self, tokens, position_ids, attention_mask, txt, img, is_sparse, *mems):
        self.calls += 1
        b, s = tokens.shape
        pos = position_ids.expand(b, s) if position_ids.shape[0] != b else position_ids
        pick = (tokens * 7919 + pos * 104729 + 13) % self.n_img
        logits = torch.zeros(b, s, self.vocab)
        logits.scatter_(2, pick.unsqueeze(-1), 10.0)
        new = tokens.unsqueeze(-1).float()
        mem = torch.cat((mems[0], new), dim=1) if mems else new
        return (logits, mem)


def test_magnify_walks_the_nine_windows(cpu_kernels):
    """generation/magnify.py:22-43 with a positional stand-in model: 64 x 64 codes come back, every one an image code, each
    generated exactly once (one model call per generated code: 4096 over the nine windows); the lines an earlier window wrote
    are GIVEN to the later, overlapping windows and come back unchanged.  (The reference's own magnify is held against this
    function, token for token, in test_reference_drivers_cpu.py.)"""
    from cogview_amd.generation import filling_sequence, magnify
    ids = ToyIds(8192, 500)
    g = torch.Generator().manual_seed(5)
    small = torch.randint(0, 8192, (1024,), generator=g)
    text = torch.cat([torch.tensor([ids["[ROI1]"]]), torch.randint(8192, 8692, (4,), generator=g), torch.tensor([ids["[BASE]"], ids["[BOI1]"]])])
    model = _PositionalOracle(8192, 8704)
    args = types.SimpleNamespace(temperature=1.0, top_k=1, top_p=0.0, is_sparse=0)
    windows = []

    def fill(model, seq, args, **kw):
        done = filling_sequence(model, seq, args, **kw)
        given = seq >= 0
        assert torch.equal(done[0][given], seq[given])
        windows.append(int((~given).sum()))
        return done

    big = magnify(model, ids, small, text, args, fill=fill)
    assert tuple(big.shape) == (1, 4096) and int(big.min()) >= 0 and int(big.max()) < 8192
    assert len(windows) == 9 and sum(windows) == 4096 and windows[0] == 18 * 32
    assert model.calls == 4096
    again = magnify(_PositionalOracle(8192, 8704), ids, small, text, args)           # the default filler, tokenizer handed on
    assert torch.equal(again, big)
