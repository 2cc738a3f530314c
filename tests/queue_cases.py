"""Shared body of the weight-gradient-queue regression tests (CPU: tests/test_host_train_step_cpu.py on the emulated ops; GPU:
tests/test_model_gpu.py through the C ABI)."""
import torch


def run_layers_under_checkpoint_case(dev):
    """Stand-alone layers under mpu.checkpoint + the tied logits: gradients bit-identical to the same model without
    checkpointing (dropout off: the recompute replays nothing random; every kernel on the path is deterministic)."""
    from cogview_amd import functional as F_
    from cogview_amd import mpu
    from cogview_amd.fp16 import FP16_Module
    from cogview_amd.model import GPT2Model

    def run(use_checkpoint):
        torch.manual_seed(3)
        model = FP16_Module(GPT2Model(2, 512, 256, 4, 0.0, 0.0, 0.0, 65, 0, False).to(dev), dtype=torch.float16, keep_half_outputs=True)
        model.train()
        g = torch.Generator().manual_seed(1)
        tokens, labels = torch.randint(0, 512, (2, 64), generator=g).to(dev), torch.randint(0, 512, (2, 64), generator=g).to(dev)
        pos = torch.arange(64, device=dev).unsqueeze(0).expand(2, -1)
        inner = model.module
        tr = inner.transformer
        h = tr.embed(tokens, pos, inner.word_embeddings)
        mask = torch.tril(torch.ones(1, 1, 64, 64, device=dev))
        for layer in tr.layers:
            h = mpu.checkpoint(lambda x, mk, layer=layer: layer(x, mk), h, mask) if use_checkpoint else layer(h, mask)
        logits = F_.tied_logits(tr.final_layernorm(h), inner.word_embeddings.weight)
        loss = mpu.vocab_parallel_cross_entropy(logits.contiguous().float(), labels).mean()
        (loss * 256).backward()
        assert not F_._WGRADS.entries and not F_._NESTED_OUTER
        return {n: p.grad.detach().float().clone() for n, p in inner.named_parameters()}, loss.item()

    plain, l0 = run(False)
    ckpt, l1 = run(True)
    assert l0 == l1
    for n in plain:
        assert torch.equal(plain[n], ckpt[n]), (n, float((plain[n] - ckpt[n]).abs().max()), float(plain[n].abs().max()))
    assert float(plain["word_embeddings.weight"].abs().max()) > 0
