"""Generate tests/golden/forward_step_txtscale.npz by running the REFERENCE's own `forward_step` (pretrain_gpt2.py:292-341)
with `--txt-loss-scale 5` (scripts/pretrain_single_node.sh:40) on a mixed text / image / pad batch.

Run in the build container only (the GPU box has no /root/reference):   python oracle/gen_golden_txtscale.py
Model: BASELINE.json configs[0] geometry (4 layers / 256 hidden / 4 heads, vocabulary 58240) drawn by the reference's
constructors under torch.manual_seed(1234) -- the same weights tests/golden/gpt2_cfg1_init.json pins tensor by tensor, so
they are not stored.  Batch: 4 rows of 128 tokens, each `text ids >= 8192 ... image ids < 8192 ... pad (loss_mask 0)` with
ragged lengths, so that the three branches of the loss weighting (text x 5, image x 1, pad x 0) are all populated.
Stored: the rows and the mask, loss / img_loss / txt_loss as forward_step returns them, the global gradient norm, per-tensor
gradient norms and five small gradients.
Shims on top of oracle/gen_golden.py's list (all CPU stand-ins for absent packages / CUDA constructors, none touches the
lines under test): apex.optimizers.FusedAdam (imported, never called), data_utils.get_tokenizer -> img_tokenizer.num_tokens =
8192 (data_utils/unified_tokenizer.py:32-67), tensorboardX, torch.cuda.LongTensor -> torch.LongTensor and Tensor.cuda ->
identity for mpu.broadcast_data (mpu/data.py:49,107), torch.cuda.synchronize -> no-op for utils.Timers.
"""
import sys
import types

import numpy as np
import torch

from gen_golden import install_shims, npz

CFG = dict(layers=4, vocab=58240, hidden=256, heads=4, rows=4, row_len=128, seed=1234, txt_loss_scale=5.0)
GRAD_TENSORS = ("transformer.final_layernorm.weight", "transformer.layers.0.input_layernorm.bias",
                "transformer.layers.3.mlp.dense_4h_to_h.bias", "transformer.layers.1.attention.query_key_value.bias",
                "transformer.layers.2.fourth_layernorm.weight")


def make_rows(c):
    g = torch.Generator().manual_seed(c["seed"] + 5)
    rows = torch.empty(c["rows"], c["row_len"], dtype=torch.int64)
    mask = torch.ones(c["rows"], c["row_len"], dtype=torch.int64)
    n_txt, n_pad = (17, 40, 3, 64), (0, 9, 30, 1)
    for r in range(c["rows"]):
        t, p = n_txt[r], n_pad[r]
        rows[r, :t] = torch.randint(8192, 58219, (t,), generator=g)
        rows[r, t:] = torch.randint(0, 8192, (c["row_len"] - t,), generator=g)
        if p:
            rows[r, -p:] = 58219                              # a pad id in the text range: excluded by the mask, not by its id
            mask[r, -p:] = 0
    return rows, mask


def main():
    mpu, st = install_shims()
    opt = types.ModuleType("apex.optimizers")
    opt.FusedAdam = torch.optim.AdamW
    sys.modules["apex.optimizers"] = opt
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = object
    sys.modules["tensorboardX"] = tbx
    du = types.ModuleType("data_utils")
    tok = types.SimpleNamespace(img_tokenizer=types.SimpleNamespace(num_tokens=8192))
    du.get_tokenizer = lambda args=None: tok
    du.make_loaders = du.detect_new_datasets = lambda *a, **k: None
    sys.modules["data_utils"] = du
    sys.modules["deepspeed"].add_config_arguments = lambda p: p
    torch.cuda.LongTensor = torch.LongTensor
    torch.cuda.current_device = lambda: "cpu"
    torch.cuda.synchronize = lambda *a, **k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    import pretrain_gpt2 as P
    from model.gpt2_modeling import GPT2Model
    from utils import Timers
    c = CFG
    torch.manual_seed(c["seed"])
    model = GPT2Model(c["layers"], c["vocab"], c["hidden"], c["heads"], 0.0, 0.0, 0.0, 256, 0, False)
    rows, mask = make_rows(c)
    args = types.SimpleNamespace(txt_loss_scale=c["txt_loss_scale"], world_size=1, is_sparse=0, fp16=False, finetune=False,
                                 max_position_embeddings=256, max_position_embeddings_finetune=256, reset_position_ids=False,
                                 reset_attention_mask=False, eod_mask_loss=False)
    loss, mems, img_loss, txt_loss = P.forward_step(iter([{"text": rows.clone(), "loss_mask": mask.clone()}]), model, args,
                                                    Timers(), [])
    loss.backward()
    params = list(model.parameters())
    for p in params:
        p.model_parallel = getattr(p, "model_parallel", False)
    gnorm = mpu.clip_grad_norm(params, 1e9)
    named = dict(model.named_parameters())
    tokens = rows[:, :-1]
    npz("forward_step_txtscale.npz", rows=rows, loss_mask=mask, txt_loss_scale=np.float64(c["txt_loss_scale"]),
        loss=loss, img_loss=img_loss, txt_loss=txt_loss, grad_norm=np.float64(gnorm),
        n_img=np.int64((tokens < 8192).sum().item()), n_txt=np.int64(((tokens >= 8192) & (mask[:, 1:] > 0)).sum().item()),
        grad_names=np.array(list(named)), grad_norms=np.array([p.grad.double().norm().item() for p in params]),
        **{"grad." + n: named[n].grad for n in GRAD_TENSORS})
    print("loss %.6f img %.6f txt %.6f gnorm %.6f" % (loss.item(), img_loss.item(), txt_loss.item(), gnorm))


if __name__ == "__main__":
    main()
