"""Generate tests/golden/gpt2_cfg1.npz + gpt2_cfg1_init.json by running the REFERENCE ITSELF on BASELINE.json configs[0]:
the one configuration the reference runs end to end on a CPU (pretrain_gpt2.py:406-448) --
4 layers / 256 hidden / 4 heads, vocabulary 58240, 4 rows of 256 tokens -> s = 255 model positions, M = 1020 rows:
no multiple of 8 / 64 / 256 anywhere, the awkward shapes of every kernel on the path.

Run in the build container only (the GPU box has no /root/reference):   python oracle/gen_golden_cfg1.py
Shims: those of oracle/gen_golden.py (same list, same reasons).  The WEIGHTS ARE NOT STORED: they are what the
reference's constructors draw under torch.manual_seed(1234) (arguments.py:123 default seed); the mirror's constructors
must draw the same bits (mpu/layers.py:42-74), which gpt2_cfg1_init.json pins tensor by tensor
(sha1 of the fp32 bytes).  Stored: token rows, loss, global gradient norm (mpu/grads.py:28-74), per-tensor gradient
norms, the complete logits of four positions, and the gradients of five small tensors.
"""
import hashlib
import json
import os

import numpy as np
import torch

from gen_golden import OUT, install_shims, npz

CFG1 = dict(layers=4, vocab=58240, hidden=256, heads=4, rows=4, row_len=256, n_ids=58219, seed=1234)
LOGIT_ROWS = ((0, 0), (1, 127), (2, 200), (3, 254))          # (sequence, position)
GRAD_TENSORS = ("transformer.final_layernorm.weight", "transformer.layers.0.input_layernorm.bias",
                "transformer.layers.3.mlp.dense_4h_to_h.bias", "transformer.layers.1.attention.query_key_value.bias",
                "transformer.layers.2.fourth_layernorm.weight")


def sha1(t):
    return hashlib.sha1(t.detach().contiguous().numpy().tobytes()).hexdigest()


def main():
    mpu, st = install_shims()
    from model.gpt2_modeling import GPT2Model
    c = CFG1
    torch.manual_seed(c["seed"])
    model = GPT2Model(c["layers"], c["vocab"], c["hidden"], c["heads"], 0.0, 0.0, 0.0, c["row_len"], 0, False)
    init = {n: {"sha1": sha1(p), "shape": list(p.shape)} for n, p in model.state_dict().items()}
    with open(os.path.join(OUT, "gpt2_cfg1_init.json"), "w") as f:
        json.dump({"cfg": c, "constructor": "GPT2Model(4, 58240, 256, 4, 0., 0., 0., 256, 0, False) under torch.manual_seed(1234)",
                   "tensors": init}, f, indent=1, sort_keys=True)
    rows = torch.randint(0, c["n_ids"], (c["rows"], c["row_len"]), generator=torch.Generator().manual_seed(c["seed"]))
    tokens, labels = rows[:, :-1].contiguous(), rows[:, 1:].contiguous()          # pretrain_gpt2.py:273-275
    s = c["row_len"] - 1
    pos = torch.arange(s).unsqueeze(0).expand(c["rows"], -1)
    mask = torch.tril(torch.ones(1, 1, s, s))
    logits, = model(tokens, pos, mask, None, None, 0)
    losses = mpu.vocab_parallel_cross_entropy(logits.contiguous().float(), labels)
    lm = torch.ones(c["rows"], s).view(-1)
    loss = torch.sum(losses.view(-1) * lm) / lm.sum()                             # pretrain_gpt2.py:324-325
    loss.backward()
    params = list(model.parameters())
    for p in params:
        p.model_parallel = getattr(p, "model_parallel", False)
    gnorm = mpu.clip_grad_norm(params, 1e9)                                       # max_norm huge: nothing is scaled
    names = [n for n, _ in model.named_parameters()]
    npz("gpt2_cfg1.npz", rows=rows, loss=loss, grad_norm=np.float64(gnorm),
        grad_names=np.array(names), grad_norms=np.array([p.grad.double().norm().item() for p in params]),
        param_norms=np.array([p.detach().double().norm().item() for p in params]),
        logit_rows=np.array(LOGIT_ROWS), logits=torch.stack([logits[b, t] for b, t in LOGIT_ROWS]),
        logits_norm=np.float64(logits.double().norm().item()),
        **{"grad." + n: dict(model.named_parameters())[n].grad for n in GRAD_TENSORS})


if __name__ == "__main__":
    main()
