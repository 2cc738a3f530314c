"""Time the REFERENCE ITSELF beside the oracle port on the bounded sample `bench.py`'s `cpu_baseline` leg uses for the 4B
headline (cogview-base-4B: 48 layers / 2560 hidden / 40 heads, vocabulary 58240): ONE row of 1089 tokens (1088 positions)
through the embedding, some of the 48 layers, the tied LM head and the cross entropy, forward + backward in fp32 on this
container's CPU threads; the layer part is scaled to 48 layers.  The reference: its own GPT2Model, imported unmodified from
/root/reference with the shims of gen_golden.py; models of 1 and 3 layers are timed (median of 3 after a warm-up; the reference's
constructor divides by the layer count), so that one layer = (T(3) - T(1)) / 2 and embedding + head + CE = T(1) - one layer.
The port: oracle/cogview_oracle.py on the same weights and row, the same way.

Build container only (the GPU box has no /root/reference); the result is committed as profiles/r05_4B_cpu_reference_vs_port.json
and quoted by `bench.py` (default configuration) as `cpu_baseline_reference`, labelled with where it was measured, next to the
port timed live on the GPU box's host cores (`cpu_baseline`).

    python oracle/time_reference_4B_sample.py
"""
import json
import os
import statistics
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
from gen_golden import install_shims            # noqa: E402

L_FULL, H, HEADS, VOCAB, ROW, N_IDS, LO, HI = 48, 2560, 40, 58240, 1089, 58219, 1, 3


def median3(fn):
    fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return statistics.median(ts)


def main():
    mpu, st = install_shims()
    from model.gpt2_modeling import GPT2Model
    from oracle import cogview_oracle as O
    s = ROW - 1
    ids = torch.randint(0, N_IDS, (1, ROW), generator=torch.Generator().manual_seed(1))
    tokens, labels = ids[:, :-1].contiguous(), ids[:, 1:].contiguous()
    pos = torch.arange(s).unsqueeze(0)
    mask = torch.tril(torch.ones(1, 1, s, s))
    omask = O.build_mask(s, s)
    lm = torch.ones(1, s).view(-1)
    out = {"config": "cogview-base-4B sample: 1 row of 1089 tokens (s = 1088), h = 2560, 40 heads, vocab 58240, fp32, forward + CE + "
                     "backward; one layer = (T(3 layers) - T(1 layer)) / 2, embedding + tied head + CE = T(1 layer) - one layer, "
                     "scaled to 48 layers; median of 3 iterations",
           "where": "build container (no GPU)", "threads": torch.get_num_threads(), "tokens_per_iteration": s}
    times = {"reference": {}, "port": {}}
    losses = {}
    for n_layers in (LO, HI):
        torch.manual_seed(1234)
        model = GPT2Model(n_layers, VOCAB, H, HEADS, 0.0, 0.0, 0.0, ROW, 0, False)

        def ref_iter():
            model.zero_grad(set_to_none=True)
            logits, = model(tokens, pos, mask, None, None, 0)
            losses_ = mpu.vocab_parallel_cross_entropy(logits.contiguous().float(), labels)
            loss = torch.sum(losses_.view(-1) * lm) / lm.sum()
            loss.backward()
            return loss.item()

        params = {n: p.detach().clone().requires_grad_(True) for n, p in model.state_dict().items()}

        def port_iter():
            for p in params.values():
                p.grad = None
            logits = O.gpt2_forward(tokens, pos, omask, params, n_layers, HEADS)
            loss = O.lm_loss(logits, labels, torch.ones(1, s))
            loss.backward()
            return loss.item()

        losses[n_layers] = (ref_iter(), port_iter())
        times["reference"][n_layers] = median3(ref_iter)
        times["port"][n_layers] = median3(port_iter)
        del model, params
    for name in ("reference", "port"):
        layer = max(times[name][HI] - times[name][LO], 1e-9) / (HI - LO)
        head = max(times[name][LO] - LO * layer, 0.0)
        full = head + L_FULL * layer
        out[name] = {"seconds_head_embedding_ce": head, "seconds_per_layer": layer, "seconds_per_row_48_layers": full,
                     "tokens_per_s": s / full}
    out["loss_reference_vs_port_at_%d_layers" % HI] = list(losses[HI])
    out["port_over_reference_speed"] = out["port"]["tokens_per_s"] / out["reference"]["tokens_per_s"]
    path = os.path.join(os.path.dirname(HERE), "profiles", "r05_4B_cpu_reference_vs_port.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
