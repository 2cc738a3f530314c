"""Time the REFERENCE ITSELF beside the oracle port on BASELINE.json configs[0] (4 layers / 256 hidden / 4 heads, vocabulary
58240, 4 rows of 256 tokens): forward + cross entropy + backward in fp32 on this container's CPU threads, median of 5 timed
iterations after one warm-up -- the reference imported unmodified from /root/reference with the shims of gen_golden.py, the
port = oracle/cogview_oracle.py on the same weights and rows.  Build container only (the GPU box has no /root/reference); the
result is committed as profiles/r05_cfg1_cpu_reference_vs_port.json and quoted by `bench.py --config cogview-tiny-18M` as
`cpu_baseline_reference` (labelled with where it was measured), next to the port timed live on the GPU box's host cores.

    python oracle/time_reference_cfg1.py
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
from gen_golden_cfg1 import CFG1                # noqa: E402


def main():
    mpu, st = install_shims()
    from model.gpt2_modeling import GPT2Model
    from oracle import cogview_oracle as O
    c = CFG1
    torch.manual_seed(c["seed"])
    model = GPT2Model(c["layers"], c["vocab"], c["hidden"], c["heads"], 0.0, 0.0, 0.0, c["row_len"], 0, False)
    rows = torch.randint(0, c["n_ids"], (c["rows"], c["row_len"]), generator=torch.Generator().manual_seed(c["seed"]))
    tokens, labels = rows[:, :-1].contiguous(), rows[:, 1:].contiguous()
    s = c["row_len"] - 1
    pos = torch.arange(s).unsqueeze(0).expand(c["rows"], -1)
    mask = torch.tril(torch.ones(1, 1, s, s))
    lm = torch.ones(c["rows"], s).view(-1)

    def ref_iter():
        model.zero_grad(set_to_none=True)
        logits, = model(tokens, pos, mask, None, None, 0)
        losses = mpu.vocab_parallel_cross_entropy(logits.contiguous().float(), labels)
        loss = torch.sum(losses.view(-1) * lm) / lm.sum()
        loss.backward()
        return loss.item()

    params = {n: p.detach().clone().requires_grad_(True) for n, p in model.state_dict().items()}
    omask = O.build_mask(s, s)

    def port_iter():
        for p in params.values():
            p.grad = None
        logits = O.gpt2_forward(tokens, pos, omask, params, c["layers"], c["heads"])
        loss = O.lm_loss(logits, labels, torch.ones(c["rows"], s))
        loss.backward()
        return loss.item()

    out = {"config": "BASELINE configs[0]: 4L/256h/4 heads, vocab 58240, 4 rows of 256 tokens (s = 255), fp32, forward + CE + backward; median of 5 iterations",
           "where": "build container (no GPU)", "threads": torch.get_num_threads(), "tokens_per_iteration": c["rows"] * s}
    for name, fn in (("reference", ref_iter), ("port", port_iter)):
        loss = fn()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        med = statistics.median(ts)
        out[name] = {"loss": loss, "seconds_per_iteration_median_of_5": med, "tokens_per_s": c["rows"] * s / med}
    out["port_over_reference_speed"] = out["port"]["tokens_per_s"] / out["reference"]["tokens_per_s"]
    path = os.path.join(os.path.dirname(HERE), "profiles", "r05_cfg1_cpu_reference_vs_port.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
