"""Golden vectors for the host side of incremental decoding, produced by the reference's own
functions.  generation/sampling.py cannot be imported as a module here (it pulls pretrain_gpt2 -> apex / deepspeed), so
the three pure functions are taken out of its source with `ast` and executed as they stand.
    python oracle/gen_golden_sampling.py        (build container only: reads /root/reference)
Test infrastructure: writes tests/golden/sampling.npz."""
import ast
import os

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/generation/sampling.py"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_functions(names):
    tree = ast.parse(open(REF).read())
    ns = {"torch": torch, "F": F}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            exec(compile(ast.Module(body=[node], type_ignores=[]), REF, "exec"), ns)
    return [ns[n] for n in names]


def main():
    top_k_logits, shrink_beams, add_marks = reference_functions(["top_k_logits", "shrink_beams", "add_interlacing_beam_marks"])
    g = torch.Generator().manual_seed(11)
    out = {}
    logits = torch.randn(3, 257, generator=g) * 3
    out["logits"] = logits.numpy().copy()
    out["topk_40"] = top_k_logits(logits.clone(), top_k=40).numpy()
    out["topk_1"] = top_k_logits(logits.clone(), top_k=1).numpy()
    out["topp_09"] = top_k_logits(logits[:1].clone(), top_p=0.9).numpy()
    out["topk_topp"] = top_k_logits(logits[1:2].clone(), top_k=30, top_p=0.5).numpy()
    tokens = torch.arange(12).view(3, 4)
    mems = [torch.arange(3 * 5 * 2, dtype=torch.float32).view(3, 5, 2), torch.ones(3, 5, 0)]
    t2, m2, s2 = shrink_beams(tokens, mems, 1, [-3.0, -1.5, -2.0])
    out["shrink_tokens"], out["shrink_mem0"], out["shrink_score"] = t2.numpy(), m2[0].numpy(), np.array(s2, dtype=np.float64)
    seq = [5, 6, -1, -1, -1, 7, -1, -1, -1, -1, -1, 9]
    add_marks(seq, nb=3, period=2)
    out["marks"] = np.array(seq)
    np.savez(os.path.join(ROOT, "tests", "golden", "sampling.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
