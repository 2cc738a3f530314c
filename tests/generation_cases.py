"""Shared body of the generation-golden tests: this package's own generation loop (cogview_amd.generation: filling_sequence,
inverse_prompt_score) over its own GPT2Model must reproduce what the REFERENCE's generation code produced over the reference's
fp32 model (oracle/gen_golden_generate.py -> tests/golden/generate_samples.npz): the 40 image codes of a text -> image fill-in
with two beams, and the two post-selection scores of a 1037-token row.

GPU: tests/test_generation_golden_gpu.py (through the C ABI; layer-input memories and the in-place key/value cache);
CPU: tests/test_generation_cpu.py (tests/cpu_ops.py emulation -- the host loop, positions, memories, masks).
The weights are not stored: the constructor under torch.manual_seed(seed) draws the reference's bits."""
import os
import types

import numpy as np
import torch

MARKERS = ["[ROI1]", "[BASE]", "[BOI1]", "[EOI1]", "[ROI2]", "[BOI2]", "[EOI2]", "[POS0]"]       # oracle/gen_golden_generate.py
COIN_FLIP = 0.004            # top-2 gap (in units of the logits' std) below which 16-bit arithmetic may pick the other token


class ToyIds:
    """The id layout the golden was generated on: image codes, text pieces, then eight markers."""

    def __init__(self, img_tokens, txt_tokens):
        self.img_tokenizer = types.SimpleNamespace(num_tokens=img_tokens)
        self.txt_tokenizer = types.SimpleNamespace(num_tokens=txt_tokens)
        self.ids = {m: img_tokens + txt_tokens + i for i, m in enumerate(MARKERS)}

    def __getitem__(self, name):
        return self.ids[name]


def load_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "generate_samples.npz"))
    keys = ("layers", "hidden", "heads", "max_pos", "max_mem", "seed", "img_tokens", "txt_tokens", "divisible_by", "n_generate", "beams")
    return z, dict(zip(keys, (int(v) for v in z["cfg"])))


def build_model(z, c, dev, kv_cache, dtype=torch.float16, seed=None, **sparse_cfg):
    from cogview_amd.fp16 import FP16_Module
    from cogview_amd.model import GPT2Model
    torch.manual_seed(c["seed"] if seed is None else seed)
    m = GPT2Model(c["layers"], int(z["vocab"]), c["hidden"], c["heads"], 0.1, 0.1, 0.1, c["max_pos"], c["max_mem"], False,
                  kv_cache=kv_cache, **sparse_cfg)
    m = m.to(dev)
    return FP16_Module(m, dtype=dtype).eval()


def check_tokens(out, z, c, key="t2i"):
    want, gaps = z[key + "_out"], z[key + "_gaps"]
    assert tuple(out.shape) == want.shape
    n_gen = int((z[key + "_seq"] < 0).sum())
    n_ctx = want.shape[1] - n_gen
    for beam in range(want.shape[0]):
        got = out[beam].tolist()
        assert got[:n_ctx] == want[beam, :n_ctx].tolist()
        for i in range(n_gen):
            if got[n_ctx + i] != int(want[beam, n_ctx + i]):
                assert gaps[i] < COIN_FLIP, (beam, i, got[n_ctx + i], int(want[beam, n_ctx + i]), float(gaps[i]))
                break
    assert float(gaps.min()) > COIN_FLIP                 # with this fixture: every token of every beam must match


def run_generation_golden_case(golden_dir, dev, kv_cache, score_atol=5e-3):
    from cogview_amd.generation import add_interlacing_beam_marks, filling_sequence, inverse_prompt_score
    z, c = load_golden(golden_dir)
    ids = ToyIds(c["img_tokens"], c["txt_tokens"])
    args = types.SimpleNamespace(temperature=1.0, top_k=1, top_p=0.0, is_sparse=0)
    model = build_model(z, c, dev, kv_cache)
    seq = torch.from_numpy(z["t2i_seq"]).to(dev)
    add_interlacing_beam_marks(seq, nb=c["beams"])
    out = filling_sequence(model, seq.clone(), args, tokenizer=ids)
    check_tokens(out.cpu(), z, c)
    scores = inverse_prompt_score(model, torch.from_numpy(z["sel_seq"]).to(dev), args, tokenizer=ids)
    assert scores.dtype == torch.float32 and tuple(scores.shape) == (2,)
    assert np.allclose(scores.cpu().numpy(), z["sel_scores"], rtol=0, atol=score_atol), (scores.tolist(), z["sel_scores"].tolist())
    return out, scores


def run_sparse_generation_golden_case(golden_dir, dev):
    """is_sparse = 2 (mpu/sparse_transformer.py:497-520, 590-601, 727-750): trailing window of 2 x 16 positions, every layer of every
    pass draws its pivots -- all text positions + a random subset of the image positions left of the window -- with
    `random.sample`; `random` is seeded as the golden's generator seeded it, so the draws are the reference's."""
    import random
    from cogview_amd.generation import filling_sequence
    z, c = load_golden(golden_dir)
    sp = dict(zip(("seed", "query_window", "key_window_times", "num_pivot", "n_generate", "random_seed"), (int(v) for v in z["sparse_cfg"])))
    ids = ToyIds(c["img_tokens"], c["txt_tokens"])
    args = types.SimpleNamespace(temperature=1.0, top_k=1, top_p=0.0, is_sparse=2)
    model = build_model(z, c, dev, False, seed=sp["seed"], query_window=sp["query_window"], key_window_times=sp["key_window_times"],
                        num_pivot=sp["num_pivot"])
    random.seed(sp["random_seed"])
    out = filling_sequence(model, torch.from_numpy(z["sparse_seq"]).to(dev), args, tokenizer=ids)
    check_tokens(out.cpu(), z, c, key="sparse")
    return out
