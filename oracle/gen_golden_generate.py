"""Golden vectors for the GENERATION drivers (generate_samples.py -> generation/sampling.py), produced by running the REFERENCE
ITSELF: its own GPT2Model (fp32, CPU) under its own `filling_sequence` (generation/sampling.py:65-201: context pass, then one
token at a time over the memories, beams, forbidden id ranges) and `inverse_prompt_score` (:222-239).

    python oracle/gen_golden_generate.py        (build container only: reads /root/reference)

Test infrastructure: writes tests/golden/generate_samples.npz, which tests/test_reference_drivers_cpu.py holds the mirrors to
(the reference's generate_samples.py, unedited, driving `cogview_amd` on the CPU-emulated ops).

Shims: those of oracle/gen_golden.py for the reference's mpu / model on a CPU, plus what `import pretrain_gpt2` pulls in at
module level and this path never calls (apex.optimizers.FusedAdam, deepspeed.add_config_arguments, tensorboardX, data_utils
-- the tokenizer below stands in for data_utils.get_tokenizer: only its id ranges and its marker ids are read).

Sampling is made deterministic the way the reference's own options allow: --top_k 1 leaves one candidate with probability 1,
so torch.multinomial has nothing to draw.  For every generated position the gap between the two largest admissible logits is
stored, so that the consumer can tell a wrong token from a coin flip between two near-equal logits under 16-bit arithmetic."""
import os
import sys
import types

import numpy as np
import torch

from gen_golden import OUT, install_shims, npz

# seed 17: of the seeds 11..20 the one whose 40 greedy choices are the most decisive (smallest top-2 gap 0.010 of the logits' std;
# the others have a choice at 0.0002 .. 0.007, which 16-bit arithmetic may legitimately flip)
CFG = dict(layers=4, hidden=256, heads=4, max_pos=1089, max_mem=1089, seed=17, img_tokens=8192, txt_tokens=500, divisible_by=128,
           n_generate=40, beams=2)
# sparse generation (is_sparse = 2, mpu/sparse_transformer.py:497-520, 590-601, 727-750): a second, smaller-window model so that 64
# generated codes outgrow the trailing window (2 x 16 positions) and every layer samples pivots (all text positions + a random
# subset of the image positions left of the window, drawn with `random.sample`: the consumer seeds `random` the same way).
# seed 27: of 23..31 the most decisive (smallest top-2 gap 0.020 of the logits' std)
SPARSE = dict(seed=27, query_window=16, key_window_times=2, num_pivot=256, n_generate=64, random_seed=99)
MARKERS = ["[ROI1]", "[BASE]", "[BOI1]", "[EOI1]", "[ROI2]", "[BOI2]", "[EOI2]", "[POS0]"]


class ToyTokenizer:
    """The id layout of data_utils/unified_tokenizer.py:25-60: image codes first, then text pieces, then the markers."""

    def __init__(self, cfg=CFG):
        self.img_tokenizer = types.SimpleNamespace(num_tokens=cfg["img_tokens"])
        self.txt_tokenizer = types.SimpleNamespace(num_tokens=cfg["txt_tokens"])
        base = cfg["img_tokens"] + cfg["txt_tokens"]
        self.ids = {m: base + i for i, m in enumerate(MARKERS)}
        self.num_tokens = base + len(MARKERS)
        self.decoded = []

    def __getitem__(self, name):
        return self.ids[name]

    def DecodeIds(self, ids):
        self.decoded.append(list(ids))
        return ["<text>"], [torch.zeros(1, 3, 8, 8)]


def padded_vocab(tok, cfg=CFG):
    n = tok.num_tokens
    while n % cfg["divisible_by"]:
        n += 1
    return n


def scenario(tok, cfg=CFG):
    """text -> image: [ROI1] five text pieces [BASE] [BOI1] then n_generate image codes to fill in;
    post-selection: [BASE] [BOI1] 1024 image codes [EOI1] [ROI1] eight text pieces, two candidates (the second with other text)."""
    g = torch.Generator().manual_seed(cfg["seed"] + 1)
    txt = lambda n: torch.randint(cfg["img_tokens"], cfg["img_tokens"] + cfg["txt_tokens"], (n,), generator=g)
    img = lambda n: torch.randint(0, cfg["img_tokens"], (n,), generator=g)
    t2i = torch.cat([torch.tensor([tok["[ROI1]"]]), txt(5), torch.tensor([tok["[BASE]"], tok["[BOI1]"]]),
                     torch.full((cfg["n_generate"],), -1, dtype=torch.long)])
    image = img(1024)
    head = torch.cat([torch.tensor([tok["[BASE]"], tok["[BOI1]"]]), image, torch.tensor([tok["[EOI1]"], tok["[ROI1]"]])])
    sel = torch.stack([torch.cat([head, txt(8)]), torch.cat([head, txt(8)])])
    return t2i, sel


def sparse_sequence(tok, cfg=CFG, sp=SPARSE):
    g = torch.Generator().manual_seed(sp["seed"] + 1)
    txt = torch.randint(cfg["img_tokens"], cfg["img_tokens"] + cfg["txt_tokens"], (5,), generator=g)
    return torch.cat([torch.tensor([tok["[ROI1]"]]), txt, torch.tensor([tok["[BASE]"], tok["[BOI1]"]]),
                      torch.full((sp["n_generate"],), -1, dtype=torch.long)])


def sampling_args(cfg=CFG):
    return types.SimpleNamespace(is_sparse=0, temperature=1.0, top_k=1, top_p=0.0, finetune=False,
                                 max_position_embeddings=cfg["max_pos"], max_position_embeddings_finetune=cfg["max_pos"])


def stub_modules(tok):
    apex = sys.modules.get("apex") or types.ModuleType("apex")
    opt = types.ModuleType("apex.optimizers")
    opt.FusedAdam = torch.optim.Adam                      # named at import time by pretrain_gpt2.py:40, never constructed here
    apex.optimizers = opt
    sys.modules.update({"apex": apex, "apex.optimizers": opt})
    sys.modules["deepspeed"].add_config_arguments = lambda parser: parser
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None})
    sys.modules["tensorboardX"] = tbx
    du = types.ModuleType("data_utils")
    du.get_tokenizer = lambda args=None: tok
    du.make_loaders = du.detect_new_datasets = lambda *a, **k: None
    sys.modules["data_utils"] = du


def main():
    mpu, st = install_shims()
    tok = ToyTokenizer()
    stub_modules(tok)
    from model.gpt2_modeling import GPT2Model
    from generation.sampling import filling_sequence, inverse_prompt_score, add_interlacing_beam_marks
    c = CFG
    vocab = padded_vocab(tok)
    torch.manual_seed(c["seed"])
    model = GPT2Model(num_layers=c["layers"], vocab_size=vocab, hidden_size=c["hidden"], num_attention_heads=c["heads"],
                      embedding_dropout_prob=0.1, attention_dropout_prob=0.1, output_dropout_prob=0.1,
                      max_sequence_length=c["max_pos"], max_memory_length=c["max_mem"], checkpoint_activations=False,
                      checkpoint_num_layers=1, parallel_output=True, query_window=128, key_window_times=6, num_pivot=768)
    model.eval()
    args = sampling_args()
    t2i, sel = scenario(tok)

    # every forward pass' last-position logits, to measure how decisive each greedy choice was
    gaps, real_forward = [], model.forward

    def recording_forward(*a, **k):
        out = real_forward(*a, **k)
        last = out[0][0, -1, :c["img_tokens"]].detach().double()          # admissible ids after [BOI1]: the image codes
        top2 = torch.topk(last, 2)[0]
        gaps.append(float(top2[0] - top2[1]) / float(last.std()))
        return out

    model.forward = recording_forward
    seq = t2i.clone()
    add_interlacing_beam_marks(seq, nb=c["beams"])
    with torch.no_grad():
        out_tokens = filling_sequence(model, seq.clone(), args)
    model.forward = real_forward
    with torch.no_grad():
        scores = inverse_prompt_score(model, sel, args)
    assert out_tokens.shape == (c["beams"], t2i.numel()) and torch.equal(out_tokens[0], out_tokens[1])

    # sparse generation
    import random
    sp = SPARSE
    torch.manual_seed(sp["seed"])
    smodel = GPT2Model(num_layers=c["layers"], vocab_size=vocab, hidden_size=c["hidden"], num_attention_heads=c["heads"],
                       embedding_dropout_prob=0.1, attention_dropout_prob=0.1, output_dropout_prob=0.1,
                       max_sequence_length=c["max_pos"], max_memory_length=c["max_mem"], checkpoint_activations=False,
                       checkpoint_num_layers=1, parallel_output=True, query_window=sp["query_window"],
                       key_window_times=sp["key_window_times"], num_pivot=sp["num_pivot"])
    smodel.eval()
    sargs = sampling_args()
    sargs.is_sparse = 2
    sseq = sparse_sequence(tok)
    sgaps, real_forward = [], smodel.forward

    def recording_sparse_forward(*a, **k):
        out = real_forward(*a, **k)
        last = out[0][0, -1, :c["img_tokens"]].detach().double()
        top2 = torch.topk(last, 2)[0]
        sgaps.append(float(top2[0] - top2[1]) / float(last.std()))
        return out

    smodel.forward = recording_sparse_forward
    random.seed(sp["random_seed"])
    with torch.no_grad():
        sparse_out = filling_sequence(smodel, sseq.clone(), sargs)
    assert sparse_out.shape == (1, sseq.numel())
    print("sparse generation", sparse_out[0, -sp["n_generate"]:].tolist())
    print("smallest gap, sparse generation: %.4f" % min(sgaps))
    print("generated", out_tokens[0, -c["n_generate"]:].tolist())
    print("smallest gap between the two best admissible logits, in units of the logits' std: %.4f" % min(gaps))
    print("scores", scores.tolist())
    npz("generate_samples.npz", cfg=np.array([c[k] for k in ("layers", "hidden", "heads", "max_pos", "max_mem", "seed", "img_tokens",
                                                             "txt_tokens", "divisible_by", "n_generate", "beams")]),
        vocab=np.int64(vocab), t2i_seq=t2i, t2i_out=out_tokens, t2i_gaps=np.array(gaps), sel_seq=sel, sel_scores=scores.double(),
        sparse_cfg=np.array([sp[k] for k in ("seed", "query_window", "key_window_times", "num_pivot", "n_generate", "random_seed")]),
        sparse_seq=sseq, sparse_out=sparse_out, sparse_gaps=np.array(sgaps))


if __name__ == "__main__":
    main()
