"""VERDICT round 5, item 8 -- ONE numerics experiment, not a product path: would a bf16 model reach the north star's 1e-3 logits
bar if the BOUNDED forward operands (Sandwich-LN outputs are O(gamma), probabilities are in [0, 1]) were written in fp16 and
multiplied on the f16 MFMA against an fp16 copy of the weights, with bf16 kept wherever its range matters?

A torch emulation of the kernel chain's ROUNDING POINTS (functional._layer_forward: which tensors are written in 16 bits, which
stay fp32) at the 4B geometry on fresh weights (the reference's initialisation: N(0, 0.02), output projections N(0, 0.02 /
sqrt(2 L)), mpu/sparse_transformer.py:344-358), one 1088-position sequence, against the same network in fp64.  Every product
is fp32-accumulated (torch fp32 matmul of the already-rounded operands), like the MFMA.  Modes:
    bf16        every 16-bit tensor and weight in bf16                      (what the bf16 model runs today)
    mixed_ln    LN outputs (a, c, final) + W_qkv, W_1, E in fp16; the rest bf16
    mixed_ln_p  mixed_ln + probabilities in fp16 against an fp16 image of V (bf16 values are exact in fp16 inside its range)
    mixed_all   mixed_ln_p + attention output / GeLU output and W_o, W_2 in fp16 (qkv, ao, mo still bf16)
    fp16        every 16-bit tensor and weight in fp16                      (the headline dtype)
The fp64 reference of a mode uses the mode's own forward weights (rounded masters), so the figure is arithmetic error only --
the same convention as bench.py's measure_parity.   GPU box:  python tools/bf16_mixed_experiment.py [layers]
"""
import json
import math
import sys

import torch

L = int(sys.argv[1]) if len(sys.argv) > 1 else 48
import os
H, NH, S, V = (int(v) for v in os.environ.get("MIXED_GEOM", "2560,40,1088,58240").split(","))      # (CPU smoke: 128,2,64,512)
dev = os.environ.get("MIXED_DEV", "cuda")
BF, HF = torch.bfloat16, torch.float16

MODES = {
    #             ln_out  w_ln   qkv  p    v    att  w_o  g    w_2  ao/mo
    "bf16":       (BF,    BF,    BF,  BF,  BF,  BF,  BF,  BF,  BF,  BF),
    "mixed_ln":   (HF,    HF,    BF,  BF,  BF,  BF,  BF,  BF,  BF,  BF),
    "mixed_ln_p": (HF,    HF,    BF,  HF,  HF,  BF,  BF,  BF,  BF,  BF),
    "mixed_all":  (HF,    HF,    BF,  HF,  HF,  HF,  HF,  HF,  HF,  BF),
    "fp16":       (HF,    HF,    HF,  HF,  HF,  HF,  HF,  HF,  HF,  HF),
}


def rnd(t, dt):
    return t.to(dt).to(t.dtype) if dt is not None else t


def sandwich_ln(x, g, b, eps=1e-5):
    c = x.abs().max() / 8
    xs = x / c
    mu = xs.mean(-1, keepdim=True)
    var = ((xs - mu) ** 2).mean(-1, keepdim=True)
    return (xs - mu) / torch.sqrt(var + eps) * g + b


def gelu(x):
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * x * (1.0 + 0.044715 * x * x)))


def make_params(gen):
    n = lambda *s, std=0.02: torch.randn(*s, device=dev, generator=gen) * std
    out_std = 0.02 / math.sqrt(2.0 * L)
    p = {"E": n(V, H), "P": n(S + 1, H), "lnf": (torch.ones(H, device=dev), torch.zeros(H, device=dev)), "layers": []}
    for _ in range(L):
        p["layers"].append({"wqkv": n(3 * H, H), "bqkv": torch.zeros(3 * H, device=dev), "wo": n(H, H, std=out_std),
                            "bo": torch.zeros(H, device=dev), "w1": n(4 * H, H), "b1": torch.zeros(4 * H, device=dev),
                            "w2": n(H, 4 * H, std=out_std), "b2": torch.zeros(H, device=dev)})
    return p


@torch.no_grad()
def forward(p, ids, mode, dt):
    """dt = torch.float64: the reference (no activation rounding; weights rounded as the mode's forward reads them);
    dt = torch.float32: the emulation (rounding points of the kernel chain)."""
    ln_o, w_ln, qkv_t, p_t, v_t, att_t, w_o, g_t, w_2, br_t = MODES[mode]
    ref = dt == torch.float64
    r = (lambda t, d: t) if ref else rnd
    W = lambda w, d: w.to(d).to(dt)
    mask = torch.tril(torch.ones(S, S, device=dev, dtype=torch.bool))
    x = (p["E"].to(w_ln)[ids] .to(dt) + p["P"].to(w_ln)[torch.arange(S, device=dev)].to(dt))     # stream (fp32 in the kernels)
    ones, zeros = torch.ones(H, device=dev, dtype=dt), torch.zeros(H, device=dev, dtype=dt)
    for lp in p["layers"]:
        a = r(sandwich_ln(x, ones, zeros), ln_o)
        qkv = r(a @ W(lp["wqkv"], w_ln).t() + lp["bqkv"].to(dt), qkv_t)
        q, k, v = (t.view(S, NH, 64).transpose(0, 1) for t in qkv.split(H, dim=-1))
        sc = (q @ k.transpose(1, 2)) / 8.0
        sc = torch.where(mask, sc, torch.full_like(sc, -10000.0))
        pr = torch.softmax(sc, dim=-1)
        att = r((r(pr, p_t) @ r(v, v_t)).transpose(0, 1).reshape(S, H), att_t)
        ao = r(att @ W(lp["wo"], w_o).t() + lp["bo"].to(dt), br_t)
        y = x + sandwich_ln(ao, ones, zeros)
        c = r(sandwich_ln(y, ones, zeros), ln_o)
        g = r(gelu(c @ W(lp["w1"], w_ln).t() + lp["b1"].to(dt)), g_t)
        mo = r(g @ W(lp["w2"], w_2).t() + lp["b2"].to(dt), br_t)
        x = y + sandwich_ln(mo, ones, zeros)
    xf = r(sandwich_ln(x, ones, zeros), ln_o)
    return xf @ W(p["E"], w_ln).t()                  # logits, before their own rounding to 16 bits


def main():
    torch.backends.cuda.matmul.allow_tf32 = False
    gen = torch.Generator(device=dev).manual_seed(1234)
    p = make_params(gen)
    ids = torch.randint(0, V - 21, (S,), device=dev, generator=gen)
    out = {"layers": L, "hidden": H, "positions": S, "weights": "fresh (reference initialisation)", "logits_rel_l2_vs_fp64": {}}
    for mode in MODES:
        ref = forward(p, ids, mode, torch.float64)
        got = forward(p, ids, mode, torch.float32)
        e32 = ((got.double() - ref).norm() / ref.norm()).item()
        stored = BF if mode != "fp16" else HF
        e16 = ((got.to(stored).double() - ref).norm() / ref.norm()).item()
        out["logits_rel_l2_vs_fp64"][mode] = {"logits_written_in_fp32": e32, "logits_as_stored_16bit": e16}
        print(mode, f"fp32-out {e32:.3e}  as stored {e16:.3e}", flush=True)
        del ref, got
    print(json.dumps(out))


if __name__ == "__main__":
    main()
