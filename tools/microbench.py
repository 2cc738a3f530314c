"""Kernel micro-benchmarks on the GPU box (HIP-event timing on the current stream).  Not part of bench.py."""
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from cogview_amd import ops


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    out = {}
    dt = torch.bfloat16
    b = int(os.environ.get("MB_BATCH", "16"))
    M = b * 1088
    for (h, tag) in ((1024, "S"), (2560, "B")):
        for (N, K, name) in ((3 * h, h, "qkv"), (h, h, "dense"), (4 * h, h, "h4h"), (h, 4 * h, "4hh")):
            x = torch.randn(M, K, device="cuda", dtype=dt)
            w = torch.randn(N, K, device="cuda", dtype=dt) * 0.02
            dy = torch.randn(M, N, device="cuda", dtype=dt)
            flops = 2.0 * M * N * K
            t = timeit(lambda: ops.gemm(x, w))
            out[f"{tag}.{name}.fwd_TF"] = flops / t / 1e12
            t = timeit(lambda: ops.gemm(dy, w, trans_b=True))
            out[f"{tag}.{name}.dgrad_TF"] = flops / t / 1e12
            t = timeit(lambda: ops.gemm(dy, x, trans_a=True, trans_b=True))
            out[f"{tag}.{name}.wgrad_TF"] = flops / t / 1e12
            t = timeit(lambda: torch.matmul(x, w.t()))
            out[f"{tag}.{name}.hipblaslt_fwd_TF"] = flops / t / 1e12
            print(json.dumps({k: round(v, 1) for k, v in out.items() if k.startswith(f"{tag}.{name}")}), flush=True)
        H = h // 64
        qkv = torch.randn(b, 1088, 3 * h, device="cuda", dtype=dt)
        q, k, v = [qkv[:, :, i * h:(i + 1) * h].view(b, 1088, H, 64) for i in range(3)]
        do = torch.randn(b, 1088, H, 64, device="cuda", dtype=dt)
        fl = 4.0 * b * H * 1088 * 1088 * 64          # full (non-causal) count, as the reference executes it
        for p in (0.0, 0.1):
            drop = None if p == 0 else (p, 1, 2)
            t = timeit(lambda: ops.attention_fwd(q, k, v, dropout=drop))
            out[f"{tag}.attn_fwd_p{p}_TF"] = fl / t / 1e12
            o, lse = ops.attention_fwd(q, k, v, dropout=drop)
            t = timeit(lambda: ops.attention_bwd(do, q, k, v, o, lse, dropout=drop))
            out[f"{tag}.attn_bwd_p{p}_TF"] = 2.5 * fl / t / 1e12
        print(json.dumps({k: round(v, 1) for k, v in out.items() if "attn" in k and k.startswith(tag)}), flush=True)
        x = torch.randn(M, h, device="cuda", dtype=dt)
        g_, b_ = torch.ones(h, device="cuda", dtype=dt), torch.zeros(h, device="cuda", dtype=dt)
        am = ops.absmax(x)
        t = timeit(lambda: ops.sandwich_ln_fwd(x, g_, b_, 1e-5, am))
        out[f"{tag}.ln_fwd_GBs"] = 2 * M * h * 2 / t / 1e9
        y, mean, rstd = ops.sandwich_ln_fwd(x, g_, b_, 1e-5, am)
        dg, db = torch.zeros_like(g_), torch.zeros_like(b_)
        t = timeit(lambda: ops.sandwich_ln_bwd(x, x, g_, mean, rstd, dgamma=dg, dbeta=db))
        out[f"{tag}.ln_bwd_GBs"] = 3 * M * h * 2 / t / 1e9
        print(json.dumps({k: round(v, 1) for k, v in out.items() if ".ln_" in k and k.startswith(tag)}), flush=True)
    V = 58240
    Mc = 8 * 1088
    lg = torch.randn(Mc, V, device="cuda", dtype=dt)
    tg = torch.randint(0, V, (Mc,), device="cuda")
    t = timeit(lambda: ops.ce_fwd(lg, tg, 0), iters=5)
    out["ce_fwd_GBs"] = Mc * V * 2 / t / 1e9
    rm, se, pr, ls = ops.ce_fwd(lg, tg, 0)
    gr = torch.ones(Mc, device="cuda")
    t = timeit(lambda: ops.ce_bwd(lg, tg, 0, rm, se, gr, out=lg), iters=5)
    out["ce_bwd_GBs"] = 2 * Mc * V * 2 / t / 1e9
    print(json.dumps({k: round(v, 1) for k, v in out.items() if k.startswith("ce")}), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/microbench.json", "w"), indent=1)


if __name__ == "__main__" and not os.environ.get("MB_VQVAE"):
    main()


def vqvae_bench(batch=32):
    from cogview_amd import vqvae
    torch.manual_seed(0)
    m = vqvae.new_model().eval().cuda()
    img = torch.randn(batch, 3, 256, 256, device="cuda")
    t = timeit(lambda: vqvae.img2code(m, img), iters=3, warm=1)
    ids = vqvae.img2code(m, img)
    t2 = timeit(lambda: vqvae.code2img(m, ids.view(batch, 32, 32)), iters=3, warm=1)
    res = {"vqvae.batch": batch, "vqvae.encode_img_per_s": batch / t, "vqvae.encode_TF": 48.32e9 * batch / t / 1e12,
           "vqvae.decode_img_per_s": batch / t2, "vqvae.decode_TF": 176.29e9 * batch / t2 / 1e12}
    print(json.dumps({k: round(v, 2) for k, v in res.items()}), flush=True)
    return res


if __name__ == "__main__" and os.environ.get("MB_VQVAE"):
    vqvae_bench(int(os.environ.get("MB_VQVAE")))
