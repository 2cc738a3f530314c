"""Where does the weight-gradient (TN) k-loop lose against the forward (NT) one?  The grouped 4B weight gradient and the K = 10240
forward under probe builds of gemm_w4_kernel (tools/probes/w4_dev.py build <tag> -DCOGV_EXP=<bits>): 1 no DMA, 2 no LDS reads,
3 neither (MFMA only), 2048 no epilogue.   python tools/probes/tn_probe.py <tag> ...   (tags as w4_dev.py run)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def one(tag):
    import torch
    from cogview_amd import ops
    from tools.microbench import timeit
    g = torch.Generator(device="cuda").manual_seed(1)
    M, h = 32640, 2560
    dt = torch.bfloat16
    rn = lambda *s: torch.randn(*s, device="cuda", dtype=dt, generator=g)
    x, x4, dy3, dy4 = rn(M, h), rn(M, 4 * h), rn(M, 3 * h), rn(M, 4 * h)
    w_2, w_1 = rn(h, 4 * h) * 0.02, rn(4 * h, h) * 0.02
    gw = [torch.zeros(n, k, device="cuda", dtype=dt) for n, k in ((3 * h, h), (h, h), (4 * h, h), (h, 4 * h))]
    cases = [
        ("wgrad grouped x4 (TN, K = 32640)", 2.0 * M * 12 * h * h, lambda: ops.gemm_grouped(
            [(dy3, x, gw[0]), (x, x, gw[1]), (dy4, x, gw[2]), (x, x4, gw[3])], accumulate=True)),
        ("fwd 4h->h plain (NT, K = 10240)", 2.0 * M * 4 * h * h, lambda: ops.gemm(x4, w_2)),
        ("dgrad h<-4h plain (NN, K = 10240)", 2.0 * M * 4 * h * h, lambda: ops.gemm(x4, w_1, trans_b=True)),
    ]
    for name, fl, f in cases:
        t = min(timeit(f, iters=6, warm=2) for _ in range(2))
        print(f"[{tag:10s}] {name:36s} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "one":
        one(sys.argv[2])
    else:
        for tag in sys.argv[1:]:
            env = dict(os.environ)
            if tag != "prod":
                env["COGVIEW_HIP_LIB"] = os.path.join(ROOT, "tools", "probes", "_exp", f"lib{tag}.so")
            subprocess.run([sys.executable, os.path.abspath(__file__), "one", tag], env=env)
