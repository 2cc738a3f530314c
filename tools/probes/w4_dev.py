"""Development loop for the generation-4 GEMM kernel: build gemm.hip with -DCOGV_W4_DEV=<layout> (only that kernel,
bf16, one layout: 1 NT forward, 2 NN dgrad, 3 TN wgrad -- seconds to compile) into tools/probes/_exp/libw4_<layout>.so,
check it against torch.matmul and time it next to the production library's generation-3 kernel.
  build (CPU box):  python tools/probes/w4_dev.py build 1
  run   (GPU box):  python tools/probes/w4_dev.py run 1
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "probes", "_exp")
SHAPES = [(26112, 2560, 10240), (26112, 10240, 2560), (26112, 7680, 2560), (26112, 2560, 2560)]


def build(layout, extra=()):
    from cogview_amd.csrc import build as B
    os.makedirs(OUT, exist_ok=True)
    others = [os.path.join(B.OBJ_DIR, os.path.basename(s)[:-4] + ".o") for s in B.sources() if not s.endswith("gemm.hip")]
    obj = os.path.join(OUT, f"w4_{layout}.o")
    subprocess.run([B._hipcc()] + B.FLAGS + [f"-DCOGV_W4_DEV={layout}"] + list(extra) + ["-c", os.path.join(B.HERE, "gemm.hip"), "-o", obj], check=True)
    subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", os.path.join(OUT, f"libw4_{layout}.so"), obj] + others, check=True)
    os.remove(obj)
    print("built", layout, flush=True)


def operands(layout, M, N, K, gen):
    import torch
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16, generator=gen)
    b = torch.randn(N, K, device="cuda", dtype=torch.bfloat16, generator=gen) * 0.05
    if layout == 1:      # C = A B^T, both K-contiguous
        return dict(a=a, b=b), a, b
    if layout == 2:      # C = A Bs, Bs [K, N] (dgrad: dY [M,K'] x W [K',N])
        bs = b.t().contiguous()
        return dict(a=a, b=bs, trans_b=True), a, b
    at = a.t().contiguous()   # wgrad: A stored [K, M], B stored [K, N]
    bs = b.t().contiguous()
    return dict(a=at, b=bs, trans_a=True, trans_b=True), a, b


def run_one(layout, variant):
    import torch
    from cogview_amd import ops
    from tools.microbench import timeit
    gen = torch.Generator(device="cuda").manual_seed(1)
    if variant == 10:       # correctness on a small ragged-free case and one real shape
        for (M, N, K) in [(512, 768, 256), (1024, 512, 1024), (26112, 2560, 2560)]:
            kw, a, b = operands(layout, M, N, K, gen)
            y = ops.gemm(kw["a"], kw["b"], trans_a=kw.get("trans_a", False), trans_b=kw.get("trans_b", False), variant=10, splitk=1)
            ref = a.float() @ b.float().t()
            err = ((y.float() - ref).norm() / ref.norm()).item()
            print(f"   check {M}x{N}x{K}: rel-L2 {err:.2e}", flush=True)
            assert err < 5e-3, err
    for (M, N, K) in SHAPES:
        if layout == 3:
            M, N, K = K, N, M     # weight gradient: the token dimension is the contraction
        kw, a, b = operands(layout, M, N, K, gen)
        f = lambda: ops.gemm(kw["a"], kw["b"], trans_a=kw.get("trans_a", False), trans_b=kw.get("trans_b", False), variant=variant, splitk=1)
        t = min(timeit(f, iters=10, warm=3) for _ in range(2))
        print(f"v{variant} layout {layout} {M}x{N}x{K}: {t*1e6:8.1f} us  {2.0*M*N*K/t/1e12:7.1f} TF", flush=True)


if __name__ == "__main__":
    layout = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if sys.argv[1] == "build":
        build(layout, sys.argv[3:])
    elif sys.argv[1] == "run":
        subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(layout), "9"])
        env = dict(os.environ, COGVIEW_HIP_LIB=os.path.join(OUT, f"libw4_{layout}.so"))
        subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(layout), "10"], env=env)
    else:
        run_one(layout, int(sys.argv[3]))
