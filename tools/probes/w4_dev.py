"""Probe builds of the generation-4 GEMM kernel: compile the bf16 units (gemm.hip -DCOGV_W4_TU=0..2) with extra flags
(e.g. -DCOGV_EXP=64: no epilogue) into tools/probes/_exp/lib<tag>.so next to the production objects, then time the
4B model's GEMMs (with their real epilogues) against the production library.
  build (CPU box):  python tools/probes/w4_dev.py build <tag> [flags...]
  run   (GPU box):  python tools/probes/w4_dev.py run <tag> [<tag> ...]      ("prod" = the production library)
"""
import concurrent.futures, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "probes", "_exp")


def build(tag, extra):
    from cogview_amd.csrc import build as B
    os.makedirs(OUT, exist_ok=True)
    objs, jobs = [], []
    for src, obj, flags in B.units():
        name = os.path.basename(obj)
        if name in ("gemm_w4_0.o", "gemm_w4_1.o", "gemm_w4_2.o"):
            o2 = os.path.join(OUT, f"{tag}_{name}")
            jobs.append([B._hipcc()] + B.FLAGS + flags + list(extra) + ["-c", src, "-o", o2])
            objs.append(o2)
        else:
            objs.append(obj)
    with concurrent.futures.ThreadPoolExecutor(max_workers=3) as ex:
        list(ex.map(lambda c: subprocess.run(c, check=True, capture_output=True), jobs))
    subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", os.path.join(OUT, f"lib{tag}.so")] + objs, check=True)
    for o in objs:
        if o.startswith(OUT):
            os.remove(o)
    print("built", tag, flush=True)


def run_one(tag):
    import torch
    from cogview_amd import ops
    from tools.microbench import timeit
    g = torch.Generator(device="cuda").manual_seed(1)
    M, h = 26112, 2560
    dt = torch.bfloat16
    rn = lambda *s: torch.randn(*s, device="cuda", dtype=dt, generator=g)
    x, x4 = rn(M, h), rn(M, 4 * h)
    w_qkv, w_d, w_1, w_2 = rn(3 * h, h) * 0.02, rn(h, h) * 0.02, rn(4 * h, h) * 0.02, rn(h, 4 * h) * 0.02
    b1, bh = rn(4 * h) * 0.02, rn(h) * 0.02
    aux = torch.empty(M, 4 * h, device="cuda", dtype=dt)
    cs = torch.zeros(4 * h, device="cuda", dtype=dt)
    amax = torch.zeros(1, device="cuda", dtype=torch.float32)
    cases = [
        ("fwd  h->4h  bias+gelu+daux (131)", 2.0 * M * 4 * h * h, lambda: ops.gemm(x, w_1, bias=b1, gelu=True, gelu_daux=aux)),
        ("fwd  4h->h  bias+drop+amax (25)", 2.0 * M * 4 * h * h, lambda: ops.gemm(x4, w_2, bias=bh, dropout=(0.1, 1, 2), absmax=amax)),
        ("fwd  h->h   bias+drop+amax (25)", 2.0 * M * h * h, lambda: ops.gemm(x, w_d, bias=bh, dropout=(0.1, 1, 2), absmax=amax)),
        ("fwd  qkv    bias (1)", 2.0 * M * 3 * h * h, lambda: ops.gemm(x, w_qkv, bias=torch.cat((bh, bh, bh)))),
        ("dgrad 4h<-h mulaux+colsum (320)", 2.0 * M * 4 * h * h, lambda: ops.gemm(x, w_2, trans_b=True, mul_aux=aux, colsum_out=cs)),
        ("dgrad h<-4h plain (0)", 2.0 * M * 4 * h * h, lambda: ops.gemm(x4, w_1, trans_b=True)),
        ("dgrad h<-h  plain (0)", 2.0 * M * h * h, lambda: ops.gemm(x, w_d, trans_b=True)),
    ]
    M2, h2 = 32640, 1024                                      # cogview-small-336M at b = 30 (K = 1024: 16 k-tiles per item)
    y2, w2q, w21, b2q, b21 = rn(M2, h2), rn(3 * h2, h2) * 0.02, rn(4 * h2, h2) * 0.02, rn(3 * h2) * 0.02, rn(4 * h2) * 0.02
    aux2 = torch.empty(M2, 4 * h2, device="cuda", dtype=dt)
    cases += [
        ("336M fwd qkv  bias (1)", 2.0 * M2 * 3 * h2 * h2, lambda: ops.gemm(y2, w2q, bias=b2q)),
        ("336M fwd h->4h bias+gelu+daux", 2.0 * M2 * 4 * h2 * h2, lambda: ops.gemm(y2, w21, bias=b21, gelu=True, gelu_daux=aux2)),
        ("336M dgrad h<-h plain (0)", 2.0 * M2 * h2 * h2, lambda: ops.gemm(y2, w2q[:h2].contiguous(), trans_b=True)),
    ]
    for nm, y, ref in (("NT", ops.gemm(x, w_d), x.float() @ w_d.float().t()), ("NN", ops.gemm(x, w_d, trans_b=True), x.float() @ w_d.float()),
                       ("TN", ops.gemm(x, x4[:, :h].contiguous(), trans_a=True, trans_b=True, splitk=1), x.float().t() @ x4[:, :h].float())):
        print(f"[{tag:10s}] check {nm}: rel-L2 {((y.float() - ref).norm() / ref.norm()).item():.2e}", flush=True)
    for name, fl, f in cases:
        t = min(timeit(f, iters=8, warm=2) for _ in range(2))
        print(f"[{tag:10s}] {name:34s} {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    elif sys.argv[1] == "run":
        for tag in sys.argv[2:]:
            env = dict(os.environ)
            if os.path.exists(tag):                       # a library path (e.g. build/ab/libcogview_<x>.so)
                env["COGVIEW_HIP_LIB"] = os.path.abspath(tag)
            elif tag != "prod":
                env["COGVIEW_HIP_LIB"] = os.path.join(OUT, f"lib{tag}.so")
            subprocess.run([sys.executable, os.path.abspath(__file__), "one", tag], env=env)
    else:
        run_one(sys.argv[2])
