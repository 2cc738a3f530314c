"""Schedule experiments on the ping-pong GEMM loop: build gemm.hip with -DCOGV_EXP=<mask> (bit 0 drops the DMA
issue, bit 1 the LDS fragment reads, bit 2 the MFMAs; results are garbage by design) and time each build.
  build (CPU box):  python tools/probes/gemm_exp.py build
  run   (GPU box):  python tools/probes/gemm_exp.py run
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "probes", "_exp")
MASKS = [int(m) for m in os.environ.get("EXP_MASKS", "0,1,2,3,4,5,6").split(",")]

def build():
    """gemm.hip itself (generations 1-3: EXP_VARIANT=9 probes the ping-pong kernel) and the bf16 NT unit of generation 4
    are rebuilt with the mask; the other objects come from the production build (run cogview_amd.csrc.build first)."""
    from cogview_amd.csrc import build as B
    os.makedirs(OUT, exist_ok=True)
    for m in MASKS:
        objs, tmp = [], []
        for src, obj, flags in B.units():
            if os.path.basename(obj) in ("gemm.o", "gemm_w4_0.o"):
                o2 = os.path.join(OUT, f"m{m}_" + os.path.basename(obj))
                subprocess.run([B._hipcc()] + B.FLAGS + flags + [f"-DCOGV_EXP={m}", "-c", src, "-o", o2], check=True, capture_output=True)
                objs.append(o2); tmp.append(o2)
            else:
                objs.append(obj)
        subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", os.path.join(OUT, f"libexp_{m}.so")] + objs, check=True)
        for o in tmp:
            os.remove(o)
        print("built", m, flush=True)

def run_one():
    import torch
    from cogview_amd import ops
    from tools.microbench import timeit
    M = int(os.environ.get("EXP_M", 32640)); N = int(os.environ.get("EXP_N", 4096)); K = int(os.environ.get("EXP_K", 1024))
    pad = int(os.environ.get("EXP_PAD", 0))     # leading-dimension padding (elements): channel-camping probe
    x = torch.randn(M, K + pad, device="cuda", dtype=torch.bfloat16)[:, :K]
    w = (torch.randn(N, K + pad, device="cuda", dtype=torch.bfloat16) * 0.05)[:, :K]
    v = int(os.environ.get("EXP_VARIANT", 10))
    t = timeit(lambda: ops.gemm(x, w, variant=v), iters=10, warm=3)
    if int(os.environ["EXP_MASK"]) & 16:
        TM, TN = (256, 128) if v == 3 else (128, 128) if v == 1 else (256, 256)
        y = ops.gemm(x, w, variant=v, out_dtype=torch.float32)
        torch.cuda.synchronize()
        print(f"   shader clock first/last workgroup: {y[0, 0].item():.0f} / {y[(M - 1) // TM * TM, (N - 1) // TN * TN].item():.0f} MHz")
    print(f"mask {os.environ['EXP_MASK']} v{v} {M}x{N}x{K} pad {pad}: {t*1e6:.1f} us  {2.0*M*N*K/t/1e12:.0f} TF-equivalent", flush=True)

if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        for m in MASKS:
            env = dict(os.environ, COGVIEW_HIP_LIB=os.path.join(OUT, f"libexp_{m}.so"), EXP_MASK=str(m))
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env)
    else:
        run_one()
