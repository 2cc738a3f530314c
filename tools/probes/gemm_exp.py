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
    from cogview_amd.csrc import build as B
    os.makedirs(OUT, exist_ok=True)
    others = [os.path.join(B.OBJ_DIR, os.path.basename(s)[:-4] + ".o") for s in B.sources() if not s.endswith("gemm.hip")]
    for m in MASKS:
        obj = os.path.join(OUT, f"gemm_{m}.o")
        subprocess.run([B._hipcc()] + B.FLAGS + [f"-DCOGV_EXP={m}", "-c", os.path.join(B.HERE, "gemm.hip"), "-o", obj], check=True,
                       capture_output=True)
        subprocess.run([B._hipcc(), "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", os.path.join(OUT, f"libexp_{m}.so"), obj] + others, check=True)
        os.remove(obj)
        print("built", m, flush=True)

def run_one():
    import torch
    from cogview_amd import ops
    from tools.microbench import timeit
    M = int(os.environ.get("EXP_M", 32640)); N = int(os.environ.get("EXP_N", 4096)); K = int(os.environ.get("EXP_K", 1024))
    pad = int(os.environ.get("EXP_PAD", 0))     # leading-dimension padding (elements): channel-camping probe
    x = torch.randn(M, K + pad, device="cuda", dtype=torch.bfloat16)[:, :K]
    w = (torch.randn(N, K + pad, device="cuda", dtype=torch.bfloat16) * 0.05)[:, :K]
    v = int(os.environ.get("EXP_VARIANT", 6))
    t = timeit(lambda: ops.gemm(x, w, variant=v), iters=10, warm=3)
    if int(os.environ["EXP_MASK"]) & 16:
        TM, TN = (256, 128) if v in (2, 3, 8) else (128, 128) if v == 4 else (256, 256)
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
