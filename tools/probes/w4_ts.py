"""Where does an item (one 256x256 output tile) of the generation-4 GEMM spend its time?  GPU box, probe build:
    python tools/probes/w4_dev.py build ts -DCOGV_W4_TS          (CPU box)
    python tools/probes/w4_ts.py                                  (GPU box; loads tools/probes/_exp/libts.so)
The probe build stamps s_memrealtime at the phase boundaries of every item in every wave and writes the per-wave sums over
the head of C (cogview_amd/csrc/gemm.hip, W4_TS).  Printed: mean microseconds per item and phase over all waves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("COGVIEW_HIP_LIB", os.path.join(ROOT, "tools", "probes", "_exp", "libts.so"))
import torch
from cogview_amd import ops

PH = ["wait+barrier", "pre-step", "k-loop", "drain+handover", "next setup+prologue", "epilogue half 0", "epilogue half 1"]


def report(name, c, ncu=256):
    torch.cuda.synchronize()
    raw = c.view(torch.int32).flatten()[: ncu * 4 * 16].view(ncu * 4, 16).cpu().to(torch.float64)
    items = raw[:, 7]
    ok = items > 0
    per = raw[ok, :7] / items[ok, None] / 100.0            # 100 MHz ticks -> microseconds per item
    tot, clk = raw[ok, 8] / 100.0, raw[ok, 9] / raw[ok, 8] * 100.0 / 1e3
    print(f"{name}: {int(ok.sum())} waves, items/wave {items[ok].mean():.2f}, kernel {tot.mean():.1f} us, "
          f"s_memtime/s_memrealtime -> {clk.mean():.3f} GHz-equivalent", flush=True)
    for k, ph in enumerate(PH):
        print(f"    {ph:22s} {per[:, k].mean():7.2f} us  (min {per[:, k].min():6.2f}  max {per[:, k].max():6.2f})")
    print(f"    {'sum per item':22s} {per.sum(1).mean():7.2f} us")


def main():
    g = torch.Generator(device="cuda").manual_seed(1)
    dt = torch.bfloat16
    rn = lambda *s: torch.randn(*s, device="cuda", dtype=dt, generator=g)
    M, h = 26112, 2560
    x, x4 = rn(M, h), rn(M, 4 * h)
    w_qkv, w_d, w_1, w_2 = rn(3 * h, h) * 0.02, rn(h, h) * 0.02, rn(4 * h, h) * 0.02, rn(h, 4 * h) * 0.02
    b1, bh = rn(4 * h) * 0.02, rn(h) * 0.02
    aux = torch.empty(M, 4 * h, device="cuda", dtype=dt)
    cs = torch.zeros(4 * h, device="cuda", dtype=dt)
    amax = torch.zeros(1, device="cuda", dtype=torch.float32)
    cases = [
        ("fwd qkv bias (K=2560, N=7680)", lambda: ops.gemm(x, w_qkv, bias=torch.cat((bh, bh, bh)))),
        ("dgrad h<-h plain (K=2560, N=2560)", lambda: ops.gemm(x, w_d, trans_b=True)),
        ("fwd h->4h bias+gelu+daux (K=2560, N=10240)", lambda: ops.gemm(x, w_1, bias=b1, gelu=True, gelu_daux=aux)),
        ("fwd 4h->h bias+drop+amax (K=10240, N=2560)", lambda: ops.gemm(x4, w_2, bias=bh, dropout=(0.1, 1, 2), absmax=amax)),
        ("dgrad 4h<-h mulaux+colsum (K=2560, N=10240)", lambda: ops.gemm(x, w_2, trans_b=True, mul_aux=aux, colsum_out=cs)),
        ("dgrad h<-4h plain (K=10240, N=2560)", lambda: ops.gemm(x4, w_1, trans_b=True)),
    ]
    for name, f in cases:
        for _ in range(3):
            f()                                            # warm clocks
        report(name, f())


if __name__ == "__main__":
    main()
