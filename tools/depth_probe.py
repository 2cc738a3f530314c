"""Per-layer error growth of the HIP forward against the fp32 CPU oracle (same storage-rounded weights).
    python tools/depth_probe.py [336M|4B] [fp16|bf16] ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import depth_check as D

CFG = {"336M": (24, 1024, 16), "4B": (48, 2560, 40)}


def build(cfg, dtype, perturb=True, seed=1234):
    from cogview_amd.fp16 import FP16_Module
    from cogview_amd.model import GPT2Model
    L, h, heads = CFG[cfg]
    torch.manual_seed(seed)
    m = GPT2Model(L, 58240, h, heads, 0.1, 0.1, 0.1, 1089, 0, False)
    if perturb:
        with torch.no_grad():
            for n, p in m.named_parameters():
                if p.dim() == 1:
                    p.add_(0.05 * torch.randn_like(p))
    return FP16_Module(m.cuda(), dtype=dtype, keep_half_outputs=True), L, heads


if __name__ == "__main__":
    cfgs = [a for a in sys.argv[1:] if a in CFG] or ["336M"]
    dts = [a for a in sys.argv[1:] if a in ("fp16", "bf16")] or ["fp16", "bf16"]
    for cfg in cfgs:
        for dt in dts:
            t0 = time.perf_counter()
            model, L, heads = build(cfg, torch.float16 if dt == "fp16" else torch.bfloat16)
            g = torch.Generator().manual_seed(1234)
            ids = torch.randint(0, 58219, (1, 1088), generator=g).cuda()
            rep = D.depth_report(model.module, ids, L, heads)
            print(f"[{cfg} {dt}] logits rel-L2 {rep['logits']:.3e}; stream: " +
                  " ".join(f"L{n}={e:.2e}" for n, e in rep["stream"].items()) +
                  f"; oracle {rep['oracle_seconds']:.1f}s, total {time.perf_counter() - t0:.1f}s", flush=True)
            del model
            torch.cuda.empty_cache()
