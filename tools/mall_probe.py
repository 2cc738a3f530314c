"""how much of the GEMM family's L2-miss-side traffic (roofline.traffic: 4.19 x the algorithmic bytes)
is served by the Infinity Cache (MALL, 256 MiB, memory side) and how much comes from HBM?  rocprofv3 on gfx950 lists no MALL
hit counter (profiles/r06_rocprofv3_counter_list_memory_side.txt); what it does list is the L2's memory-side read request
count and the sum of requests in flight per cycle:  average read latency = TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum
(cycles; the counter's own description).  An Infinity-Cache hit returns sooner than an HBM access, so three workloads are run
under the same counters:
    hbm     one pass over an 8-GiB buffer (nothing can hit: 32 x the Infinity Cache)          -> latency of an HBM read, loaded
    mall    40 passes over a 96-MiB buffer (3 x the aggregate L2, 0.375 x the Infinity Cache)   -> latency of an Infinity-Cache hit
    gemm    the 4B step's forward / dgrad launches with their epilogues (tools/mb_gemm_ab.py's shapes)
and the GEMM's hit fraction is read off the latency scale:  f = (L_hbm - L_gemm) / (L_hbm - L_mall)  (queueing differs between
a pure stream and a GEMM's bursty panels, so this is an estimate with that caveat, stated next to the number).
  GPU box:  bash tools/evidence.sh mall        (runs this under rocprofv3 --pmc and prints the report)
  python tools/mall_probe.py run | report <dir>
"""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    from cogview_amd import ops
    dev = "cuda"
    big = torch.empty(8 << 30, dtype=torch.uint8, device=dev).view(torch.float32)
    big.fill_(1.0)
    small = torch.ones(96 << 18, dtype=torch.float32, device=dev)          # 96 MiB
    torch.cuda.synchronize()
    for _ in range(3):
        big.sum()                                                          # "hbm": reduce kernels over 8 GiB
    for _ in range(40):
        small.sum()                                                        # "mall"
    torch.cuda.synchronize()
    g = torch.Generator(device=dev).manual_seed(1)
    dt = torch.float16
    rn = lambda *s: torch.randn(*s, device=dev, dtype=dt, generator=g)
    M, h = 26112, 2560
    x, x4 = rn(M, h), rn(M, 4 * h)
    w_qkv, w_d, w_1, w_2 = rn(3 * h, h) * 0.02, rn(h, h) * 0.02, rn(4 * h, h) * 0.02, rn(h, 4 * h) * 0.02
    b1, bh, b3 = rn(4 * h) * 0.02, rn(h) * 0.02, rn(3 * h) * 0.02
    aux = torch.empty(M, 4 * h, device=dev, dtype=dt)
    cs = torch.zeros(4 * h, device=dev, dtype=dt)
    amax = torch.zeros(1, device=dev, dtype=torch.float32)
    dy3 = rn(M, 3 * h)
    for _ in range(3):
        ops.gemm(x, w_qkv, bias=b3)
        ops.gemm(x, w_d, bias=bh, dropout=(0.1, 1, 2), absmax=amax)
        ops.gemm(x, w_1, bias=b1, gelu=True, gelu_daux=aux)
        ops.gemm(x4, w_2, bias=bh, dropout=(0.1, 1, 2), absmax=amax)
        ops.gemm(x, w_2, trans_b=True, mul_aux=aux, colsum_out=cs)
        ops.gemm(x4, w_1, trans_b=True)
        ops.gemm(x, w_d, trans_b=True)
        ops.gemm(dy3, w_qkv, trans_b=True)
    torch.cuda.synchronize()


def report(d):
    cc = glob.glob(d + "/*/*counter_collection.csv")[0]
    kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
    dur = {r["Dispatch_Id"]: (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt))}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cc)):
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    groups = collections.defaultdict(list)
    n_reduce = 0
    for did in sorted(per, key=lambda k: int(k)):
        ns, name = dur.get(did, (0.0, "?"))
        c = per[did]
        if "gemm_w4" in name:
            groups["gemm"].append((ns, c))
        elif "reduce_kernel" in name and c.get("TCC_EA0_RDREQ_sum", 0) > 1e5:
            # the first three large reductions are the 8-GiB passes, the following ones the 96-MiB passes
            n_reduce += 1
            groups["hbm" if ns > 5e5 else "mall"].append((ns, c))
    out = {}
    for k in ("hbm", "mall", "gemm"):
        v = groups.get(k, [])
        if k == "mall":
            v = v[5:]                                   # the first passes still fill the cache
        if not v:
            continue
        req = sum(c["TCC_EA0_RDREQ_sum"] for _, c in v)
        lvl = sum(c["TCC_EA0_RDREQ_LEVEL_sum"] for _, c in v)
        dram = sum(c.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) for _, c in v)
        t = sum(ns for ns, _ in v)
        out[k] = dict(dispatches=len(v), read_requests=req, avg_read_latency_cycles=lvl / max(req, 1), requests_destined_for_dram=dram,
                      requests_per_us=req / (t / 1e3))
        print(f"{k:5s} dispatches {len(v):4d}  memory-side read requests {req:.4g}  ({100 * dram / max(req, 1):.0f} % 'destined for DRAM (MC)')  "
              f"avg latency {lvl / max(req, 1):7.1f} cycles  {req / (t / 1e3):8.1f} requests/us")
    if all(k in out for k in ("hbm", "mall", "gemm")):
        lh, lm, lg = (out[k]["avg_read_latency_cycles"] for k in ("hbm", "mall", "gemm"))
        f = (lh - lg) / (lh - lm) if lh != lm else float("nan")
        print(f"GEMM family: Infinity-Cache hit fraction of its memory-side reads ~ {f:.2f}  (latency scale: HBM {lh:.0f}, Infinity Cache {lm:.0f}, "
              f"GEMM {lg:.0f} cycles; > 1 or < 0 means the scale does not transfer -- queueing differs between a stream and a GEMM)")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
