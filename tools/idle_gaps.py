"""Where is the GPU idle inside a training step?  From a rocprofv3 --kernel-trace of bench.py: the gaps between consecutive
kernels of the last timed step (end of one to start of the next on the device clock), summed and listed by what follows."""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(({"s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"]), "n": r["Kernel_Name"]} for r in csv.DictReader(open(f))), key=lambda r: r["s"])
# steps end with the AdamW kernel: take the span between the last two
adam = [i for i, r in enumerate(rows) if "adamw_kernel" in r["n"]]
a, b = adam[-2] + 1, adam[-1] + 1
step = rows[a:b]
span = step[-1]["e"] - rows[a - 1]["e"]
busy = sum(r["e"] - r["s"] for r in step)
gaps = []
prev_e = rows[a - 1]["e"]
for r in step:
    g = r["s"] - prev_e
    if g > 0:
        gaps.append((g, r["n"]))
    prev_e = max(prev_e, r["e"])
short = lambda n: re.sub(r"_ZN12_GLOBAL__N_1\d+|\(anonymous namespace\)::|void ", "", n)[:70]
print(f"last step: {len(step)} kernels, span {span / 1e6:.2f} ms, kernel time {busy / 1e6:.2f} ms, idle {sum(g for g, _ in gaps) / 1e6:.3f} ms in {len(gaps)} gaps")
hist = collections.Counter()
for g, _ in gaps:
    hist["<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"] += g
print("idle by gap size (ms):", {k: round(v / 1e6, 3) for k, v in hist.items()})
by = collections.defaultdict(lambda: [0, 0])
for g, n in gaps:
    by[short(n)][0] += g; by[short(n)][1] += 1
for n, (g, c) in sorted(by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {g / 1e6:8.3f} ms  {c:5d} gaps  before {n}")
print("largest gaps:", [(round(g / 1e3, 1), short(n)[:40]) for g, n in sorted(gaps, reverse=True)[:8]])
