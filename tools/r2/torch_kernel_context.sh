#!/bin/bash
# Which torch element-wise kernels (fills, dtype copies) run inside the 4B train step, and between which of our kernels:
# rocprofv3 kernel trace of two bench steps, reduced on the box to a (previous kernel, torch kernel, next kernel) histogram.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r2ctx; mkdir -p $O
(cd /tmp; export TMPDIR=/tmp; rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python $R/bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-second-dtype --no-kernel-timing > $O/prof.log 2>&1)
python - <<PY
import csv, glob, collections, re
f = glob.glob("$O/prof/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"(\w+_kernel)", n)
    if "at::native" in n:
        k = re.search(r"native::(\w+)", n); fn = re.search(r"(\w+Functor|\w+_kernel_cuda|\w+Op)", n)
        return "TORCH:" + (fn.group(1) if fn else (k.group(1) if k else n[:40]))
    return m.group(1) if m else n[:40]
names = [short(r["Kernel_Name"]) for r in rows]
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
half = len(rows) // 3 * 2            # look at the last third (a steady-state step)
hist = collections.Counter(); tim = collections.Counter()
for i in range(half, len(rows) - 1):
    if names[i].startswith("TORCH") or "copyBuffer" in names[i] or "fillBuffer" in names[i]:
        key = (names[i - 1], names[i], names[i + 1]); hist[key] += 1; tim[key] += dur[i]
with open("$O/context.txt", "w") as out:
    for key, n in sorted(hist.items(), key=lambda kv: -tim[kv[0]]):
        out.write(f"{n:5d} x {tim[key] / n / 1e3:8.1f} us  total {tim[key] / 1e6:7.3f} ms   {key[0]}  ->  {key[1]}  ->  {key[2]}\n")
print(open("$O/context.txt").read()[:6000])
PY
rm -rf $O/prof
