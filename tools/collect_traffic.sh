#!/bin/bash
# HBM traffic of the dominant kernel family (gemm_w4_kernel + gemm_pp64_kernel + gemm_glds_kernel) for the bench workload, per MI355X_MICROARCH.md:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slot limits), counters only with --kernel-trace.
# Units: KiB; gfx950 correction: FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads -> x2.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/traffic_$c -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing "$@" > /dev/null 2>&1
done
python - <<PY
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$R/gpurun_out/traffic_%s/*/*counter_collection.csv" % c)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            key = "gemm" if any(k in r["Kernel_Name"] for k in ("gemm_glds", "gemm_pp64", "gemm_w4")) else ("attn" if "attn" in r["Kernel_Name"] else "other")
            agg[key].append(float(r["Counter_Value"]))
    out[c] = {k: {"launches": len(v), "avg_KiB": sum(v) / len(v), "total_GiB": sum(v) / 1048576} for k, v in agg.items()}
g = out["FETCH_SIZE"]["gemm"]["avg_KiB"] * 2 * 1024 + out["WRITE_SIZE"]["gemm"]["avg_KiB"] * 1024
out["gemm_hbm_bytes_per_launch_corrected"] = g
out["note"] = "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB units"
json.dump(out, open("$R/gpurun_out/gemm_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
