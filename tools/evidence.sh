#!/bin/bash
# One entry point for every kind of evidence file under profiles/ (GPU box; `gpurun -- 'bash tools/evidence.sh <what> ...'`).
# Everything is written under gpurun_out/ev/; copy what should be judged into profiles/rNN_<name>.
#
#   suite                     the whole GPU suite (--durations), smoke(), the driver's bench command
#                             -> profiles/rNN_gpu_tests_final_full_suite.log, rNN_smoke_final.log, rNN_bench_4B_b24_fp16_bf16leg_final.json
#   lines                     the other configurations' bench lines: 336M, VQ-VAE, cfg 1 (tiny-18M)
#                             -> profiles/rNN_bench_336M_*.json, rNN_bench_vqvae_*.json, rNN_bench_cfg1_*.json
#   kernel-stats [bench args] rocprofv3 --kernel-trace --stats of 5 steps of the bench command
#                             -> profiles/rNN_bench_4B_b24_fp16_kernel_stats.csv
#   traffic [bench args]      HBM-side traffic of the GEMM family: FETCH_SIZE / WRITE_SIZE in separate --pmc passes, gfx950 x2
#                             read correction (tools/collect_traffic.sh) -> profiles/rNN_gemm_hbm_traffic_pmc_4B_b24.json
#   mall                      Infinity-Cache hits vs HBM reads of the GEMM family from the L2's memory-side read latency
#                             (tools/mall_probe.py) -> profiles/rNN_gemm_mall_vs_hbm.txt
#   pmc-gemm                  counters of gemm_w4_kernel on the step's launches with their real epilogues (MFMA busy, clock,
#                             wave-cycle split) -> profiles/rNN_gemm_w4_pmc_mfma_busy_clock.txt
#   pmc-attn                  counters of the three dense attention kernels at the bench shape (stored keep bits): MFMA / VALU
#                             busy, waves per SIMD, VALU instructions per score, LDS conflicts -> profiles/rNN_attention_pmc.txt
#   bench-ab VAR v1 v2 [...]  alternating short bench runs (2 x each) with the environment variable VAR set to each value:
#                             tokens/s, ms/step and the per-family table -> profiles/rNN_<what>_ab.log
#                             (e.g. bench-ab COGV_WGRAD_QUEUE 1 0; bench-ab COGV_LN_BWD_LEAN 0 1; bench-ab COGVIEW_HIP_LIB "" build/ab/libcogview_x.so)
#   decode [batches]          captured 4B decode step, 1024-position memory (tools/mb_decode.py) -> profiles/rNN_decode_*.log
#   decode-stats [batch]      kernel statistics of the captured decode step -> profiles/rNN_decode_kernel_stats_b<batch>.csv
#   attn [lib ...]            the dense attention kernels as the train step runs them, per library (tools/mb_attn_train.py)
#   yardstick                 the vendor library on the 4B GEMM shapes with its kernel names (tools/probes/hipblaslt_names.py;
#                             a yardstick, not a dependency) -> profiles/rNN_hipblaslt_yardstick_4B_shapes.txt
# Probes added in round 6 (python tools/probes/<name>.py on the GPU box; each header says what it needs):
#   w4_dev.py build <tag> -DCOGV_EXP=<bits> / run <tag>...   probe builds of gemm_w4_kernel on the 4B / 336M shapes (epilogue by parts)
#   aux_l2_probe.py, tn_probe.py, group_m_sweep.sh           aux operand's memory side; TN vs NT / NN k-loops by parts; raster group sweep
#   attn_ts.py (build with -DCOGV_ATTN_TS), attn_layout_probe.py   where a dK.dV wave's life goes; fused-QKV rows vs head-major operands
#   ln_pair_bench.py, ln_pair_diag.py                        LN2' + LN3' in one pass against the two launches (time; bit differences)
# A/B builds: COGV_VARIANT=name COGV_HIPCC_EXTRA="-DX=1" python cogview_amd/csrc/build.py -> build/ab/libcogview_name.so
# (travels with the snapshot), selected at run time with COGVIEW_HIP_LIB.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; OUT=$R/gpurun_out/ev; mkdir -p $OUT
what=$1; shift
prof() { (cd /tmp; export TMPDIR=/tmp; "$@"); }
line() { python - "$1" <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f = d.get("roofline", {}).get("by_family", {})
print(round(d["value"]), d["unit"], round(d["ms_per_step"], 2), "ms/step", "end-to-end", round(d.get("mfma_roofline_frac_end_to_end", 0), 4),
      {k: (round(v["achieved"], 1), round(v["frac"], 3), round(v["share_of_step_time"], 4)) for k, v in f.items() if isinstance(v, dict) and "frac" in v})
P
}
case $what in
suite)
  ( time timeout 1500 python -m pytest tests -m gpu -q --durations=25 ) > $OUT/gpu_tests.log 2>&1; tail -40 $OUT/gpu_tests.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.log
  ( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
  line $OUT/bench_default.json; grep -E "logits rel-L2|real" $OUT/bench_default.err ;;
lines)
  for c in cogview-small-336M vqvae cogview-tiny-18M; do
    timeout 900 python bench.py --config $c --steps 20 --warmup 5 > $OUT/bench_$c.json 2> $OUT/bench_$c.err; echo "== $c"; line $OUT/bench_$c.json
  done ;;
kernel-stats)
  rm -rf $OUT/kstats
  prof rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kstats -- python $R/bench.py --steps 5 --warmup 2 --no-second-dtype --no-cpu-baseline --no-kernel-timing "$@" > $OUT/kstats.log 2>&1
  cp $(ls $OUT/kstats/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv; head -14 $OUT/kernel_stats.csv | cut -c1-170 ;;
traffic)
  bash tools/collect_traffic.sh "$@" | tail -5; cp gpurun_out/gemm_traffic.json $OUT/gemm_hbm_traffic_pmc.json ;;
mall)
  # Infinity-Cache (MALL) hits vs HBM reads of the GEMM family, from the L2's memory-side read latency (tools/mall_probe.py)
  rm -rf $OUT/mall
  prof rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_sum --kernel-trace --output-format csv -d $OUT/mall -- python $R/tools/mall_probe.py run > $OUT/mall.log 2>&1
  python tools/mall_probe.py report $OUT/mall | tee $OUT/gemm_mall_vs_hbm.txt ;;
pmc-gemm)
  rm -rf $OUT/pmc_gemm_a $OUT/pmc_gemm_b
  prof rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_gemm_a -- python $R/tools/mb_gemm_ab.py pmc > $OUT/pmc_gemm_a.log 2>&1
  prof rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc_gemm_b -- python $R/tools/mb_gemm_ab.py pmc > $OUT/pmc_gemm_b.log 2>&1
  python tools/pmc_gemm_report.py $OUT/pmc_gemm_a $OUT/pmc_gemm_b | tee $OUT/gemm_w4_pmc.txt ;;
pmc-attn)
  rm -rf $OUT/pmc_attn_a $OUT/pmc_attn_b $OUT/pmc_attn_c
  export PMC_ATTN_DTYPE=${PMC_ATTN_DTYPE:-fp16}
  prof rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/pmc_attn_a -- python $R/tools/pmc_attn.py > $OUT/pmc_attn_a.log 2>&1
  prof rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/pmc_attn_b -- python $R/tools/pmc_attn.py > $OUT/pmc_attn_b.log 2>&1
  prof rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_attn_c -- python $R/tools/pmc_attn.py > $OUT/pmc_attn_c.log 2>&1
  # executed score elements: 24 x 40 (batch x heads) x 153 visited 64 x 64 blocks x 4096
  ( python tools/pmc_kernels_report.py --match attn --scores 6.016e8 $OUT/pmc_attn_a $OUT/pmc_attn_b; python tools/pmc_report.py $OUT/pmc_attn_c attn ) | tee $OUT/attention_pmc.txt ;;
bench-ab)
  var=$1; shift
  for rep in 1 2; do for v in "$@"; do
    env "$var=$v" timeout 600 python bench.py --steps 12 --warmup 3 --no-second-dtype --no-cpu-baseline > $OUT/ab_tmp.json 2> $OUT/ab_tmp.err
    echo -n "$var=$v rep $rep: "; line $OUT/ab_tmp.json
  done; done | tee $OUT/bench_ab_$var.log
  grep -A40 "launches by kernel family" $OUT/ab_tmp.err | head -44 >> $OUT/bench_ab_$var.log ;;
decode)
  for b in ${@:-1 2 4 8}; do MB_DECODE_BATCH=$b MB_DECODE_GRAPH_ONLY=1 timeout 300 python tools/mb_decode.py 2>&1 | grep GraphDecoder; done | tee $OUT/decode.log ;;
decode-stats)
  b=${1:-1}; rm -rf $OUT/dstats
  MB_DECODE_BATCH=$b MB_DECODE_GRAPH_ONLY=1 prof rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dstats -- python $R/tools/mb_decode.py > $OUT/dstats.log 2>&1
  cp $(ls $OUT/dstats/*/*kernel_stats.csv | head -1) $OUT/decode_kernel_stats_b$b.csv; head -12 $OUT/decode_kernel_stats_b$b.csv | cut -c1-170 ;;
attn)
  for rep in 1 2; do for lib in "${@:-}"; do COGVIEW_HIP_LIB=$lib timeout 300 python tools/mb_attn_train.py 2>&1 | grep '"rep": [12]'; done; done | tee $OUT/attention_train_path.log ;;
yardstick)
  rm -rf $OUT/hbl
  prof rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/hbl -- python $R/tools/probes/hipblaslt_names.py > $OUT/hipblaslt.log 2>&1
  ( grep hipBLASLt $OUT/hipblaslt.log; python - <<P
import csv, glob
for f in glob.glob("$OUT/hbl/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "Cijk" in r["Name"]:
            print(r["Calls"], "calls", round(float(r["AverageNs"]) / 1e3, 1), "us avg:", r["Name"])
P
  ) | tee $OUT/hipblaslt_yardstick.txt ;;
*) sed -n 2,30p $0 ;;
esac
