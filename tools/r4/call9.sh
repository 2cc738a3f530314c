#!/bin/bash
# round 4, GPU call 9: NaN-guard test, 336M / VQ-VAE / tiny bench lines, 4B fp16 line with the deferred guard + NT stores
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "nan_guard or train_steps or golden" 2>&1 | tail -2
timeout 600 python bench.py --config cogview-small-336M --steps 20 --warmup 3 > gpurun_out/r4/c9_336m.json 2> gpurun_out/r4/c9_336m.err
timeout 600 python bench.py --config vqvae --steps 8 --warmup 2 > gpurun_out/r4/c9_vqvae.json 2> gpurun_out/r4/c9_vqvae.err
timeout 600 python bench.py --config cogview-tiny-18M --steps 20 --warmup 5 > gpurun_out/r4/c9_tiny.json 2> gpurun_out/r4/c9_tiny.err
timeout 600 python bench.py --dtype fp16 --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/r4/c9_4b.json 2> gpurun_out/r4/c9_4b.err
python - <<'PY'
import json
for f in ("c9_336m", "c9_vqvae", "c9_tiny", "c9_4b"):
    try:
        d = json.loads(open(f"gpurun_out/r4/{f}.json").read().strip().splitlines()[-1])
        r = d.get("roofline", {})
        print(f, d["dtype"], round(d["value"], 1), d["unit"], round(d["ms_per_step"], 2), "ms", round(d["mfma_roofline_frac_end_to_end"], 4),
              "| dominant", round(r.get("achieved", 0), 1), round(r.get("frac", 0), 3), "|", {k: (round(v["value"], 1), round(v["mfma_roofline_frac_end_to_end"], 4)) for k, v in d.items() if k.endswith("_leg")},
              d["config"].get("logits_rel_l2_vs_fp32_reference", {}).get("measured"), (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/r4/c9_tiny.err
