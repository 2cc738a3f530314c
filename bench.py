#!/usr/bin/env python
"""bench.py -- train tokens/sec of the CogView GPT hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full training step of pretrain_gpt2.py's loop on one synthetic batch already resident in
HBM: forward (GPT2Model) + fused cross entropy + backward + data-parallel gradient all-reduce + overflow check /
global-norm clip / AdamW + 16-bit parameter write.  Default workload: the configuration BASELINE.json's metric is
quoted on -- the 4B CogView-base GPT (48 layers / 2560 hidden / 40 heads; 16 B/param of weights, master copy,
Adam moments and gradients = 64 GB, so the whole model fits one 288-GB MI355X and every rank is a full data-parallel
replica, BASELINE.json configs[3]) -- rows of 1089 random tokens (1088 model positions), vocab 58240, weak scaling
(per-GPU micro-batch fixed).  `--config cogview-small-336M` runs configs[1].  Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

CONFIGS = {
    # name: (layers, hidden, heads)           BASELINE.json configs[1] / configs[3]
    "cogview-small-336M": (24, 1024, 16),
    "cogview-base-4B": (48, 2560, 40),
}
# per-GPU micro-batch (sequences): b x 1088 rows must fill whole rounds of 256-row GEMM tiles on 256 CUs.
#   336M: 30 x 1088 = 127.5 -> 128 row tiles;  4B: 24 x 1088 = 102 row tiles exactly, and 102 x {10, 30, 40}
#   column tiles of 256 are 3.98 / 11.95 / 15.94 rounds (activations 129 GB + 64 GB of model state < 288 GB)
DEFAULT_BATCH = {"cogview-small-336M": 30, "cogview-base-4B": 24}
METRIC = {"cogview-base-4B": "train tokens/sec/node (seq1089, 4B GPT) at 1/2/4/8 MI355X; % MFMA roofline",
          "cogview-small-336M": "train tokens/sec/node (seq1089, 336M GPT) at 1/2/4/8 MI355X; % MFMA roofline"}
VOCAB = 58240            # 58219 tokens padded to a multiple of 128 (arguments.py --make-vocab-size-divisible-by)
N_TOKEN_IDS = 58219
ROW = 1089               # tokens per data row; the model sees ROW-1 = 1088 positions (pretrain_gpt2.py:273-275)
PEAK_MFMA_TFLOPS = 2500.0   # MI355X dense bf16/fp16 MFMA peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def flops_per_token(L, h, V, s=ROW - 1):
    """fwd+bwd, full (non-causal-discounted) attention, no recompute credit."""
    return 3.0 * (L * (24.0 * h * h + 4.0 * s * h) + 2.0 * h * V)


def gemm_flops(M, N, K):
    return 2.0 * M * N * K


def cpu_baseline(L, h, heads, sample_layers=4):
    """The CPU oracle (oracle/cogview_oracle.py, fp32, torch CPU threads) timed on a bounded sample of the same
    workload: ONE 1088-token sequence through the embedding, `sample_layers` of the L layers, the tied LM head and
    the cross entropy, forward + backward; the layer part is scaled by L / sample_layers."""
    from oracle import cogview_oracle as O
    torch.manual_seed(0)
    s = ROW - 1
    g = torch.Generator().manual_seed(1)
    p = {"word_embeddings.weight": torch.randn(VOCAB, h, generator=g) * 0.02,
         "transformer.position_embeddings.weight": torch.randn(ROW, h, generator=g) * 0.02,
         "transformer.final_layernorm.weight": torch.ones(h), "transformer.final_layernorm.bias": torch.zeros(h)}
    for l in range(sample_layers):
        pre = f"transformer.layers.{l}."
        for ln in ("input_layernorm", "post_attention_layernorm", "third_layernorm", "fourth_layernorm"):
            p[pre + ln + ".weight"], p[pre + ln + ".bias"] = torch.ones(h), torch.zeros(h)
        for name, (o, i) in {"attention.query_key_value": (3 * h, h), "attention.dense": (h, h),
                             "mlp.dense_h_to_4h": (4 * h, h), "mlp.dense_4h_to_h": (h, 4 * h)}.items():
            p[pre + name + ".weight"] = torch.randn(o, i, generator=g) * 0.02
            p[pre + name + ".bias"] = torch.zeros(o)
    for t in p.values():
        t.requires_grad_(True)
    ids = torch.randint(0, N_TOKEN_IDS, (1, ROW), generator=g)
    tokens, labels = ids[:, :-1], ids[:, 1:]
    pos = torch.arange(s).unsqueeze(0)
    mask = O.build_mask(s, s)

    def run(n_layers):
        t0 = time.perf_counter()
        logits = O.gpt2_forward(tokens, pos, mask, p, n_layers, heads)
        loss = O.lm_loss(logits, labels, torch.ones(1, s))
        loss.backward()
        return time.perf_counter() - t0

    run(0)                                  # warm-up (head only)
    t_head = run(0)
    t_full = run(sample_layers)
    t_layers = max(t_full - t_head, 1e-9) * (L / sample_layers)
    return {"value": s / (t_head + t_layers), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 sequence x 1088 positions, fp32 oracle fwd+bwd: embedding + {sample_layers} of {L} layers "
                      f"(scaled x{L / sample_layers:g}) + tied LM head + CE; head {t_head:.2f}s, "
                      f"{sample_layers} layers {t_full - t_head:.2f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cogview-base-4B", choices=list(CONFIGS))
    ap.add_argument("--batch", type=int, default=0,
                    help="micro-batch per GPU (sequences of 1089 tokens); default per config (DEFAULT_BATCH)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16"])
    ap.add_argument("--dropout", type=float, default=0.1, help="reference default (arguments.py:30,40)")
    ap.add_argument("--checkpoint-activations", action="store_true",
                    help="recompute each layer in backward (the reference's scripts do; 288 GB HBM makes it unnecessary)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()
    if args.batch <= 0:
        args.batch = DEFAULT_BATCH[args.config]

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # COGV_BENCH_ONE_DEVICE=1 (development only): all ranks share cuda:0 and talk over gloo -- exercises the N > 1
    # control flow of this script on a one-GPU box; RCCL refuses two ranks on one device.  Never set by the driver.
    one_dev = os.environ.get("COGV_BENCH_ONE_DEVICE") == "1"
    torch.cuda.set_device(0 if one_dev else local_rank)
    import torch.distributed as dist
    if not dist.is_initialized():
        if world > 1:
            dist.init_process_group("gloo" if one_dev else "nccl", init_method="env://", world_size=world, rank=rank)
        else:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)

    from cogview_amd import mpu, ops, training
    from cogview_amd.fp16 import FP16_Module, FP16_Optimizer
    from cogview_amd.model import GPT2Model, PyTorchDistributedDataParallel, gpt2_get_params_for_weight_decay_optimization
    from cogview_amd.optim import FusedAdam

    mpu.initialize_model_parallel(1)
    torch.manual_seed(1234)
    mpu.model_parallel_cuda_manual_seed(1234)
    L, h, heads = CONFIGS[args.config]
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    t0 = time.perf_counter()
    model = GPT2Model(L, VOCAB, h, heads, args.dropout, args.dropout, args.dropout, ROW, 0, args.checkpoint_activations)
    n_params = sum(p.numel() for p in model.parameters())
    model = FP16_Module(model.cuda(), dtype=dtype, keep_half_outputs=True)
    ddp = None
    if world > 1:
        model = ddp = PyTorchDistributedDataParallel(model, process_group=mpu.get_data_parallel_group())
    inner = model
    while hasattr(inner, "module"):
        inner = inner.module
    groups = gpt2_get_params_for_weight_decay_optimization(inner)
    for grp in groups:
        for p in grp["params"]:
            if not hasattr(p, "model_parallel"):
                p.model_parallel = False
    opt = FP16_Optimizer(FusedAdam(groups, lr=1.5e-4, weight_decay=0.01), dynamic_loss_scale=True,
                         dynamic_loss_args={"init_scale": 2 ** 16 if dtype == torch.float16 else 1.0,
                                            "scale_window": 1000, "min_scale": 1, "delayed_shift": 2})
    assert opt._arena is not None
    model.train()
    log(f"[bench] {args.config}: {n_params / 1e6:.1f}M params, built in {time.perf_counter() - t0:.1f}s, "
        f"rank {rank}/{world}, micro-batch {args.batch}, dtype {args.dtype}, dropout {args.dropout}, "
        f"recompute {args.checkpoint_activations}")

    gen = torch.Generator().manual_seed(1234 + mpu.get_data_parallel_rank())
    text = torch.randint(0, N_TOKEN_IDS, (args.batch, ROW), generator=gen).cuda()      # resident in HBM
    loss_mask = torch.ones(args.batch, ROW, device="cuda")
    batch = training.get_batch(text, loss_mask)

    def step():
        return training.train_step(batch, model, opt, clip_grad=1.0, log=False, world_size=world)

    for _ in range(args.warmup):
        loss, skipped = step()
    # ---- timed region: barrier + synchronize on both sides, exactly K steps
    timing = None
    if not args.no_kernel_timing:
        timing = ops.enable_gemm_timing()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, skipped = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    gemm_stats = ops.collect_gemm_timing() if timing is not None else None
    t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = t.item()
    final_loss = loss.item()
    assert final_loss == final_loss, "loss is NaN"

    tokens_per_step = world * args.batch * (ROW - 1)
    value = tokens_per_step * args.steps / elapsed
    fpt = flops_per_token(L, h, VOCAB)
    out = {
        "metric": METRIC[args.config], "value": value, "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.config} ({L}L/{h}h/{heads} heads, {n_params / 1e6:.1f}M params), rows of 1089 "
                               f"random token ids -> 1088 model positions, vocab {VOCAB}, full train step "
                               f"(fwd+CE+bwd+grad all-reduce+clip+AdamW)",
                   "global_batch": world * args.batch, "seq_len": ROW, "model_positions": ROW - 1,
                   "parallelism": f"dp{world}", "dropout": args.dropout,
                   "activation_recompute": bool(args.checkpoint_activations), "loss": final_loss,
                   "loss_scale": opt.loss_scale},
        "model_tflops_per_gpu": value / world * fpt / 1e12,
        "mfma_roofline_frac_end_to_end": value / world * fpt / 1e12 / PEAK_MFMA_TFLOPS,
    }
    if gemm_stats is not None:
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_w4_kernel<%s> (256x256x64 tiles, 4 waves of 128x128, persistent work queues, 16x16x32 MFMA; NT fwd, "
                                                       "NN dgrad, TN wgrad grouped four per launch)" % args.dtype,
                           "achieved": gemm_stats["tflops"], "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": gemm_stats["tflops"] / PEAK_MFMA_TFLOPS, "traffic": None,
                           "launches": gemm_stats["launches"], "avg_launch_ms": gemm_stats["avg_ms"],
                           "share_of_step_time": gemm_stats["total_ms"] / (elapsed * 1e3),
                           "by_variant_tflops": gemm_stats["by_variant"]}
        log("[bench] GEMM launches by shape (TFLOP/s, launches, avg ms):")
        for k, v in gemm_stats["by_shape"].items():
            log(f"    {k:44s} {v['tflops']:8.1f} {v['launches']:6d} {v['avg_ms']:9.4f}")
    if gemm_stats is not None:
        # HBM-side traffic of the dominant kernel: measured off-line with rocprofv3 PMC passes (tools/collect_traffic.sh,
        # FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 x2 read correction) for the DEFAULT workload only.
        tpath = os.path.join(ROOT, "profiles", "r01_gemm_hbm_traffic_pmc_%s_b%d.json" % (args.config.split("-")[-1], args.batch))
        if os.path.exists(tpath):
            t = json.load(open(tpath))
            algo = gemm_stats["algo_bytes"] / max(gemm_stats["launches"], 1)
            out["roofline"]["traffic"] = t["gemm_hbm_bytes_per_launch_corrected"]
            out["roofline"]["traffic_unit"] = "bytes per launch (L2-miss side: FETCH_SIZE*2 + WRITE_SIZE, includes Infinity-Cache hits)"
            out["roofline"]["algorithmic_bytes_per_launch"] = algo
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(L, h, heads)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
