#!/usr/bin/env python
"""bench.py -- train tokens/sec of the CogView GPT hot path on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 --steps 10 --warmup 3            (spawns its own 8 ranks through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full training step of pretrain_gpt2.py's loop on one synthetic batch already resident in
HBM: forward (GPT2Model) + fused cross entropy + backward + data-parallel gradient all-reduce + overflow check /
global-norm clip / AdamW + 16-bit parameter write.  Default workload: the configuration BASELINE.json's metric is
quoted on -- the 4B CogView-base GPT (48 layers / 2560 hidden / 40 heads; 16 B/param of weights, master copy,
Adam moments and gradients = 64 GB, so the whole model fits one 288-GB MI355X and every rank is a full data-parallel
replica, BASELINE.json configs[3]) -- rows of 1089 random tokens (1088 model positions), vocab 58240, weak scaling
(per-GPU micro-batch fixed).  `--config cogview-small-336M` runs configs[1]; `--model-parallel 2` runs configs[2] (the
4B model split column/row-wise over adjacent rank pairs, vocab 58368); `--config vqvae` runs configs[4] (VQ-VAE
tokenizer: img2code + code2img of 256 images of 256x256 per step, replicas only); `--config cogview-tiny-18M` runs
configs[0] (4 L / 256 h / 4 heads, rows of 256 -- the case the reference itself runs on a CPU).  Without `--dtype` the
run measures fp16 -- the reference's own storage type (fp16/fp16.py) and the one whose logits meet north_star's 1e-3 bar
at 48 layers: it is the headline `value` / `dtype` -- and, at N = 1, the same step in bf16 (BASELINE configs[3]'s
extension; logits 2e-3..5e-3, above the bar) reported under "bf16_leg".  Prints ONE JSON line (rank 0).
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
    "cogview-tiny-18M": (4, 256, 4),          # BASELINE.json configs[0]: rows of 256 tokens (s = 255), b = 4
}
ROW_LEN = {"cogview-tiny-18M": 256}            # tokens per data row (default ROW = 1089)
# per-GPU micro-batch (sequences): b x 1088 rows must fill whole rounds of 256-row GEMM tiles on 256 CUs.
#   30 x 1088 = 127.5 -> 128 row tiles, and 128 x {10, 30, 40} column tiles of 256 are exactly 5 / 15 / 20 rounds.  4B since
#   round 6 (24 before: 102 row tiles, 3.98 / 11.95 / 15.94 rounds): the fixed 19 ms of the optimizer pass are spread over 25 %
#   more tokens -- +0.8 % tokens/s in alternating runs (profiles/r06_batch_24_vs_30.log); HBM high-water mark 239 GB allocated /
#   255 GB reserved of 288 (204 / 219 at 24), reported by every run as peak_hbm_gb.
DEFAULT_BATCH = {"cogview-small-336M": 30, "cogview-base-4B": 30, "cogview-tiny-18M": 4}
METRIC = {"cogview-base-4B": "train tokens/sec/node (seq1089, 4B GPT) at 1/2/4/8 MI355X; % MFMA roofline",
          "cogview-small-336M": "train tokens/sec/node (seq1089, 336M GPT) at 1/2/4/8 MI355X; % MFMA roofline",
          "cogview-tiny-18M": "train tokens/sec/node (seq256, 18M GPT, BASELINE configs[0]) at 1/2/4/8 MI355X; % MFMA roofline"}
N_TOKEN_IDS = 58219


def padded_vocab(mp):
    """58219 tokens padded to a multiple of 128 x model-parallel size (arguments.py --make-vocab-size-divisible-by,
    utils / pretrain_gpt2.py:get_model): 58240 at MP=1, 58368 at MP=2."""
    m = 128 * mp
    return (N_TOKEN_IDS + m - 1) // m * m


VOCAB = padded_vocab(1)
# bars of tests/test_depth_parity_gpu.py for the logits rel-L2 against the fp32 CPU oracle at FULL depth; the value
# next to them in the JSON line is MEASURED by this run on the model it timed (measure_parity)
LOGITS_TOLERANCE = {"fp16": 1e-3, "bf16": 8e-3}
PEAK_FP32_MFMA_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 (exact fp32), MI355X_MICROARCH.md
ROW = 1089               # tokens per data row; the model sees ROW-1 = 1088 positions (pretrain_gpt2.py:273-275)
PEAK_MFMA_TFLOPS = 2500.0   # MI355X dense bf16/fp16 MFMA peak (MI355X_MICROARCH.md)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def flops_per_token(L, h, V, s=ROW - 1):
    """fwd+bwd, full (non-causal-discounted) attention, no recompute credit."""
    return 3.0 * (L * (24.0 * h * h + 4.0 * s * h) + 2.0 * h * V)


def flops_per_token_causal(L, h, V, s=ROW - 1):
    """SURVEY.md section 8(d)'s "causal-discounted variant": the two attention products (4 s h per token and layer, forward)
    counted over the visible half of the score matrix only, (s + 1) / 2s of it -- 24.35 GFLOP/token at 4B against 25.148."""
    return 3.0 * (L * (24.0 * h * h + 4.0 * s * h * (s + 1) / (2.0 * s)) + 2.0 * h * V)


def hardware_flops_per_token(L, h, V, s=ROW - 1, recompute=False):
    """What the kernels EXECUTE per token: the attention products over the 64 x 64 score blocks the kernels visit (a block on
    the diagonal is computed whole: nb (nb + 1) / 2 of nb^2 blocks, nb = ceil(s / 64): 153 / 289 at s = 1088), and, with
    activation recompute, a second forward pass of every layer ("hardware FLOPs if recompute is on")."""
    nb = (s + 63) // 64
    att = 4.0 * (64.0 * 64.0 * nb * (nb + 1) / 2.0) / s * h      # per token and layer, forward: 2 products x 2 flops, h = heads x 64
    layer_fwd = 24.0 * h * h + att
    return 3.0 * (L * layer_fwd + 2.0 * h * V) + (L * layer_fwd if recompute else 0.0)


PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s HBM3E peak (6.3 TB/s is what a float4 copy reaches)


def by_family(stats, step_ms_sampled):
    """roofline.by_family: every kernel family bracketed by HIP events in the sampled steps (ops.timed_launch), each against
    the roofline that bounds it -- attention on EXECUTED FLOPs (visited 64 x 64 blocks; backward = 2.5 x forward) against the
    dense MFMA peak, the HBM-bound families on their algorithmic bytes against the 8 TB/s peak."""
    out = {}
    bv = stats["by_variant"]
    g = stats["gemm"]
    out["gemm"] = {"bound": "mfma", "achieved": g["tflops"], "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": g["tflops"] / PEAK_MFMA_TFLOPS,
                   "launches": g["launches"], "share_of_step_time": g["total_ms"] / step_ms_sampled}
    for fam in ("attention", "layernorm", "adamw", "grad_stats"):
        v = bv.get(fam)
        if v is None:
            continue
        if fam == "attention":
            out[fam] = {"bound": "mfma", "achieved": v["tflops"], "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s (executed: visited 64x64 blocks)",
                        "frac": v["tflops"] / PEAK_MFMA_TFLOPS}
        else:
            out[fam] = {"bound": "hbm", "achieved": v["algo_gbytes_per_s"], "peak": PEAK_HBM_GBS, "unit": "GB/s (algorithmic bytes)",
                        "frac": v["algo_gbytes_per_s"] / PEAK_HBM_GBS}
        out[fam].update(launches=v["launches"], avg_launch_ms=v["avg_ms"], share_of_step_time=v["total_ms"] / step_ms_sampled)
    out["by_launch"] = {k: v for k, v in stats["by_shape"].items() if k.split(" ")[0] in ("attention", "layernorm", "adamw")}
    return out


def gemm_flops(M, N, K):
    return 2.0 * M * N * K


def ops_gemm_families():
    from cogview_amd import ops
    return ops.GEMM_FAMILIES


def _cpu_thread_candidates():
    """Thread counts worth trying for the CPU leg: the physical cores, half of them (one socket / no
    cross-CCD traffic) and 32; never more than the logical CPUs the process may use."""
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or logical
    except Exception:
        phys = logical
    phys = min(phys, logical)
    return sorted({max(1, min(c, logical)) for c in (phys, phys // 2, 32)}, reverse=True), phys, logical


def cpu_baseline(L, h, heads, sample_layers=2, row=ROW):
    """The CPU oracle (oracle/cogview_oracle.py, fp32, torch CPU threads) timed on a bounded sample of the same
    workload, by SURVEY section 8(d)'s protocol: ONE 1088-token sequence, forward + backward; the layer part (`sample_layers`
    of the L layers, scaled by L / sample_layers) and the head part (embedding + final LayerNorm + tied LM head + cross
    entropy) are each the MEDIAN OF 3 timed passes after a warm-up pass; the thread count is the fastest of
    {physical cores, half of them, 32} on one layer (all logical CPUs oversubscribe the box: 8.9 tokens/s on 128 threads
    in round 5 where 8 threads of the build container did 23.6)."""
    from oracle import cogview_oracle as O
    torch.manual_seed(0)
    s = row - 1
    sample_layers = min(sample_layers, L)
    g = torch.Generator().manual_seed(1)
    p = {"word_embeddings.weight": torch.randn(VOCAB, h, generator=g) * 0.02,
         "transformer.position_embeddings.weight": torch.randn(row, h, generator=g) * 0.02,
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
    ids = torch.randint(0, N_TOKEN_IDS, (1, row), generator=g)
    tokens, labels = ids[:, :-1], ids[:, 1:]
    pos = torch.arange(s).unsqueeze(0)
    mask = O.build_mask(s, s)
    x0 = (torch.randn(1, s, h, generator=g) * 0.02).requires_grad_(True)

    def run_head():
        t0 = time.perf_counter()
        logits = O.gpt2_forward(tokens, pos, mask, p, 0, heads)
        O.lm_loss(logits, labels, torch.ones(1, s)).backward()
        return time.perf_counter() - t0

    def run_layers(n):
        t0 = time.perf_counter()
        x = x0
        for l in range(n):
            x = O.transformer_layer(x, mask, p, f"transformer.layers.{l}.", heads)
        x.float().square().mean().backward()
        return time.perf_counter() - t0

    before = torch.get_num_threads()
    cands, phys, logical = _cpu_thread_candidates()
    run_layers(1)                           # warm-up: allocator, MKL thread pool
    trial = {}
    for c in cands:
        torch.set_num_threads(c)
        run_layers(1)
        trial[c] = run_layers(1)
    best = min(trial, key=trial.get)
    torch.set_num_threads(best)
    run_head()                              # warm-up of the head shapes
    t_head = sorted(run_head() for _ in range(3))[1]
    t_samp = sorted(run_layers(sample_layers) for _ in range(3))[1]
    torch.set_num_threads(before)
    t_layers = t_samp * (L / sample_layers)
    return {"value": s / (t_head + t_layers), "unit": "tokens/s", "cores": best, "kind": "port",
            "physical_cores": phys, "logical_cpus": logical,
            "threads_tried_seconds_per_layer": {str(c): round(v, 3) for c, v in trial.items()},
            "sample": f"1 sequence x {s} positions, fp32 oracle fwd+bwd, warm-up + median of 3 each: {sample_layers} of {L} layers "
                      f"({t_samp:.2f}s, scaled x{L / sample_layers:g}, extrapolated) + embedding / final LayerNorm / tied LM head / CE "
                      f"({t_head:.2f}s); {best} threads = fastest of {cands} on one layer"}


def measure_parity(inner, L, heads, row=ROW):
    """The CPU oracle as the CHECKER of the model this run just timed (part of the cpu_baseline leg: rank 0, N = 1):
    one 1088-position sequence through ALL layers of the timed model's current weights (as stored, widened to fp32) on
    the CPU in fp32, against the HIP forward of the same sequence -- relative L2 of the logits and of the residual stream
    after 1, 2, 4, ... layers (oracle/depth_check.py).  The oracle forward is also a CPU timing sample (forward only)."""
    from oracle import depth_check as D
    ids = torch.randint(0, N_TOKEN_IDS, (1, row - 1), generator=torch.Generator().manual_seed(4321)).cuda()
    rep = D.depth_report(inner, ids, L, heads)
    return {"logits_rel_l2": rep["logits"],
            "logits_rel_l2_with_fp32_output": rep["logits_fp32_out"],     # the same logits without their last rounding to 16 bits
            "residual_stream_rel_l2_after_n_layers": {str(n): e for n, e in rep["stream"].items()},
            "against": "oracle/cogview_oracle.py fp32 on the CPU, all %d layers, the timed model's weights as stored, "
                       "1 sequence x %d positions, dropout off" % (L, row - 1),
            "oracle_forward_seconds": rep["oracle_seconds"],
            "oracle_forward_tokens_per_s": rep["tokens"] / rep["oracle_seconds"]}


def cpu_baseline_vqvae(n_img=16):
    """The CPU oracle (oracle/cogview_oracle.py vqvae_encode / vqvae_decode = F.conv2d / F.conv_transpose2d on the
    production channel sizes, fp32, torch CPU threads) on a bounded sample: `n_img` images of 256x256."""
    from oracle import cogview_oracle as O
    g = torch.Generator().manual_seed(0)
    ch, ed, ne = 512, 256, 8192
    def w(*shape):
        fan = 1
        for d in shape[1:]:
            fan *= d
        return torch.randn(*shape, generator=g) / fan ** 0.5
    p = {"enc_b.blocks.0.weight": w(ch, 3, 4, 4), "enc_b.blocks.0.bias": torch.zeros(ch),
         "enc_b.blocks.2.weight": w(ch, ch, 4, 4), "enc_b.blocks.2.bias": torch.zeros(ch),
         "enc_b.blocks.4.weight": w(ch, ch, 4, 4), "enc_b.blocks.4.bias": torch.zeros(ch),
         "enc_b.blocks.6.weight": w(ed, ch, 1, 1), "enc_b.blocks.6.bias": torch.zeros(ed),
         "quantize_t.embed": torch.randn(ed, ne, generator=g),
         "dec.blocks.0.weight": w(ed, ch, 4, 4), "dec.blocks.0.bias": torch.zeros(ch),
         "dec.blocks.2.weight": w(ch, ch, 4, 4), "dec.blocks.2.bias": torch.zeros(ch),
         "dec.blocks.4.weight": w(ch, ch, 4, 4), "dec.blocks.4.bias": torch.zeros(ch),
         "dec.blocks.6.weight": w(3, ch, 1, 1), "dec.blocks.6.bias": torch.zeros(3)}
    img = torch.randn(n_img, 3, 256, 256, generator=g)
    with torch.no_grad():
        O.vqvae_decode(O.vqvae_encode(img[:1], p)[0], p)          # warm-up
        t0 = time.perf_counter()
        ids = O.vqvae_encode(img, p)[0]
        t1 = time.perf_counter()
        O.code2img_denorm(O.vqvae_decode(ids, p))
        t2 = time.perf_counter()
    return {"value": n_img / (t2 - t0), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_img} images 256x256, fp32 oracle (production channels 512 / 256 / 8192 codes): "
                      f"encode+quantise {t1 - t0:.2f}s, decode {t2 - t1:.2f}s"}


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: become the launcher -- re-exec this script under
    torch.distributed.run with one rank per GPU on 127.0.0.1 (what scripts/pretrain_single_node.sh:49 does for the
    reference with the deepspeed launcher).  Rank 0 of the spawned job prints the one JSON line; the launcher form
    `python -m torch.distributed.run ... bench.py --gpus N` keeps working (WORLD_SIZE is then already set)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] --gpus {args.gpus} without a launcher: spawning {args.gpus} ranks: {' '.join(cmd)}")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def setup_dist(args):
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                    # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
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
    return world, rank


def sampled_step(i, steps):
    """Which timed steps carry HIP events around their launches (~1500 event pairs per 4B step since round 5 brackets attention,
    LayerNorm and the optimizer too: ~1 % of such a step; round 4 measured 0.6 % for the 870 GEMM pairs,
    profiles/r04_bench_event_overhead_ab.log): one step in eight from 16 steps on (the driver's 20: steps 7 and 15), one in four
    below that, every step of a run shorter than four."""
    if steps < 4:
        return True
    period = 8 if steps >= 16 else 4
    return i % period == period - 1


def timed_steps(step, args, world):
    """W untimed warm-up steps, then exactly K steps bracketed by barrier + synchronize; MAX over ranks.
    Returns (elapsed seconds, kernel-timing statistics or None, value returned by the last step)."""
    import torch.distributed as dist
    from cogview_amd import ops
    last = None
    for _ in range(args.warmup):
        last = step()
    timing = None if args.no_kernel_timing else ops.enable_gemm_timing()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if timing is not None:               # HIP events around every kernel family's launches of SOME timed steps only
            ops.sample_gemm_timing(sampled_step(i, args.steps))
        last = step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    stats = ops.collect_gemm_timing() if timing is not None else None
    t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.item(), stats, last


def latest_profile(pattern):
    """Newest profiles/rNN_<pattern> (measured off-line, see tools/collect_traffic.sh), or None."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern)))
    return hits[-1] if hits else None


def run_gpt(args, dtype_name, world, rank, mp, parity=False):
    """One measurement of the GPT train step in `dtype_name`; returns the JSON object (without cpu_baseline).
    parity: after the timed steps, check the timed model against the CPU oracle at full depth (measure_parity)."""
    from cogview_amd import mpu, training
    from cogview_amd.fp16 import FP16_Module, FP16_Optimizer
    from cogview_amd.model import GPT2Model, PyTorchDistributedDataParallel, gpt2_get_params_for_weight_decay_optimization
    from cogview_amd.optim import FusedAdam

    torch.manual_seed(1234)
    mpu.model_parallel_cuda_manual_seed(1234)
    torch.cuda.reset_peak_memory_stats()                  # peak_hbm_gb is this leg's own high-water mark
    L, h, heads = CONFIGS[args.config]
    row = ROW_LEN.get(args.config, ROW)
    vocab = padded_vocab(mp)
    dp_world = world // mp
    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float16
    t0 = time.perf_counter()
    model = GPT2Model(L, vocab, h, heads, args.dropout, args.dropout, args.dropout, row, 0, args.checkpoint_activations)
    n_params = sum(p.numel() for p in model.parameters())          # per rank (a shard when mp > 1)
    model = FP16_Module(model.cuda(), dtype=dtype, keep_half_outputs=True)
    ddp = None
    if dp_world > 1:
        model = ddp = PyTorchDistributedDataParallel(model, process_group=mpu.get_data_parallel_group(),
                                                     shard_optimizer=args.shard_optimizer)
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
    if ddp is not None:
        opt.attach_data_parallel(ddp)
    model.train()
    log(f"[bench] {args.config}: {n_params / 1e6:.1f}M params per rank, built in {time.perf_counter() - t0:.1f}s, "
        f"rank {rank}/{world} (mp {mp} x dp {dp_world}), micro-batch {args.batch}, dtype {dtype_name}, "
        f"dropout {args.dropout}, recompute {args.checkpoint_activations}")

    gen = torch.Generator().manual_seed(1234 + mpu.get_data_parallel_rank())     # one batch per model-parallel group
    text = torch.randint(0, N_TOKEN_IDS, (args.batch, row), generator=gen).cuda()      # resident in HBM
    loss_mask = torch.ones(args.batch, row, device="cuda")
    batch = training.get_batch(text, loss_mask)

    def step():
        return training.train_step(batch, model, opt, clip_grad=1.0, world_size=world, check_forward_nan=True)

    elapsed, gemm_stats, (loss, skipped) = timed_steps(step, args, world)
    final_loss = loss.item()
    assert final_loss == final_loss, "loss is NaN"

    tokens_per_step = dp_world * args.batch * (row - 1)
    value = tokens_per_step * args.steps / elapsed
    fpt = flops_per_token(L, h, vocab, s=row - 1)
    out = {
        "metric": METRIC[args.config], "value": value, "unit": "tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype_name, "data": "synthetic",
        "config": {"workload": f"{args.config} ({L}L/{h}h/{heads} heads, {n_params / 1e6:.1f}M params per rank), rows of "
                               f"{row} random token ids -> {row - 1} model positions, vocab {vocab}, full train step "
                               f"(fwd+CE+nan guard+bwd+grad all-reduce+clip+AdamW)",
                   "global_batch": dp_world * args.batch, "seq_len": row, "model_positions": row - 1,
                   "parallelism": f"dp{dp_world}" + (f"-mp{mp}" if mp > 1 else "") + ("-sharded-optimizer" if (ddp is not None and ddp.shard is not None) else ""),
                   "dropout": args.dropout,
                   "activation_recompute": bool(args.checkpoint_activations), "loss": final_loss,
                   "loss_scale": opt.loss_scale, "skipped_last_step": int(skipped),
                   "logits_rel_l2_vs_fp32_reference": {"measured": None, "tolerance_in_tests": LOGITS_TOLERANCE[dtype_name]}},
        "model_tflops_per_gpu": value / world * fpt / 1e12,
        "mfma_roofline_frac_end_to_end": value / world * fpt / 1e12 / PEAK_MFMA_TFLOPS,
        # HBM high-water mark of this rank over warm-up + timed steps (torch caching allocator; of 288 GB per MI355X)
        "peak_hbm_gb": {"allocated": torch.cuda.max_memory_allocated() / 1e9, "reserved": torch.cuda.max_memory_reserved() / 1e9},
    }
    # SURVEY.md section 8(d)'s other two FLOP conventions for the same measured tokens/s: attention over the visible half of
    # the scores only, and the FLOPs the kernels execute (visited score blocks; + the recompute forward with --checkpoint-activations)
    fpt_c = flops_per_token_causal(L, h, vocab, s=row - 1)
    fpt_hw = hardware_flops_per_token(L, h, vocab, s=row - 1, recompute=bool(args.checkpoint_activations))
    out["flops_per_token"] = {"model_full_attention": fpt, "model_causal_discounted": fpt_c, "hardware_executed": fpt_hw}
    out["model_tflops_causal_discounted"] = value / world * fpt_c / 1e12
    out["mfma_roofline_frac_causal_discounted"] = value / world * fpt_c / 1e12 / PEAK_MFMA_TFLOPS
    out["hardware_tflops_per_gpu"] = value / world * fpt_hw / 1e12
    out["mfma_roofline_frac_hardware_flops"] = value / world * fpt_hw / 1e12 / PEAK_MFMA_TFLOPS
    sampled = sum(1 for i in range(args.steps) if sampled_step(i, args.steps))          # steps whose launches carried HIP events (timed_steps)
    if gemm_stats is not None:
        all_stats, gemm_stats = gemm_stats, dict(gemm_stats["gemm"], by_variant={k: v for k, v in gemm_stats["by_variant"].items() if k in ops_gemm_families()},
                                                  by_shape=gemm_stats["by_shape"])
        out["roofline"] = {"bound": "mfma", "kernel": "gemm_w4_kernel<%s> (256x256x64 tiles, 4 waves of 128x128, persistent work queues, 16x16x32 MFMA; NT fwd, "
                                                       "NN dgrad, TN wgrad in grouped launches of whole rounds of tiles)" % dtype_name,
                           "achieved": gemm_stats["tflops"], "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": gemm_stats["tflops"] / PEAK_MFMA_TFLOPS, "traffic": None,
                           "launches": gemm_stats["launches"], "avg_launch_ms": gemm_stats["avg_ms"],
                           "sampled_steps": sampled, "share_of_step_time": gemm_stats["total_ms"] / (elapsed / args.steps * sampled * 1e3),
                           "by_variant_tflops": gemm_stats["by_variant"]}
        out["roofline"]["by_family"] = by_family(all_stats, elapsed / args.steps * sampled * 1e3)
        log("[bench] launches by kernel family and shape (TFLOP/s, GB/s of algorithmic bytes, launches, avg ms):")
        for k, v in gemm_stats["by_shape"].items():
            log(f"    {k:72s} {v['tflops']:8.1f} {v['gbytes_per_s']:8.1f} {v['launches']:6d} {v['avg_ms']:9.4f}")
        # HBM-side traffic of the dominant kernel: measured off-line with rocprofv3 PMC passes (tools/collect_traffic.sh,
        # FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 x2 read correction) for the single-GPU workloads (collected on the bf16 build).
        tpath = latest_profile("gemm_hbm_traffic_pmc_%s_b%d.json" % (args.config.split("-")[-1], args.batch))
        if tpath is not None and mp == 1:            # both 16-bit storage types move the same bytes
            t = json.load(open(tpath))
            algo = gemm_stats["algo_bytes"] / max(gemm_stats["launches"], 1)
            out["roofline"]["traffic"] = t["gemm_hbm_bytes_per_launch_corrected"]
            out["roofline"]["traffic_unit"] = "bytes per launch (L2-miss side: FETCH_SIZE*2 + WRITE_SIZE, includes Infinity-Cache hits)"
            out["roofline"]["traffic_source"] = os.path.relpath(tpath, ROOT)
            out["roofline"]["algorithmic_bytes_per_launch"] = algo
    if parity:
        t0 = time.perf_counter()
        par = measure_parity(inner, L, heads, row)
        out["config"]["logits_rel_l2_vs_fp32_reference"].update(
            measured=par["logits_rel_l2"], measured_by="this run, on the model it timed (after its %d steps)" % (args.steps + args.warmup),
            detail=par)
        log(f"[bench] {dtype_name}: logits rel-L2 vs the fp32 CPU oracle at full depth {par['logits_rel_l2']:.3e} "
            f"(bar {LOGITS_TOLERANCE[dtype_name]:g}); oracle forward {par['oracle_forward_seconds']:.1f}s, "
            f"leg {time.perf_counter() - t0:.1f}s")
    # release this model's HBM (a second dtype leg may follow in the same process)
    del step, batch, opt, model, inner, groups, ddp
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return out


VQ_BATCH = 256
VQ_FLOPS_PER_IMAGE = {"encode": 48.32e9, "decode": 176.29e9}


def run_vqvae(args, world, rank):
    """BASELINE configs[4]: one step = img2code then code2img of `batch` normalised 256x256 images already resident in
    HBM (production tokenizer: 512 channels, 256-d codes, 8192 entries; fp32 end to end on the exact-fp32 MFMA).
    Replicas only: every rank tokenizes its own batch, no collective on the data path."""
    from cogview_amd import vqvae
    torch.manual_seed(0)
    model = vqvae.new_model().cuda().eval()
    b = args.batch
    gen = torch.Generator().manual_seed(rank)
    img = torch.randn(b, 3, 256, 256, generator=gen).cuda()

    def step():
        ids = vqvae.img2code(model, img)
        out = vqvae.code2img(model, ids.view(b, 32, 32))
        return ids, out

    elapsed, stats, (ids, out) = timed_steps(step, args, world)
    assert ids.shape == (b, 1024) and out.shape == (b, 3, 256, 256) and bool(torch.isfinite(out).all())
    value = world * b * args.steps / elapsed
    fl = VQ_FLOPS_PER_IMAGE["encode"] + VQ_FLOPS_PER_IMAGE["decode"]
    res = {"metric": "VQ-VAE encode+decode images/sec (256x256 -> 32x32 codes -> 256x256), batch 256; % fp32-MFMA roofline",
           "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"VQ-VAE tokenizer (vqvae/api.py new_model: 512 channels, 256-d codes, 8192 entries), "
                                  f"img2code + code2img of {b} images of 256x256 per step per GPU, fp32 (exact-fp32 MFMA)",
                      "global_batch": world * b, "parallelism": f"replicas{world}",
                      "distinct_codes_used": int(ids.unique().numel())},
           "model_tflops_per_gpu": value / world * fl / 1e12,
           "mfma_roofline_frac_end_to_end": value / world * fl / 1e12 / PEAK_FP32_MFMA_TFLOPS}
    sampled = sum(1 for i in range(args.steps) if sampled_step(i, args.steps))
    if stats is not None:
        conv = stats["by_variant"].get("conv", {"tflops": 0.0, "launches": 0, "avg_ms": 0.0})
        conv_ms = conv["avg_ms"] * conv["launches"]
        res["roofline"] = {"bound": "mfma", "kernel": "conv_kernel (implicit GEMM, v_mfma_f32_32x32x2_f32: 4x4 s2 conv, 1x1, "
                                                      "4-parity 4x4 s2 transposed conv)",
                           "achieved": conv["tflops"], "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": conv["tflops"] / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                           "launches": conv["launches"], "avg_launch_ms": conv["avg_ms"],
                           "sampled_steps": sampled, "share_of_step_time": conv_ms / (elapsed / args.steps * sampled * 1e3),
                           "by_kernel_family_tflops": stats["by_variant"]}
        log("[bench] VQ-VAE launches (TFLOP/s, launches, avg ms):")
        for k, v in stats["by_shape"].items():
            log(f"    {k:52s} {v['tflops']:8.1f} {v['launches']:6d} {v['avg_ms']:9.4f}")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cogview-base-4B", choices=list(CONFIGS) + ["vqvae"])
    ap.add_argument("--batch", type=int, default=0,
                    help="micro-batch per GPU (sequences of 1089 tokens / images); default per config")
    ap.add_argument("--dtype", default=None, choices=["bf16", "fp16"],
                    help="default: fp16 (the reference's dtype; meets the 1e-3 logits bar) as the headline value, plus a bf16 leg "
                         "in the same line when N=1")
    ap.add_argument("--model-parallel", type=int, default=1,
                    help="model-parallel size (BASELINE configs[2]: 2); ranks r, r+1 form a group (mpu/initialize.py)")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="data-parallel exchange as reduce-scatter + owned-slice AdamW + all-gather (ZeRO stage 1 shape, "
                         "scripts/ds_config_zero.json) instead of the bucketed all-reduce")
    ap.add_argument("--dropout", type=float, default=0.1, help="reference default (arguments.py:30,40)")
    ap.add_argument("--checkpoint-activations", action="store_true",
                    help="recompute each layer in backward (the reference's scripts do; 288 GB HBM makes it unnecessary)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-second-dtype", action="store_true", help="skip the bf16 leg of the default run")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the full-depth logits check of the timed model against the CPU oracle (about one CPU "
                         "minute per dtype at 4B; part of the cpu_baseline leg, so --no-cpu-baseline skips it too)")
    args = ap.parse_args()
    if args.batch <= 0:
        args.batch = VQ_BATCH if args.config == "vqvae" else DEFAULT_BATCH[args.config]

    # stdout carries the ONE JSON line and nothing else: the mirrors print what the reference prints ("> initializing model
    # parallel ..."), which goes to stderr for the length of the run
    json_out, sys.stdout = sys.stdout, sys.stderr
    world, rank = setup_dist(args)
    import torch.distributed as dist
    from cogview_amd import mpu

    if args.config == "vqvae":
        mpu.initialize_model_parallel(1)
        out = run_vqvae(args, world, rank)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_vqvae()
    else:
        mp = args.model_parallel
        assert world % mp == 0, "--gpus must be a multiple of --model-parallel"
        mpu.initialize_model_parallel(mp)
        L, h, heads = CONFIGS[args.config]
        parity = world == 1 and mp == 1 and not args.no_cpu_baseline and not args.no_parity
        out = run_gpt(args, args.dtype or "fp16", world, rank, mp, parity=parity)
        if args.dtype is None and world == 1 and not args.no_second_dtype:
            leg = run_gpt(args, "bf16", world, rank, mp, parity=parity)
            out["bf16_leg"] = {k: leg[k] for k in ("value", "unit", "ms_per_step", "dtype", "model_tflops_per_gpu",
                                                   "mfma_roofline_frac_end_to_end", "mfma_roofline_frac_causal_discounted",
                                                   "mfma_roofline_frac_hardware_flops")}
            out["bf16_leg"]["loss"] = leg["config"]["loss"]
            out["bf16_leg"]["loss_scale"] = leg["config"]["loss_scale"]
            out["bf16_leg"]["logits_rel_l2_vs_fp32_reference"] = leg["config"]["logits_rel_l2_vs_fp32_reference"]
            if "roofline" in leg:
                out["bf16_leg"]["roofline"] = {k: leg["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "launches",
                                                                               "avg_launch_ms", "share_of_step_time")}
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(L, h, heads, row=ROW_LEN.get(args.config, ROW))
            # the REFERENCE ITSELF timed beside the port where /root/reference exists (the build container): configs[0] whole
            # (oracle/time_reference_cfg1.py), the 4B headline on the same bounded sample as the port above
            # (oracle/time_reference_4B_sample.py) -- committed under profiles/ and quoted here: kind "reference", measured
            # THERE, next to the port measured HERE
            rname = {"cogview-tiny-18M": "cfg1_cpu_reference_vs_port.json", "cogview-base-4B": "4B_cpu_reference_vs_port.json"}
            rpath = latest_profile(rname[args.config]) if args.config in rname else None
            if rpath is not None:
                r = json.load(open(rpath))
                out["cpu_baseline_reference"] = {
                    "value": r["reference"]["tokens_per_s"], "unit": "tokens/s", "cores": r["threads"], "kind": "reference",
                    "sample": r["config"], "measured_where": r["where"] + ", not on this box",
                    "port_on_the_same_cores": r["port"]["tokens_per_s"], "source": os.path.relpath(rpath, ROOT)}
    if rank == 0:
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1:
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
