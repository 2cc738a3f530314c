"""Contention experiment (round-3 verdict item 6a): the 4B training step while a side-stream kernel HOLDS n CUs for the whole
step -- what RCCL's channel kernels do during the data-parallel backward.  Tests the claim that the persistent one-workgroup-
per-CU GEMM degrades gracefully (per-XCD work queues: a workgroup that starts late or never simply takes fewer items).
    python tools/probes/contention.py [--steps 4] [--held 0,8,16,32,48,64]
Prints tokens/s per setting and the ratio to (256 - n) / 256."""
import argparse, ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--held", default="0,8,16,32,48,64,0")
    ap.add_argument("--config", default="cogview-base-4B")
    ap.add_argument("--dtype", default="fp16")
    a = ap.parse_args()
    hog = ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "_cu_hog.so"))
    hog.cu_hog_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
    import bench
    from cogview_amd import mpu, training
    from cogview_amd.fp16 import FP16_Module, FP16_Optimizer
    from cogview_amd.model import GPT2Model, gpt2_get_params_for_weight_decay_optimization
    from cogview_amd.optim import FusedAdam
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
    dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
    mpu.initialize_model_parallel(1)
    torch.manual_seed(1234); mpu.model_parallel_cuda_manual_seed(1234)
    L, h, heads = bench.CONFIGS[a.config]
    b = bench.DEFAULT_BATCH[a.config]
    dtype = torch.float16 if a.dtype == "fp16" else torch.bfloat16
    model = FP16_Module(GPT2Model(L, bench.VOCAB, h, heads, 0.1, 0.1, 0.1, bench.ROW, 0, False).cuda(), dtype=dtype, keep_half_outputs=True)
    groups = gpt2_get_params_for_weight_decay_optimization(model.module)
    for grp in groups:
        for p in grp["params"]:
            if not hasattr(p, "model_parallel"):
                p.model_parallel = False
    opt = FP16_Optimizer(FusedAdam(groups, lr=1.5e-4, weight_decay=0.01), dynamic_loss_scale=True,
                         dynamic_loss_args={"init_scale": 2 ** 16 if dtype == torch.float16 else 1.0, "scale_window": 1000, "min_scale": 1, "delayed_shift": 2})
    model.train()
    text = torch.randint(0, bench.N_TOKEN_IDS, (b, bench.ROW), generator=torch.Generator().manual_seed(1234)).cuda()
    batch = training.get_batch(text, torch.ones(b, bench.ROW, device="cuda"))
    step = lambda: training.train_step(batch, model, opt, clip_grad=1.0)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    base_ms = None
    rows = []
    for n in [int(x) for x in a.held.split(",")]:
        # the hog runs for the whole measurement (estimated from the unloaded step time, +60 % margin, capped by its own clock)
        est = (base_ms or 650.0) * a.steps * (256.0 / max(256 - n, 64)) * 1.6 if n else 0.0
        torch.cuda.synchronize()
        if n:
            hog.cu_hog_launch(n, est, ctypes.c_void_p(side.cuda_stream))
            time.sleep(0.05)                       # let the hog take its CUs before the step's first kernel is enqueued
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.current_stream().synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / a.steps
        torch.cuda.synchronize()                   # hog's tail
        if n == 0 and base_ms is None:
            base_ms = ms
        tok = b * (bench.ROW - 1) / ms * 1e3
        rows.append({"held_cus": n, "ms_per_step": round(ms, 2), "tokens_per_s": round(tok, 1),
                     "vs_unloaded": round(base_ms / ms, 4) if base_ms else None, "cu_fraction_left": round((256 - n) / 256, 4)})
        print(json.dumps(rows[-1]), flush=True)


if __name__ == "__main__":
    main()
