"""Which Python lines launch the small torch kernels (copies, fills, casts) inside a 4B training step?  torch profiler with
stacks around ONE step of a 6-layer model at the 4B width (same per-layer op sequence), grouped by (op, innermost repo frame)."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
from torch.profiler import ProfilerActivity, profile
import bench
from cogview_amd import mpu, training
from cogview_amd.fp16 import FP16_Module, FP16_Optimizer
from cogview_amd.model import GPT2Model, gpt2_get_params_for_weight_decay_optimization
from cogview_amd.optim import FusedAdam
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29579")
dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
mpu.initialize_model_parallel(1)
torch.manual_seed(1234); mpu.model_parallel_cuda_manual_seed(1234)
L, h, heads, b = 6, 2560, 40, 8
model = FP16_Module(GPT2Model(L, bench.VOCAB, h, heads, 0.1, 0.1, 0.1, bench.ROW, 0, False).cuda(), dtype=torch.float16, keep_half_outputs=True)
groups = gpt2_get_params_for_weight_decay_optimization(model.module)
for grp in groups:
    for p in grp["params"]:
        if not hasattr(p, "model_parallel"):
            p.model_parallel = False
opt = FP16_Optimizer(FusedAdam(groups, lr=1.5e-4, weight_decay=0.01), dynamic_loss_scale=True, dynamic_loss_args={"init_scale": 2 ** 16, "scale_window": 1000, "min_scale": 1, "delayed_shift": 2})
model.train()
text = torch.randint(0, bench.N_TOKEN_IDS, (b, bench.ROW), generator=torch.Generator().manual_seed(1)).cuda()
batch = training.get_batch(text, torch.ones(b, bench.ROW, device="cuda"))
step = lambda: training.train_step(batch, model, opt, clip_grad=1.0, check_forward_nan=True)
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_type.name != "CPU":
        continue
    name = ev.name
    if not any(k in name for k in ("copy_", "fill_", "zero_", "clone", "contiguous", "aten::to", "aten::_to_copy", "cat", "empty_strided", "aten::mul", "aten::add", "aten::sum", "index", "nonzero", "where", "isfinite", "aten::all", "lt", "bitwise")):
        continue
    frame = next((f for f in ev.stack if "/cogview_amd/" in f or "bench.py" in f or "/tools/" in f), ev.stack[0] if ev.stack else "?")
    dev = sum(k.duration for k in ev.kernels) if ev.kernels else 0.0
    a = agg[(name, frame.split("/root/repo/")[-1] if "/root/repo/" in frame else frame[-90:])]
    a[0] += 1; a[1] += dev
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print(f"model: {L} layers; calls and device microseconds of small torch ops in ONE step, by launching line")
for (name, frame), (n, us) in rows[:40]:
    print(f"{us:9.1f} us {n:5d}  {name:28s} {frame}")
