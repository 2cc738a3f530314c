"""Which Python lines launch the non-cogview kernels (copies, fills, casts) during one train step?  Runs the
336M config for 2 steps under torch.profiler with stacks and prints the aten ops by call site."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29588")
import torch, torch.distributed as dist
dist.init_process_group("nccl", init_method="env://", world_size=1, rank=0)
from cogview_amd import mpu, training
from cogview_amd.fp16 import FP16_Module, FP16_Optimizer
from cogview_amd.model import GPT2Model, gpt2_get_params_for_weight_decay_optimization
from cogview_amd.optim import FusedAdam
mpu.initialize_model_parallel(1); torch.manual_seed(1); mpu.model_parallel_cuda_manual_seed(1)
L, h, heads, b = 4, 1024, 16, 8
model = FP16_Module(GPT2Model(L, 58240, h, heads, 0.1, 0.1, 0.1, 1089, 0, False).cuda(), dtype=torch.bfloat16, keep_half_outputs=True)
groups = gpt2_get_params_for_weight_decay_optimization(model.module)
for g in groups:
    for p in g["params"]:
        if not hasattr(p, "model_parallel"): p.model_parallel = False
opt = FP16_Optimizer(FusedAdam(groups, lr=1e-4, weight_decay=0.01), dynamic_loss_scale=True,
                     dynamic_loss_args={"init_scale": 1.0, "scale_window": 1000, "min_scale": 1, "delayed_shift": 2})
model.train()
text = torch.randint(0, 58219, (b, 1089)).cuda(); lm = torch.ones(b, 1089, device="cuda")
batch = training.get_batch(text, lm)
for _ in range(2): training.train_step(batch, model, opt, clip_grad=1.0, log=False, world_size=1)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    training.train_step(batch, model, opt, clip_grad=1.0, log=False, world_size=1)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::zeros", "aten::add_", "aten::mul_", "aten::sum"):
        st = [s for s in (ev.stack or []) if "cogview_amd" in s or "bench" in s or "tools" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (name, site), n in cnt.most_common(40):
    print(f"{n:5d} {name:18s} {site}")
