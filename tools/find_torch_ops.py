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
import traceback
from torch.utils._python_dispatch import TorchDispatchMode
cnt = collections.Counter(); size = collections.Counter()


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(k in name for k in ("copy_", "fill_", "zero_", "clone", "zeros", "_to_copy", "sum", "add", "mul", "cat", "ones", "full")):
            st = [f for f in traceback.extract_stack() if "cogview_amd" in f.filename or "tools" in f.filename]
            site = f"{os.path.relpath(st[-1].filename, ROOT)}:{st[-1].lineno}" if st else "?"
            t = out if isinstance(out, torch.Tensor) else (args[0] if args and isinstance(args[0], torch.Tensor) else None)
            cnt[(name, site)] += 1
            if t is not None and t.is_cuda:
                size[(name, site)] += t.numel() * t.element_size()
        return out


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with Log():
    training.train_step(batch, model, opt, clip_grad=1.0, log=False, world_size=1)
    torch.cuda.synchronize()
for (name, site), n in cnt.most_common(60):
    print(f"{n:5d} {name:28s} {size[(name, site)] / 1e6:10.2f} MB  {site}")
