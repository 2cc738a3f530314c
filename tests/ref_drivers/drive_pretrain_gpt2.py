"""Run the REFERENCE's own driver functions -- pretrain_gpt2.py: the main loop train :482-566 (-> report_iteration_metrics,
utils.save_checkpoint, evaluate_and_print_results -> evaluate :569-607), setup_model_and_optimizer (get_model :58-107,
get_optimizer_param_groups :110-122, get_optimizer :125-157, get_learning_rate_scheduler :160-179), get_batch :258-288,
forward_step :292-341, backward_step :344-391, train_step :406-448 -- UNEDITED, imported from /root/reference, over the
`cogview_amd` mirrors bound exactly as INTEGRATION.md section 2 prescribes (sys.modules aliases for mpu / model / fp16 / vqvae /
apex.optimizers).  Executed by tests/test_reference_drivers_cpu.py in a subprocess (the aliases must not leak into the test
session); build container only: the GPU box has no /root/reference.

No GPU here, so (test scaffolding, all listed):
  * cogview_amd.ops' entry points are replaced by tests/cpu_ops.py (torch-CPU restatement of the kernels' semantics);
  * `model.cuda(...)` is the identity, torch.cuda.current_device() -> 0, torch.cuda.synchronize() -> no-op (utils.Timers), and
    tensors answer is_cuda = True (the mirrors -- like the reference's FP16_Optimizer -- refuse CPU half parameters);
  * not installed and not on the path under test: deepspeed (the script takes its non-DeepSpeed branch), tensorboardX (logging),
    and the reference's data_utils (needs lmdb + torchvision; forward_step only asks its tokenizer for the image / text id
    split, data_utils/unified_tokenizer.py:32-33: 8192 image codes) -- stand-ins of a few lines each.
Workload: BASELINE.json configs[0] (4 layers / 256 hidden / 4 heads, vocabulary 58240, 4 rows of 256 tokens), the rows and
the expected loss / gradient norm from tests/golden/gpt2_cfg1.npz, which oracle/gen_golden_cfg1.py produced by running the
reference's OWN modules in fp32: the reference's train_step over the mirrors must land on the same loss."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.append(REF)                       # arguments.py, learning_rates.py, utils.py, pretrain_gpt2.py resolve here

import numpy as np
import torch

# ---- no-GPU scaffolding
torch.Tensor.is_cuda = property(lambda self: True)
torch.nn.Module.cuda = lambda self, device=None: self
torch.cuda.current_device = lambda: 0
torch.cuda.synchronize = lambda *a, **k: None
import cpu_ops
cpu_ops.install()

# ---- INTEGRATION.md section 2, verbatim: one call binds mpu / model / fp16 / vqvae / apex.optimizers
import cogview_amd
cogview_amd.bind_reference_names()

# ---- stand-ins for what is not installed / not on the path under test
ds = types.ModuleType("deepspeed")
ds.add_config_arguments = lambda parser: parser
sys.modules["deepspeed"] = ds
tbx = types.ModuleType("tensorboardX")
tbx.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None, "add_scalar": lambda self, *a, **k: None})
sys.modules["tensorboardX"] = tbx
du = types.ModuleType("data_utils")
_tok = types.SimpleNamespace(img_tokenizer=types.SimpleNamespace(num_tokens=8192))
du.get_tokenizer = lambda args=None: _tok
du.make_loaders = du.detect_new_datasets = lambda *a, **k: None
sys.modules["data_utils"] = du

import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % (29600 + os.getpid() % 300), world_size=1, rank=0)
import mpu                                                  # == cogview_amd.mpu
mpu.initialize_model_parallel(1)

import pretrain_gpt2 as P                                   # the reference's script, unedited
assert os.path.realpath(P.__file__).startswith(REF + "/"), P.__file__
from utils import Timers                                    # the reference's utils.py
assert P.GPT2Model is cogview_amd.model.GPT2Model and P.FP16_Optimizer is cogview_amd.fp16.FP16_Optimizer
assert P.DDP is cogview_amd.model.PyTorchDistributedDataParallel and P.Adam is cogview_amd.optim.FusedAdam

gold = np.load(os.path.join(ROOT, "tests", "golden", "gpt2_cfg1.npz"))
rows = torch.from_numpy(gold["rows"])
args = types.SimpleNamespace(
    num_layers=4, vocab_size=58240, hidden_size=256, num_attention_heads=4,
    hidden_dropout=float(os.environ.get("COGV_DRV_DROPOUT", "0")), attention_dropout=float(os.environ.get("COGV_DRV_DROPOUT", "0")),
    max_position_embeddings=256, max_position_embeddings_finetune=256, max_memory_length=0,
    checkpoint_activations=os.environ.get("COGV_DRV_CHECKPOINT_ACTIVATIONS") == "1",      # --checkpoint-activations (the reference's scripts set it)
    checkpoint_num_layers=1, query_window=128, key_window_times=6, num_pivot=768, deepspeed=False, fp16=True,
    cpu_optimizer=False, cpu_torch_adam=False, lr=1.5e-4, weight_decay=0.01, loss_scale=None, dynamic_loss_scale=True,
    loss_scale_window=1000, min_scale=1, hysteresis=2, lr_decay_iters=None, train_iters=100, warmup=0.01,
    lr_decay_style="linear", lr_decay_ratio=0.1, train_data=["synthetic"], finetune=False, is_sparse=0, txt_loss_scale=1.0,
    world_size=1, clip_grad=1.0, fp32_allreduce=False, iteration=0)

torch.manual_seed(1234)                                     # arguments.py:123 default seed; what the golden's generator used
mpu.model_parallel_cuda_manual_seed(1234)
torch.manual_seed(1234)
model, optimizer, lr_scheduler = P.setup_model_and_optimizer(args)
assert isinstance(model, P.DDP) and isinstance(model.module, P.FP16_Module) and isinstance(optimizer, P.FP16_Optimizer)
assert optimizer._arena is not None, "the fused flat-arena optimizer path is the one under test"


def batches():
    while True:
        yield {"text": rows.clone(), "loss_mask": torch.ones_like(rows)}


it, timers = batches(), Timers()
out = {}
w0 = model.module.module.word_embeddings.weight.detach().float().clone()

# step 1: the reference's default dynamic loss scale (2^32) overflows fp16 gradients -> train_step reports a skipped iteration
lm, skipped, mems, img_loss, txt_loss = P.train_step(it, model, optimizer, lr_scheduler, args, timers, [])
out["step1"] = {"loss": float(lm.detach()), "skipped": int(skipped), "scale_after": float(optimizer.loss_scale), "lr_steps": lr_scheduler.num_iters,
                "params_unchanged": bool(torch.equal(w0, model.module.module.word_embeddings.weight.detach().float()))}

# step 2 at a loss scale fp16 can carry: gradients are real, but AnnealingLR's warm-up starts at lr = 0 (learning_rates.py:
# iteration 0 of 1 warm-up step), so this update moves nothing -- exactly as in the reference -- and the scheduler advances
optimizer.loss_scaler.cur_scale = 2.0 ** 12
lm, skipped, mems, img_loss, txt_loss = P.train_step(it, model, optimizer, lr_scheduler, args, timers, [])
gn = (optimizer._host_stats[0] ** 0.5) / (2.0 ** 12)
out["step2"] = {"loss": float(lm.detach()), "skipped": int(skipped), "grad_norm": gn, "lr_steps": lr_scheduler.num_iters,
                "img_loss": float(img_loss), "txt_loss": float(txt_loss), "lr_used": 0.0, "lr_next": optimizer.param_groups[0]["lr"],
                "params_unchanged": bool(torch.equal(w0, model.module.module.word_embeddings.weight.detach().float()))}
out["golden"] = {"loss": float(gold["loss"]), "grad_norm": float(gold["grad_norm"])}

# step 3: the first update with lr > 0 (same loss going in: the parameters had not moved); step 4 sees its effect
lm, skipped, *_ = P.train_step(it, model, optimizer, lr_scheduler, args, timers, [])
w1 = model.module.module.word_embeddings.weight.detach().float()
out["step3"] = {"loss": float(lm.detach()), "skipped": int(skipped), "max_param_change": float((w1 - w0).abs().max()),
                "adam_steps": optimizer._step_count}
lm, skipped, *_ = P.train_step(it, model, optimizer, lr_scheduler, args, timers, [])
out["step4"] = {"loss": float(lm.detach()), "skipped": int(skipped), "lr_steps": lr_scheduler.num_iters}

# the reference's main loop itself -- pretrain_gpt2.train (:482-566): train_step, the logging block (report_iteration_metrics,
# timers.log), save_checkpoint every second iteration (the reference's utils.py) and evaluate_and_print_results -> evaluate
# (:569-607: model.eval(), forward_step under no_grad, model.train()) on the same rows.  report_memory reads the CUDA allocator:
# stubbed.  No rng block in the file (torch.cuda.get_rng_state; see drive_checkpoint_interop.py).
import io
import tempfile
from contextlib import redirect_stdout
P.report_memory = lambda name: None
with tempfile.TemporaryDirectory() as tmp:
    args.iteration, args.train_iters = 4, 8
    args.log_interval, args.save, args.save_interval, args.eval_interval, args.eval_iters, args.do_valid = 2, tmp, 2, 4, 2, True
    args.no_save_optim, args.no_save_rng, args.exit_interval = False, True, None
    buf = io.StringIO()
    with redirect_stdout(buf):
        P.train(model, optimizer, lr_scheduler, it, batches(), timers, args)
    log = buf.getvalue()
    out["train_loop"] = {
        "iteration": args.iteration, "lr_steps": lr_scheduler.num_iters, "adam_steps": optimizer._step_count,
        "saved": sorted(d for d in os.listdir(tmp) if d.isdigit()), "tracker": open(os.path.join(tmp, "latest_checkpointed_iteration.txt")).read(),
        "logged_iterations": [int(l.split("iteration")[1].split("/")[0]) for l in log.splitlines() if l.startswith(" iteration")],
        "lm_losses": [float(l.split("lm loss")[1].split("|")[0]) for l in log.splitlines() if l.startswith(" iteration")],
        "validation": [float(l.split("LM loss:")[1].split("|")[0]) for l in log.splitlines() if "validation loss at" in l],
        "training_mode_restored": bool(model.training)}
    lm_eval = P.evaluate(batches(), model, args, timers)
    out["train_loop"]["evaluate_again"] = lm_eval
print("RESULT " + json.dumps(out), flush=True)
