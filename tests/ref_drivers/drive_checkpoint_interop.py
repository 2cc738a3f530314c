"""Checkpoint files cross the boundary in both directions through the REFERENCE's own utils.py (save_checkpoint :188-234,
load_checkpoint :289-380, unedited, imported from /root/reference).  Three stages, each its own process (the reference's and the
mirrors' `mpu` / `model` / `fp16` cannot live in one interpreter); tests/test_reference_drivers_cpu.py runs them in order:

  ref_save   REFERENCE stack (its own mpu / model, fp32 on the CPU, shims of oracle/gen_golden.py): seeded GPT2Model ->
             utils.save_checkpoint -> <dir>/ref ; prints the logits it computes for a fixed batch.
  mirror     MIRROR stack (INTEGRATION.md section 2 aliases, CPU-emulated ops): pretrain_gpt2.setup_model_and_optimizer, then
             utils.load_checkpoint(<dir>/ref, --finetune: weights only, as a release file is loaded) -> same logits as the
             reference printed; three train_steps; utils.save_checkpoint -> <dir>/mirror (weights + FP16_Optimizer state +
             AnnealingLR state); a FRESH model / optimizer, utils.load_checkpoint(<dir>/mirror) -> the next train_step equals the
             uninterrupted run's; prints the trained model's logits for the fixed batch.
             Also generate_samples.setup_model (:51-68) on the reference-written file, DeepSpeed-layout branch and plain branch.
  ref_load   REFERENCE stack: utils.load_checkpoint(<dir>/mirror) into its own fp32 GPT2Model -> same logits as the mirror
             printed: a file written over the mirrors is a reference checkpoint.

Scaffolding (no GPU): as tests/ref_drivers/drive_pretrain_gpt2.py; torch.cuda.get_rng_state / set_rng_state (which the reference's
rng block calls, utils.py:219-225, 363-368) are stand-ins in the mirror stage -- no kernel of this package reads that state; its
dropout streams are the (seed, offset) pairs of mpu.get_cuda_rng_tracker().get_states(), which the reference's block does save --
and the reference stages run with --no-save-rng / --no-load-rng (the two stacks' dropout states are not interchangeable);
TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD=1 in both stacks: the reference's `torch.load(name, map_location='cpu')` predates torch 2.6's
weights_only default, which refuses the loss-scaler object (and the numpy rng state) its own checkpoints carry."""
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
MODE, DIR = sys.argv[1], sys.argv[2]
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

os.environ["TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD"] = "1"
import torch

CFG = dict(layers=4, vocab=8704, hidden=256, heads=4, max_pos=64, rows=4, row_len=33, seed=77)


def fixed_batch():
    g = torch.Generator().manual_seed(CFG["seed"] + 5)
    return torch.randint(0, CFG["vocab"] - 12, (CFG["rows"], CFG["row_len"]), generator=g)


def base_args(**kw):
    a = types.SimpleNamespace(
        num_layers=CFG["layers"], vocab_size=CFG["vocab"], hidden_size=CFG["hidden"], num_attention_heads=CFG["heads"],
        hidden_dropout=0.0, attention_dropout=0.0, max_position_embeddings=CFG["max_pos"],
        max_position_embeddings_finetune=CFG["max_pos"], max_memory_length=0, checkpoint_activations=False, checkpoint_num_layers=1,
        query_window=128, key_window_times=6, num_pivot=768, deepspeed=False, fp16=True, cpu_optimizer=False, cpu_torch_adam=False,
        lr=1e-3, weight_decay=0.01, loss_scale=None, dynamic_loss_scale=True, loss_scale_window=1000, min_scale=1, hysteresis=2,
        lr_decay_iters=None, train_iters=100, warmup=0.0, lr_decay_style="linear", lr_decay_ratio=0.1, train_data=["synthetic"],
        finetune=False, is_sparse=0, txt_loss_scale=1.0, world_size=1, clip_grad=1.0, fp32_allreduce=False, iteration=0,
        save=None, load=None, no_save_optim=False, no_save_rng=True, no_load_optim=False, no_load_rng=True)
    a.__dict__.update(kw)
    return a


def logits_summary(logits):
    lg = logits.detach().float()
    probe = [(0, 0), (1, 7), (2, 19), (3, 31)]
    return {"norm": float(lg.double().norm()), "rows": [lg[b, t, :64].tolist() for b, t in probe]}


if MODE in ("ref_save", "ref_load"):
    from gen_golden import install_shims
    import gen_golden_generate as G
    mpu, st = install_shims()
    G.stub_modules(None)
    import utils as U                                        # the reference's utils.py
    assert os.path.realpath(U.__file__).startswith(REF + "/")
    from model.gpt2_modeling import GPT2Model
    torch.manual_seed(CFG["seed"])
    model = GPT2Model(CFG["layers"], CFG["vocab"], CFG["hidden"], CFG["heads"], 0.0, 0.0, 0.0, CFG["max_pos"], 0, False)
    model.eval()
    if MODE == "ref_save":
        U.save_checkpoint(1, model, None, None, base_args(save=os.path.join(DIR, "ref")))
    else:
        it = U.load_checkpoint(model, None, None, base_args(load=os.path.join(DIR, "mirror")))
        assert it == 3, it
    rows = fixed_batch()
    tokens = rows[:, :-1].contiguous()
    s = tokens.shape[1]
    with torch.no_grad():
        logits, = model(tokens, torch.arange(s).unsqueeze(0).expand_as(tokens), torch.tril(torch.ones(1, 1, s, s)), None, None, 0)
    print("RESULT " + json.dumps({"mode": MODE, "logits": logits_summary(logits)}), flush=True)
    sys.exit(0)

assert MODE == "mirror"
sys.path.append(REF)
# ---- no-GPU scaffolding + INTEGRATION.md section 2 (see drive_pretrain_gpt2.py)
torch.Tensor.is_cuda = property(lambda self: True)
torch.nn.Module.cuda = lambda self, device=None: self
torch.cuda.current_device = lambda: 0
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.get_rng_state = lambda *a, **k: torch.zeros(16, dtype=torch.uint8)
torch.cuda.set_rng_state = lambda *a, **k: None
import cpu_ops
cpu_ops.install()
import cogview_amd
cogview_amd.bind_reference_names()
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
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % (29100 + os.getpid() % 150), world_size=1, rank=0)
import mpu
mpu.initialize_model_parallel(1)
import pretrain_gpt2 as P
import utils as U
for mod in (P, U):
    assert os.path.realpath(mod.__file__).startswith(REF + "/"), mod.__file__

rows = fixed_batch()


def batches():
    while True:
        yield {"text": rows.clone(), "loss_mask": torch.ones_like(rows)}


def eval_logits(model):
    tokens = rows[:, :-1].contiguous()
    s = tokens.shape[1]
    model.eval()
    with torch.no_grad():
        logits, *_ = model(tokens, torch.arange(s).unsqueeze(0).expand_as(tokens), torch.tril(torch.ones(1, 1, s, s)), None, None, 0)
    model.train()
    return logits


def fresh(seed):
    torch.manual_seed(seed)
    mpu.model_parallel_cuda_manual_seed(seed)
    torch.manual_seed(seed)
    return P.setup_model_and_optimizer(base_args())


out = {"mode": MODE}
timers = U.Timers()
# 1. a reference-written file, loaded the way a release file is (weights only; the fp32 masters must follow the loaded weights)
model, optimizer, lr_scheduler = fresh(1)                    # other weights than the file's: the load must replace every tensor
it = U.load_checkpoint(model, optimizer, lr_scheduler, base_args(load=os.path.join(DIR, "ref"), finetune=True))
assert it == 0                                               # --finetune restarts the iteration count (utils.py:343-344)
out["logits_after_loading_reference_file"] = logits_summary(eval_logits(model))

# 2. three steps (the first at a loss scale fp16 gradients can carry), then save through the reference's save_checkpoint
optimizer.loss_scaler.cur_scale = 2.0 ** 12
args, data = base_args(save=os.path.join(DIR, "mirror")), batches()
losses = []
for _ in range(3):
    lm, skipped, *_ = P.train_step(data, model, optimizer, lr_scheduler, args, timers, [])
    assert skipped == 0
    losses.append(float(lm.detach()))
# ... with the rng block (the default): python / numpy / torch generators and the dropout states -- the model-parallel tracker's and,
# under a reserved name, the default one -- as plain tensors
args.no_save_rng = False
mpu.random.manual_seed(4242)
for _ in range(5):
    mpu.random.next_dropout_stream()
tracker_before = {k: v.tolist() for k, v in mpu.get_cuda_rng_tracker().get_states().items()}
U.save_checkpoint(3, model, optimizer, lr_scheduler, args)
mpu.random.manual_seed(1)                                     # (the resumed process starts from other states)
out["losses"] = losses
out["logits_of_saved_model"] = logits_summary(eval_logits(model))
lm4, skipped, *_ = P.train_step(data, model, optimizer, lr_scheduler, args, timers, [])
out["step4_uninterrupted"] = {"loss": float(lm4.detach()), "lr_steps": lr_scheduler.num_iters, "adam_steps": optimizer._step_count,
                              "scale": float(optimizer.loss_scale)}

# 3. resume: a fresh model / optimizer / scheduler, the reference's load_checkpoint, the same step 4
model2, optimizer2, lr2 = fresh(2)
it = U.load_checkpoint(model2, optimizer2, lr2, base_args(load=os.path.join(DIR, "mirror"), no_load_rng=False))
assert it == 3, it
tracker_after = {k: v.tolist() for k, v in mpu.get_cuda_rng_tracker().get_states().items()}
out["dropout_states_restored"] = tracker_after == tracker_before and tracker_after["cogview-amd-default-dropout-state"] == [4242, 5]
lm4b, skipped, *_ = P.train_step(batches(), model2, optimizer2, lr2, args, timers, [])
out["step4_resumed"] = {"loss": float(lm4b.detach()), "lr_steps": lr2.num_iters, "adam_steps": optimizer2._step_count,
                        "scale": float(optimizer2.loss_scale)}
# 4. generate_samples.setup_model (:51-68), both of its branches, on the reference-written file: the DeepSpeed-layout branch
#    (:56-61: reads <load>/<iteration>/mp_rank_00_model_states.pt itself, checkpoint["module"]) and the plain one (load_checkpoint)
tv, tvu = types.ModuleType("torchvision"), types.ModuleType("torchvision.utils")
tvu.save_image = lambda *a, **k: None
tv.utils = tvu
sys.modules["torchvision"], sys.modules["torchvision.utils"] = tv, tvu
import generate_samples as S
assert os.path.realpath(S.__file__).startswith(REF + "/")
for branch, ds_flag in (("deepspeed_layout", True), ("plain", False)):
    torch.manual_seed(3)
    m = S.setup_model(base_args(load=os.path.join(DIR, "ref"), deepspeed=ds_flag))
    out["setup_model_" + branch] = logits_summary(eval_logits(m))

w_a = torch.cat([p.detach().float().view(-1) for p in model.parameters()])
w_b = torch.cat([p.detach().float().view(-1) for p in model2.parameters()])
out["weights_after_step4_equal"] = bool(torch.equal(w_a, w_b))
out["weights_after_step4_maxdiff"] = float((w_a - w_b).abs().max())
print("RESULT " + json.dumps(out), flush=True)
