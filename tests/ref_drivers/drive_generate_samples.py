"""Run the REFERENCE's own generation driver -- generate_samples.py: prepare_tokenizer :268-287, setup_model :51-68 (->
pretrain_gpt2.get_model), generate_images_once :142-203, post_selection :237-263 -- and, under it, the reference's own
generation/sampling.py (filling_sequence :65-201, inverse_prompt_score :222-239, get_batch :53-63 -> pretrain_gpt2.
get_masks_and_position_ids), all UNEDITED and imported from /root/reference, over the `cogview_amd` mirrors bound as
INTEGRATION.md section 2 prescribes.  Executed by tests/test_reference_drivers_cpu.py in a subprocess; build container only.

What it proves: GPT2Model.forward(tokens, position_ids, attention_mask, txt_indices_bool, img_indices_bool, is_sparse, *mems)
-> (logits, *mems) of the mirror is the reference's, as its generation code uses it: the context pass with a [1, 1, s, s] mask,
then single-token passes with `attention_mask = 0` over growing memories (one per layer + the embedding output), beams expanded
with mem.expand, logits edited in place by the caller (temperature, forbidden id ranges, top-k), and the score of a full 1037-token
row.  Expected tokens / scores: tests/golden/generate_samples.npz, which oracle/gen_golden_generate.py produced by running the
reference's own fp32 model under the same functions.

No GPU here, so (test scaffolding, all listed):
  * cogview_amd.ops' entry points are replaced by tests/cpu_ops.py; `model.cuda(...)` is the identity, torch.cuda.current_device()
    -> 0, torch.cuda.empty_cache() -> no-op, tensors answer is_cuda = True;
  * not installed and not on the path under test: deepspeed, tensorboardX, torchvision (save_image writes a marker file: the
    script chmods what it saved), and the reference's data_utils (lmdb, torchvision, sentencepiece model files): the toy
    tokenizer of oracle/gen_golden_generate.py stands in -- generate_samples reads only its id ranges, its marker ids and
    DecodeIds (the VQ-VAE decoder: exercised on the GPU by tests/test_vqvae_gpu.py, not here)."""
import json
import os
import sys
import tempfile
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.append(REF)                       # arguments.py, utils.py, pretrain_gpt2.py, generate_samples.py, generation/ resolve here

import numpy as np
import torch

# ---- no-GPU scaffolding
torch.Tensor.is_cuda = property(lambda self: True)
torch.nn.Module.cuda = lambda self, device=None: self
torch.cuda.current_device = lambda: 0
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda: None
import cpu_ops
cpu_ops.install()

# ---- INTEGRATION.md section 2, verbatim: one call binds mpu / model / fp16 / vqvae / apex.optimizers
import cogview_amd
cogview_amd.bind_reference_names()

# ---- stand-ins for what is not installed / not on the path under test
import gen_golden_generate as G                              # the toy tokenizer + the scenario the golden was generated on
ds = types.ModuleType("deepspeed")
ds.add_config_arguments = lambda parser: parser
sys.modules["deepspeed"] = ds
tbx = types.ModuleType("tensorboardX")
tbx.SummaryWriter = type("SummaryWriter", (), {"__init__": lambda self, *a, **k: None, "add_scalar": lambda self, *a, **k: None})
sys.modules["tensorboardX"] = tbx
tok = G.ToyTokenizer()
du = types.ModuleType("data_utils")
du.get_tokenizer = lambda args=None: tok
du.make_loaders = du.detect_new_datasets = lambda *a, **k: None
sys.modules["data_utils"] = du
saved = []
tv, tvu = types.ModuleType("torchvision"), types.ModuleType("torchvision.utils")


def save_image(tensor, path, **kw):
    saved.append((os.path.basename(path), tuple(tensor.shape)))
    open(path, "wb").close()


tvu.save_image = save_image
tv.utils = tvu
sys.modules["torchvision"], sys.modules["torchvision.utils"] = tv, tvu

import torch.distributed as dist
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % (29300 + os.getpid() % 300), world_size=1, rank=0)
import mpu                                                  # == cogview_amd.mpu
mpu.initialize_model_parallel(1)

import generate_samples as S                                # the reference's script, unedited
import generation.sampling as RS                            # the reference's sampling code, unedited
import pretrain_gpt2 as P
for mod in (S, RS, P):
    assert os.path.realpath(mod.__file__).startswith(REF + "/"), mod.__file__
assert S.GPT2Model is cogview_amd.model.GPT2Model and S.FP16_Module is cogview_amd.fp16.FP16_Module
assert S.filling_sequence is RS.filling_sequence and S.get_model is P.get_model

c = G.CFG
args = types.SimpleNamespace(
    num_layers=c["layers"], hidden_size=c["hidden"], num_attention_heads=c["heads"], hidden_dropout=0.1, attention_dropout=0.1,
    max_position_embeddings=c["max_pos"], max_position_embeddings_finetune=c["max_pos"], max_memory_length=c["max_mem"],
    checkpoint_activations=False, checkpoint_num_layers=1, query_window=128, key_window_times=6, num_pivot=768, deepspeed=False,
    fp16=True, load=None, finetune=False, make_vocab_size_divisible_by=c["divisible_by"], seed=c["seed"],
    is_sparse=0, temperature=1.0, top_k=1, top_p=0.0, generation_task="text2image", max_inference_batch_size=c["beams"], debug=False)

P.set_random_seed(args.seed)                                # generate_samples.py:308
S.prepare_tokenizer(args)                                   # :311 -> args.vocab_size, padded
torch.manual_seed(args.seed)                                # the seed the golden's constructor ran under
model = S.setup_model(args)                                 # :314 -> pretrain_gpt2.get_model over the mirrors
assert isinstance(model, P.DDP) and isinstance(model.module, S.FP16_Module)

gold = np.load(os.path.join(ROOT, "tests", "golden", "generate_samples.npz"))
assert args.vocab_size == int(gold["vocab"])
t2i, sel = torch.from_numpy(gold["t2i_seq"]), torch.from_numpy(gold["sel_seq"])
out = {"vocab": args.vocab_size}

with tempfile.TemporaryDirectory() as tmp:
    # text -> image: generate_images_once hands the marked sequence to filling_sequence, decodes every beam, saves one image per beam
    # + the concatenation
    S.generate_images_once(model, args, "five text pieces", seq=t2i.clone(), num=c["beams"], output_path=os.path.join(tmp, "t2i"))
    out["t2i_tokens"] = [list(map(int, ids)) for ids in tok.decoded]
    out["saved"] = [name for name, _ in saved]
    out["saved_concat_shape"] = list(saved[-1][1])
    # post-selection: inverse_prompt_score of two candidate rows, written to scores_rank_0.txt
    args.generation_task = "post-selection"
    S.post_selection(model, args, "two candidates", sel.clone(), os.path.join(tmp, "sel"))
    lines = open(os.path.join(tmp, "sel", "scores_rank_0.txt")).read().splitlines()
    out["sel_text"], out["sel_scores"] = lines[0], [float(x) for x in lines[1].split("\t")]

# sparse generation (is_sparse = 2): a model with a 2 x 16-position trailing window; every layer of every pass samples its pivots
# with `random.sample`, so `random` is seeded as the golden's generator seeded it
import random
sp = G.SPARSE
args.query_window, args.key_window_times, args.num_pivot = sp["query_window"], sp["key_window_times"], sp["num_pivot"]
args.is_sparse, args.max_inference_batch_size = 2, 1
torch.manual_seed(sp["seed"])
smodel = S.setup_model(args)
tok.decoded.clear()
with tempfile.TemporaryDirectory() as tmp:
    random.seed(sp["random_seed"])
    S.generate_images_once(smodel, args, "five text pieces, sparse", seq=torch.from_numpy(gold["sparse_seq"]).clone(), num=1,
                           output_path=os.path.join(tmp, "sparse"))
out["sparse_tokens"] = [list(map(int, ids)) for ids in tok.decoded]
args.is_sparse = 0

# the mirror's generation functions against the reference's, on the same model / stand-in
import generation.magnify                                   # (`generation.magnify` the attribute is the function: __init__ re-exports it)
RM = sys.modules["generation.magnify"]
import cogview_amd.generation as MG
from generation_cases import ToyIds
from test_generation_cpu import _PositionalOracle
assert os.path.realpath(RM.__file__).startswith(REF + "/")
args.generation_task = "text2image"
with torch.no_grad():
    ours = MG.inverse_prompt_score(model, sel.clone(), args, tokenizer=tok)
    theirs = RS.inverse_prompt_score(model, sel.clone(), args)
out["score_mirror_vs_reference_fn"] = float((ours.float() - theirs.float()).abs().max())
g = torch.Generator().manual_seed(5)
small = torch.randint(0, c["img_tokens"], (1024,), generator=g)
text = torch.cat([torch.tensor([tok["[ROI1]"]]), torch.randint(c["img_tokens"], c["img_tokens"] + c["txt_tokens"], (4,), generator=g),
                  torch.tensor([tok["[BASE]"], tok["[BOI1]"]])])
ref_big = RM.magnify(_PositionalOracle(c["img_tokens"], args.vocab_size), tok, small, text, args)
our_big = MG.magnify(_PositionalOracle(c["img_tokens"], args.vocab_size), ToyIds(c["img_tokens"], c["txt_tokens"]), small, text, args)
out["magnify_equal"] = bool(torch.equal(ref_big, our_big))
out["magnify_shape"] = list(ref_big.shape)
print("RESULT " + json.dumps(out), flush=True)
