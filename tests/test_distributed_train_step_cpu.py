"""CPU, two (and four) processes over gloo, the WHOLE train step on the CPU-emulated ops (tests/cpu_ops.py): what the N > 1 legs of
`bench.py` run -- data parallel with the bucketed exchange issued as each layer's backward finishes (the weight-gradient queue
releases a layer's bucket one launch late), the sharded form (reduce-scatter, owned-slice AdamW, all-gather), and two-way
tensor parallel (column / row parallel linears, head-sharded attention, vocab-parallel embedding / logits / cross entropy),
and model parallel 2 x data parallel 2 on four ranks.
The GPU versions of the same three checks are tests/test_model_gpu.py::test_two_rank_* / test_two_way_tensor_parallel_on_one_gpu;
these run where the CPU suite runs.

Scaffolding (no GPU): tensors answer is_cuda = True and torch.cuda's stream / event objects are inert stand-ins (the exchange is
issued on a side stream on a GPU; on the CPU every collective completes in program order)."""
import contextlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


class _Inert:
    def __init__(self, *a, **k):
        pass

    def wait_stream(self, s):
        pass

    def wait_event(self, e):
        pass

    def record(self, s=None):
        pass

    def synchronize(self):
        pass

    def query(self):
        return True


def _cpu_scaffolding(world):
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))      # one process per rank on the same cores
    sys.path.insert(0, ROOT)
    from tests import cpu_ops
    cpu_ops.install()
    torch.Tensor.is_cuda = property(lambda self: True)
    torch.cuda.Stream = _Inert
    torch.cuda.Event = _Inert
    torch.cuda.current_stream = lambda *a, **k: _Inert()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.current_device = lambda: 0


def _entry(rank, world, port, fn_name, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    try:
        _cpu_scaffolding(world)
        ret[rank] = ("ok", globals()[fn_name](rank, world))
    except Exception:
        import traceback
        ret[rank] = (traceback.format_exc(), None)
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_entry, args=(r, world, port, fn_name, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
        out = []
        for r in range(world):
            assert ret.get(r) is not None and ret[r][0] == "ok", f"rank {r}: {ret.get(r)}"
            out.append(ret[r][1])
        return out


def _golden():
    z = np.load(os.path.join(GOLDEN, "gpt2_small.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def _build(g, dtype=torch.float16):
    from cogview_amd.fp16 import FP16_Module
    from cogview_amd.model import GPT2Model
    L_, V_, H_, NH_, P_, S_, B_ = [int(v) for v in g["cfg"]]
    torch.manual_seed(0)
    m = GPT2Model(L_, V_, H_, NH_, 0.0, 0.0, 0.0, P_, 0, False)
    m.load_state_dict({k[6:]: v for k, v in g.items() if k.startswith("param.")})
    return FP16_Module(m, dtype=dtype, keep_half_outputs=True)


def _optimizer(model):
    from cogview_amd.fp16 import FP16_Optimizer
    from cogview_amd.model import gpt2_get_params_for_weight_decay_optimization
    from cogview_amd.optim import FusedAdam
    groups = gpt2_get_params_for_weight_decay_optimization(model.module)
    for grp in groups:
        for p in grp["params"]:
            if not hasattr(p, "model_parallel"):
                p.model_parallel = False
    return FP16_Optimizer(FusedAdam(groups, lr=1e-3, weight_decay=0.01), dynamic_loss_scale=True,
                          dynamic_loss_args={"init_scale": 2 ** 10, "scale_window": 100, "min_scale": 1, "delayed_shift": 1})


# ------------------------------------------------------------------------------------------------ data parallel, 2 ranks
def _dp_step(rank, world, shard):
    from cogview_amd import mpu, training
    from cogview_amd.model import PyTorchDistributedDataParallel
    mpu.initialize_model_parallel(1)
    g = _golden()
    S_, B_ = int(g["cfg"][5]), int(g["cfg"][6])
    half = B_ // world
    sl = slice(rank * half, (rank + 1) * half)
    model = _build(g)
    ddp = PyTorchDistributedDataParallel(model, process_group=mpu.get_data_parallel_group(), bucket_layers=1, shard_optimizer=shard)
    assert ddp.overlap and len(ddp._buckets) == 2 and (ddp.shard is not None) == shard
    opt = _optimizer(model)
    opt.attach_data_parallel(ddp)
    launched = []
    real_launch = ddp._launch
    ddp._launch = lambda s, e: (launched.append((s, e)), real_launch(s, e))[1]
    pos = torch.arange(S_).unsqueeze(0).expand(half, -1)
    batch = (g["tokens"][sl], g["labels"][sl], torch.ones_like(g["loss_mask"][sl]), 0, pos)
    before = model.module._cogv_arena.data.detach().float().clone()
    loss, _, _, _ = training.forward_step(batch, ddp, log=False, world_size=world)
    training.backward_step(opt, ddp, loss, 1.0)
    assert len(launched) == 2, launched                      # both layer buckets left DURING backward, not at the end
    grads = None if shard else (model.module._cogv_arena.grad.detach().float() / opt.loss_scale)
    opt.step()
    assert not opt.overflow
    flat = model.module._cogv_arena.data.detach().float()
    parts = [torch.empty_like(flat) for _ in range(world)]
    dist.all_gather(parts, flat)
    assert torch.equal(parts[0], parts[1]), "replicas diverged after one data-parallel step"
    assert not torch.equal(flat, before)
    return {"grads": grads, "params": flat, "loss": float(loss.detach())}


def w_dp2(rank, world):
    return _dp_step(rank, world, shard=False)


def w_dp2_sharded(rank, world):
    return _dp_step(rank, world, shard=True)


def w_dp2_reference_style_no_optimizer(rank, world):
    """A wrapper constructed the way the reference constructs torch's DDP (device_ids=[i], output_device=i) and NO optimizer that
    would finish the exchange (plain loss.backward()): the callback queued on the autograd engine finishes it at the end of the
    pass -- the ranks hold the same, averaged gradients when backward() returns."""
    from cogview_amd import mpu, training
    from cogview_amd.model import PyTorchDistributedDataParallel
    mpu.initialize_model_parallel(1)
    g = _golden()
    S_, B_ = int(g["cfg"][5]), int(g["cfg"][6])
    half = B_ // world
    sl = slice(rank * half, (rank + 1) * half)
    model = _build(g)
    ddp = PyTorchDistributedDataParallel(model, device_ids=[0], output_device=0, process_group=mpu.get_data_parallel_group())
    assert ddp.auto_sync and not ddp._sync_consumer
    pos = torch.arange(S_).unsqueeze(0).expand(half, -1)
    batch = (g["tokens"][sl], g["labels"][sl], torch.ones_like(g["loss_mask"][sl]), 0, pos)
    loss, _, _, _ = training.forward_step(batch, ddp, log=False, world_size=world)
    (loss * 1024.0).backward()
    assert not ddp.needs_reduction and ddp._pending == []
    grads = model.module._cogv_arena.grad.detach().float() / 1024.0
    parts = [torch.empty_like(grads) for _ in range(world)]
    dist.all_gather(parts, grads)
    assert torch.equal(parts[0], parts[1]), "gradients differ across the ranks after backward()"
    return {"grads": grads, "params": None, "loss": float(loss.detach())}


def test_reference_style_wrapper_finishes_the_exchange_at_the_end_of_backward(one_rank):
    out = _run("w_dp2_reference_style_no_optimizer")
    assert rel(out[0]["grads"], one_rank["grads"]) < 2e-3


def _one_rank_whole_batch():
    """The same step in ONE process on the whole batch (this process: a one-rank gloo group)."""
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
    from cogview_amd import mpu, training
    if not mpu.model_parallel_is_initialized():
        mpu.initialize_model_parallel(1)
    g = _golden()
    S_, B_ = int(g["cfg"][5]), int(g["cfg"][6])
    model = _build(g)
    opt = _optimizer(model)
    pos = torch.arange(S_).unsqueeze(0).expand(B_, -1)
    batch = (g["tokens"], g["labels"], torch.ones_like(g["loss_mask"]), 0, pos)
    loss, _, _, _ = training.forward_step(batch, model, log=False)
    training.backward_step(opt, model, loss, 1.0)
    grads = model.module._cogv_arena.grad.detach().float() / opt.loss_scale
    opt.step()
    return {"grads": grads, "params": model.module._cogv_arena.data.detach().float(), "loss": float(loss.detach())}


@pytest.fixture()
def one_rank(monkeypatch):
    from tests import cpu_ops
    cpu_ops.install(monkeypatch.setattr)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    return _one_rank_whole_batch()


@pytest.mark.parametrize("worker", ["w_dp2", "w_dp2_sharded"])
def test_two_rank_data_parallel_train_step(one_rank, worker):
    """Two replicas, half the batch each: bit-identical replicas after the step; the averaged gradients and the updated
    parameters equal those of one process on the whole batch to 16-bit round-off (the mean of two half-batch means is the whole
    batch's mean; the halves were rounded to fp16 before they were averaged)."""
    out = _run(worker)
    two = out[0]
    if two["grads"] is not None:
        e = rel(two["grads"], one_rank["grads"])
        assert e < 2e-3, e
    # the first AdamW update is ~ lr * sign(g) per element, so a gradient whose sign differs by round-off moves its weight by
    # 2 lr the other way: compare the UPDATES, loosely, and the weights at the scale of one update
    start = _golden_flat_like(one_rank["params"])
    assert float((one_rank["params"] - start).abs().max()) > 5e-4
    assert rel(two["params"] - start, one_rank["params"] - start) < 5e-2
    assert rel(two["params"], one_rank["params"]) < 1e-3
    assert abs(0.5 * (out[0]["loss"] + out[1]["loss"]) - one_rank["loss"]) < 2e-3 * one_rank["loss"]


def _golden_flat_like(flat):
    g = _golden()
    model = _build(g)
    return model.module._cogv_arena.data.detach().float()


# ------------------------------------------------------------------------------------------------ tensor parallel, 2 ranks
_TP = dict(L=2, V=512, H=256, NH=4, S=64, B=2)


def _tp_build_and_run(c=None, hidden_dropout=0.0):
    from cogview_amd import mpu
    from cogview_amd.fp16 import FP16_Module
    from cogview_amd.model import GPT2Model
    c = c or _TP
    torch.manual_seed(4321)
    mpu.model_parallel_cuda_manual_seed(4321)
    m = GPT2Model(c["L"], c["V"], c["H"], c["NH"], 0.0, 0.0, hidden_dropout, c["S"] + 1, 0, False)
    model = FP16_Module(m, dtype=torch.float16, keep_half_outputs=True)
    model.train()
    g = torch.Generator().manual_seed(11)
    tokens = torch.randint(0, c["V"], (c["B"], c["S"]), generator=g)
    labels = torch.randint(0, c["V"], (c["B"], c["S"]), generator=g)
    pos = torch.arange(c["S"]).unsqueeze(0).expand(c["B"], -1)
    logits, = model(tokens, pos, 0, None, None, 0)
    loss = mpu.vocab_parallel_cross_entropy(logits.contiguous().float(), labels).mean()
    (loss * 256.0).backward()
    return model, logits, loss


def _tp_slice(name, t, rank, world):
    """The shard of the full tensor `t` that model-parallel rank `rank` owns (None: replicated); mpu/layers.py:42-74."""
    if name.endswith("word_embeddings.weight") or "dense_h_to_4h" in name:
        return t.chunk(world, 0)[rank]
    if "query_key_value" in name:
        slabs = t.chunk(3 * world, 0)
        return torch.cat([slabs[rank], slabs[rank + world], slabs[rank + 2 * world]], 0)
    if name.endswith("attention.dense.weight") or name.endswith("dense_4h_to_h.weight"):
        return t.chunk(world, 1)[rank]
    return None


def w_tp2(rank, world):
    from cogview_amd import mpu
    mpu.initialize_model_parallel(world)
    model, logits, loss = _tp_build_and_run()
    return {"logits": logits.detach().float(), "loss": float(loss.detach()),
            "params": {n: p.detach().float() for n, p in model.module.named_parameters()},
            "grads": {n: p.grad.detach().float() / 256.0 for n, p in model.module.named_parameters()}}


def w_mp2_dp2(rank, world):
    """Four ranks: two model-parallel groups (ranks {0,1}, {2,3}) that are data-parallel replicas of each other (groups {0,2},
    {1,3}; mpu/initialize.py:49-75).  Each replica takes one of the two rows; the gradients are exchanged over the data-parallel
    groups during backward."""
    from cogview_amd import mpu, training
    from cogview_amd.fp16 import FP16_Module
    from cogview_amd.model import GPT2Model, PyTorchDistributedDataParallel
    mpu.initialize_model_parallel(2)
    assert mpu.get_model_parallel_rank() == rank % 2 and mpu.get_data_parallel_rank() == rank // 2
    c = _TP
    torch.manual_seed(4321)
    mpu.model_parallel_cuda_manual_seed(4321)
    m = GPT2Model(c["L"], c["V"], c["H"], c["NH"], 0.0, 0.0, 0.0, c["S"] + 1, 0, False)
    model = FP16_Module(m, dtype=torch.float16, keep_half_outputs=True)
    ddp = PyTorchDistributedDataParallel(model, process_group=mpu.get_data_parallel_group(), bucket_layers=1)
    assert ddp.world == 2 and ddp.overlap
    opt = _optimizer(model)
    opt.attach_data_parallel(ddp)
    g = torch.Generator().manual_seed(11)
    tokens = torch.randint(0, c["V"], (c["B"], c["S"]), generator=g)
    labels = torch.randint(0, c["V"], (c["B"], c["S"]), generator=g)
    d = mpu.get_data_parallel_rank()
    pos = torch.arange(c["S"]).unsqueeze(0)
    batch = (tokens[d:d + 1], labels[d:d + 1], torch.ones(1, c["S"]), 0, pos)
    loss, _, _, _ = training.forward_step(batch, ddp, log=False, world_size=2)
    training.backward_step(opt, ddp, loss, 1.0)
    grads = {n: p.grad.detach().float() / opt.loss_scale for n, p in model.module.named_parameters()}
    opt.step()
    assert not opt.overflow
    flat = model.module._cogv_arena.data.detach().float()
    parts = [torch.empty_like(flat) for _ in range(2)]
    dist.all_gather(parts, flat, group=mpu.get_data_parallel_group())
    assert torch.equal(parts[0], parts[1]), "data-parallel replicas of a model shard diverged"
    return {"grads": grads, "mp_rank": mpu.get_model_parallel_rank(), "loss": float(loss.detach())}


def test_model_parallel_2_x_data_parallel_2_train_step(monkeypatch):
    """BASELINE configs[2] x configs[3] in miniature (world 4): every rank's averaged gradients equal the matching slice of the
    one-rank model's gradients on the whole batch; the data-parallel replicas of each shard end bit-identical."""
    out = _run("w_mp2_dp2", world=4)
    from tests import cpu_ops
    cpu_ops.install(monkeypatch.setattr)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
    from cogview_amd import mpu
    if not mpu.model_parallel_is_initialized():
        mpu.initialize_model_parallel(1)
    model, logits, loss = _tp_build_and_run()                 # one rank, both rows, loss = mean over all tokens
    full_g = {n: p.grad.detach().float() / 256.0 for n, p in model.module.named_parameters()}
    assert abs(0.5 * (out[0]["loss"] + out[2]["loss"]) - float(loss.detach())) < 2e-3 * float(loss.detach())
    worst = 0.0
    for r in range(4):
        for n, gfull in full_g.items():
            gs = _tp_slice(n, gfull, out[r]["mp_rank"], 2)
            want = gfull if gs is None else gs
            e = rel(out[r]["grads"][n], want)
            worst = max(worst, e)
            assert e < 2e-2, (n, r, e)
    print(f"mp 2 x dp 2 on the CPU emulation: worst gradient rel-L2 {worst:.2e}")


_TP_CHUNKED = dict(L=2, V=512, H=128, NH=2, S=256, B=4)      # 1024 rows = four 256-row tiles: four row chunks


def w_tp2_row_chunks(rank, world):
    """Row-parallel forward in row chunks (functional._row_parallel_chunked: chunk i's all-reduce under chunk i + 1's GEMM) against
    the whole-tensor form, hidden dropout ON: logits and every gradient must be bit-identical (same masks through dropout_row0,
    same k-loop per element, same element-wise sum across the ranks)."""
    from cogview_amd import functional as F, mpu
    mpu.initialize_model_parallel(world)
    res = []
    for chunks in ("1", "4"):
        os.environ["COGV_MP_ROW_CHUNKS"] = chunks
        model, logits, loss = _tp_build_and_run(_TP_CHUNKED, hidden_dropout=0.1)
        res.append((logits.detach().clone(), float(loss.detach()), {n: p.grad.detach().clone() for n, p in model.module.named_parameters()}))
    os.environ.pop("COGV_MP_ROW_CHUNKS", None)
    assert F.mp_row_chunks(1024) == [(0, 256), (256, 512), (512, 768), (768, 1024)] and F.mp_row_chunks(300) == [(0, 256), (256, 300)]
    assert F.mp_row_chunks(26112) == [(0, 6656), (6656, 13056), (13056, 19712), (19712, 26112)]
    same = torch.equal(res[0][0], res[1][0]) and res[0][1] == res[1][1] and all(torch.equal(res[0][2][n], res[1][2][n]) for n in res[0][2])
    return {"same": same, "dropped": float((res[1][0] == 0).float().mean())}


def test_row_parallel_forward_in_row_chunks_is_bit_identical(monkeypatch):
    out = _run("w_tp2_row_chunks")
    assert out[0]["same"] and out[1]["same"]


def test_two_way_tensor_parallel_forward_backward(monkeypatch):
    """BASELINE configs[2] in miniature on the CPU emulation: each rank draws the full master weights under the same seed and
    keeps its shard, so the two-rank model is the one-rank model: shards equal the slices, the concatenated logits, the loss and
    every gradient (sharded ones against the matching slice) agree to 16-bit round-off."""
    out = _run("w_tp2")
    from tests import cpu_ops
    cpu_ops.install(monkeypatch.setattr)
    monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0)
    from cogview_amd import mpu
    if not mpu.model_parallel_is_initialized():
        mpu.initialize_model_parallel(1)
    model, logits, loss = _tp_build_and_run()
    full_p = {n: p.detach().float() for n, p in model.module.named_parameters()}
    full_g = {n: p.grad.detach().float() / 256.0 for n, p in model.module.named_parameters()}
    got_logits = torch.cat([out[0]["logits"], out[1]["logits"]], dim=-1)
    assert rel(got_logits, logits.detach().float()) < 2e-3
    assert abs(out[0]["loss"] - float(loss.detach())) < 1e-3 * float(loss.detach()) and out[0]["loss"] == out[1]["loss"]
    worst = 0.0
    for r in range(2):
        for n, t in full_p.items():
            sl = _tp_slice(n, t, r, 2)
            want_p = t if sl is None else sl
            assert torch.equal(out[r]["params"][n], want_p), n
            gs = _tp_slice(n, full_g[n], r, 2)
            want_g = full_g[n] if gs is None else gs
            e = rel(out[r]["grads"][n], want_g)
            worst = max(worst, e)
            assert e < 2e-2, (n, r, e)
    print(f"two-way tensor parallel on the CPU emulation: worst gradient rel-L2 {worst:.2e}")
