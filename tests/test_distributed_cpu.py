"""CPU, multi-process (gloo): the N>1 host logic -- process-group topology, the four tensor-parallel mappings,
broadcast_data, data-parallel gradient averaging (both allreduce_params orders), loss-scaler overflow sync and
the model-parallel weight sharding rule.  No compute kernels are involved (there is no GPU here)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _entry(rank, world, port, fn_name, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", world_size=world, rank=rank)
    try:
        globals()[fn_name](rank, world)
        ret[rank] = "ok"
    except Exception as e:                      # surface the failure in the parent
        import traceback
        ret[rank] = traceback.format_exc()
    finally:
        dist.destroy_process_group()


def _run(fn_name, world):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_entry, args=(r, world, _free_port_shared[0], fn_name, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
        for r in range(world):
            assert ret.get(r) == "ok", f"rank {r}: {ret.get(r)}"


_free_port_shared = [0]


@pytest.fixture(autouse=True)
def _port():
    _free_port_shared[0] = _free_port()


# ------------------------------------------------------------------------------------------------ workers
def w_topology(rank, world):
    from cogview_amd import mpu
    mpu.initialize_model_parallel(2)
    assert mpu.model_parallel_is_initialized()
    assert mpu.get_model_parallel_world_size() == 2 and mpu.get_data_parallel_world_size() == world // 2
    assert mpu.get_model_parallel_rank() == rank % 2 and mpu.get_data_parallel_rank() == rank // 2
    assert mpu.get_model_parallel_src_rank() == (rank // 2) * 2            # adjacent ranks form a model-parallel group
    t = torch.tensor([float(rank)])
    dist.all_reduce(t, group=mpu.get_model_parallel_group())
    assert t.item() == (rank // 2) * 4 + 1                                  # r + (r^1)
    t = torch.tensor([float(rank)])
    dist.all_reduce(t, group=mpu.get_data_parallel_group())
    assert t.item() == sum(r for r in range(world) if r % 2 == rank % 2)
    mpu.destroy_model_parallel()
    assert not mpu.model_parallel_is_initialized()


def w_mappings(rank, world):
    from cogview_amd import mpu
    mpu.initialize_model_parallel(world)
    x = (torch.arange(6, dtype=torch.float32).view(1, 6) + 10 * rank).requires_grad_(True)
    # copy: identity forward, all-reduce backward
    y = mpu.copy_to_model_parallel_region(x)
    y.backward(torch.ones_like(y) * (rank + 1))
    assert torch.equal(y.detach(), x.detach()) and torch.equal(x.grad, torch.full_like(x, sum(range(1, world + 1))))
    # reduce: all-reduce forward (in place on its input), identity backward
    x2 = x.detach().clone().requires_grad_(True)
    y2 = mpu.reduce_from_model_parallel_region(x2 * 1.0)
    expect = sum(torch.arange(6, dtype=torch.float32).view(1, 6) + 10 * r for r in range(world))
    assert torch.equal(y2.detach(), expect)
    y2.backward(torch.ones_like(y2))
    assert torch.equal(x2.grad, torch.ones_like(x2))
    # scatter / gather are inverses along the last dim
    full = torch.arange(12, dtype=torch.float32).view(2, 6).requires_grad_(True)
    part = mpu.scatter_to_model_parallel_region(full)
    w = 6 // world
    assert torch.equal(part.detach(), full.detach()[:, rank * w:(rank + 1) * w])
    back = mpu.gather_from_model_parallel_region(part)
    assert torch.equal(back.detach(), full.detach())
    back.sum().backward()
    assert torch.equal(full.grad, torch.ones_like(full))


def w_broadcast_data(rank, world):
    from cogview_amd import mpu
    mpu.initialize_model_parallel(2)
    keys = ['text', 'loss_mask']
    if mpu.get_model_parallel_rank() == 0:
        g = torch.Generator().manual_seed(100 + rank)
        data = {'text': torch.randint(0, 58219, (3, 1089), generator=g), 'loss_mask': torch.ones(3, 1089, dtype=torch.int64)}
    else:
        data = None
    out = mpu.broadcast_data(keys, data, torch.int64)
    g = torch.Generator().manual_seed(100 + mpu.get_model_parallel_src_rank())
    assert torch.equal(out['text'].cpu(), torch.randint(0, 58219, (3, 1089), generator=g))
    assert out['loss_mask'].shape == (3, 1089)


def w_ddp(rank, world):
    from cogview_amd import mpu
    from cogview_amd.model import DistributedDataParallel
    mpu.initialize_model_parallel(1)
    torch.manual_seed(rank)                                   # replicas start different ...
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    ddp = DistributedDataParallel(net)
    ref = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    torch.manual_seed(0)
    ref0 = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    for p, q in zip(net.parameters(), ref0.parameters()):     # ... and end up with rank 0's parameters
        assert torch.equal(p.data, q.data)
    for order in (False, True):
        for p in net.parameters():
            p.grad = torch.full_like(p, float(rank + 1))
        ddp.needs_reduction = True
        ddp.allreduce_params(reduce_after=order)
        mean = sum(range(1, world + 1)) / world
        for p in net.parameters():
            assert torch.allclose(p.grad, torch.full_like(p, mean))
        ddp.allreduce_params()                                # needs_reduction False -> no-op
        for p in net.parameters():
            assert torch.allclose(p.grad, torch.full_like(p, mean))
    assert set(ddp.state_dict().keys()) == set(net.state_dict().keys())      # unwrapped keys (model/distributed.py:27-29)


def w_overflow_sync(rank, world):
    from cogview_amd import mpu
    from cogview_amd.fp16 import DynamicLossScaler
    mpu.initialize_model_parallel(world)
    sc = DynamicLossScaler(init_scale=2 ** 8)
    assert sc.sync_overflow(rank == 1) is True                # one shard overflowed -> every rank skips
    assert sc.sync_overflow(False) is False


def w_sharding_rule(rank, world):
    """_initialize_affine_weight: every rank draws the same master and keeps its strided slabs
    (stride=3 for QKV keeps [q_r; k_r; v_r]); concatenating ranks' shards reproduces the master."""
    from cogview_amd import mpu
    from cogview_amd.mpu.layers import _initialize_affine_weight
    mpu.initialize_model_parallel(world)
    torch.manual_seed(7)
    w = torch.empty(12 // world, 4)
    master = _initialize_affine_weight(w, 12, 4, 12 // world, 0, torch.nn.init.normal_, stride=3, return_master_weight=True)
    slabs = torch.split(master, 12 // world // 3, dim=0)
    assert torch.equal(w, torch.cat(slabs[rank::world], dim=0))
    parts = [torch.empty_like(w) for _ in range(world)]
    dist.all_gather(parts, w)
    q = torch.cat([p_[0:len(p_) // 3] for p_ in parts])
    assert torch.equal(q, master[0:4])                          # all ranks' q rows = master's q block
    wr = torch.empty(5, 8 // world)
    m2 = _initialize_affine_weight(wr, 5, 8, 8 // world, 1, torch.nn.init.normal_, return_master_weight=True)
    assert torch.equal(wr, m2[:, rank * (8 // world):(rank + 1) * (8 // world)])
    emb = mpu.VocabParallelEmbedding(16, 4)
    assert (emb.vocab_start_index, emb.vocab_end_index) == (rank * 16 // world, (rank + 1) * 16 // world)
    assert emb.weight.model_parallel and emb.weight.shape == (16 // world, 4)
    col = mpu.ColumnParallelLinear(4, 12, gather_output=False)
    row = mpu.RowParallelLinear(12, 4, input_is_parallel=True)
    assert col.weight.shape == (12 // world, 4) and row.weight.shape == (4, 12 // world) and row.bias.shape == (4,)
    assert col.bias.model_parallel and not hasattr(row.bias, "model_parallel")


def w_shard_plan(rank, world):
    """ShardPlan (reduce-scatter / all-gather exchange): the ranks' owned slices tile the arena exactly once, the
    restricted chunk table lists exactly the owned elements, and reduce_region leaves the MEAN gradient in the owner's
    slice (gloo, CPU tensors: the collectives' host logic; the stream / kernel side is covered on the GPU)."""
    from types import SimpleNamespace
    from cogview_amd import mpu
    from cogview_amd.arena import ParamArena
    from cogview_amd.model.distributed import ShardPlan
    mpu.initialize_model_parallel(1)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(n)) for n in (1000, 70000, 300, 4096, 131072 + 5, 640)]
    arena = ParamArena(params, torch.float32, torch.device("cpu"))
    cut1, cut2 = arena.offsets[2], arena.offsets[4]
    ddp = SimpleNamespace(world=world, data_parallel_group=mpu.get_data_parallel_group(), arena=arena,
                          _buckets=[(1, (cut2, arena.total)), (0, (cut1, cut2))])
    ddp._allreduce_mean = lambda g: (g.div_(world), dist.all_reduce(g, group=ddp.data_parallel_group))
    plan = ShardPlan(ddp)
    assert plan.regions == [(0, cut1), (cut1, cut2), (cut2, arena.total)]
    cover = torch.zeros(arena.total, dtype=torch.int32)
    for r in range(world):
        for a, b in plan.owned(r):
            assert a % 8 == 0 and b > a
            cover[a:b] += 1
    assert bool((cover == 1).all()), "owned slices must tile the arena exactly once"
    own = plan.owned()
    st, ln, grp, nrm = arena.chunk_table(lambda p: 0, lambda p: True, owned=own)
    listed = torch.zeros(arena.total, dtype=torch.int32)
    for a, n in zip(st.tolist(), ln.tolist()):
        listed[a:a + n] += 1
    want = torch.zeros(arena.total, dtype=torch.int32)
    for (p_, off) in zip(arena.params, arena.offsets):
        for a, b in own:
            lo, hi = max(a, off), min(b, off + p_.numel())
            if hi > lo:
                want[lo:hi] = 1
    assert torch.equal(listed, want), "restricted chunk table != owned parameter elements"
    arena.grad.copy_(torch.arange(arena.total, dtype=torch.float32) * (rank + 1))
    for s_, e_ in plan.regions:
        plan.reduce_region(s_, e_)
    mean = torch.arange(arena.total, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    for a, b in own:
        assert torch.allclose(arena.grad[a:b], mean[a:b], rtol=1e-6), "owned slice does not hold the mean gradient"


def w_activation_partitioning(rank, world):
    """mpu.partition_activations_in_checkpoint(True) (mpu/random.py:298-360): a checkpointed function's tensor inputs
    (all but the last, the mask) are kept as this model-parallel rank's 1/p slice and all-gathered for the recompute;
    outputs and gradients equal the un-checkpointed run."""
    from cogview_amd import mpu
    from cogview_amd.mpu import random as R
    mpu.initialize_model_parallel(world)
    torch.manual_seed(5)                                  # the same activation on every model-parallel rank
    x = torch.randn(4, 6, 16, requires_grad=True)
    mask = torch.tril(torch.ones(6, 6))
    w = torch.randn(16, 16, requires_grad=True)

    def fn(h, m):
        return torch.tanh(h @ w) * m.sum(-1).view(1, 6, 1)
    ref = fn(x, mask)
    ref.sum().backward()
    gx, gw = x.grad.clone(), w.grad.clone()
    x.grad = w.grad = None
    assert R.partition_activation(x)[1] is None           # off by default: kept whole
    mpu.partition_activations_in_checkpoint(True)
    try:
        piece, shape = R.partition_activation(x)
        assert piece.numel() == x.numel() // world and shape == tuple(x.shape)
        assert torch.equal(R.gather_activation(piece, shape), x.detach())
        out = mpu.checkpoint(fn, x, mask)
        saved = out.grad_fn.saved_tensors
        assert saved[0].numel() == x.numel() // world and saved[1].numel() == mask.numel()      # the mask stays whole
        out.sum().backward()
    finally:
        mpu.partition_activations_in_checkpoint(False)
    assert torch.equal(out, ref) and torch.allclose(x.grad, gx) and torch.allclose(w.grad, gw)


def w_checkpoint_mp2_dp2(rank, world):
    """utils.save/load_checkpoint on a 2 x 2 grid: each model-parallel rank's file is written once (by its data-parallel
    rank 0), the tracker by global rank 0, and every rank reloads its own shard."""
    import tempfile
    from types import SimpleNamespace
    from cogview_amd import mpu, utils
    from cogview_amd.model import GPT2Model
    mpu.initialize_model_parallel(2)
    box = [tempfile.mkdtemp() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    args = SimpleNamespace(save=box[0], load=box[0], deepspeed=False, no_save_optim=True, no_save_rng=True,
                           no_load_optim=True, no_load_rng=True, finetune=False)
    torch.manual_seed(3)
    m = GPT2Model(2, 64, 128, 2, 0.0, 0.0, 0.0, 16, 0, False)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.add_(float(mpu.get_model_parallel_rank()))          # make the two shards' files distinguishable
    utils.save_checkpoint(40, m, None, None, args)
    files = sorted(os.listdir(os.path.join(box[0], "40")))
    assert files == ["mp_rank_00_model_states.pt", "mp_rank_01_model_states.pt"], files
    torch.manual_seed(9)
    m2 = GPT2Model(2, 64, 128, 2, 0.0, 0.0, 0.0, 16, 0, False)
    assert utils.load_checkpoint(m2, None, None, args) == 40
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    assert m2.transformer.layers[0].attention.query_key_value.weight.shape == (3 * 128 // 2, 128)


def w_trainer_data_path(rank, world):
    """pretrain_gpt2: flags, the learning-rate scheduler wiring and the data path -- CompactBinaryDataset ->
    RandomMappingDataset -> global batches sliced per data-parallel rank, resumable at an iteration."""
    import tempfile
    import numpy as np
    from cogview_amd import mpu
    from cogview_amd import pretrain_gpt2 as P
    from cogview_amd.data_utils import write_compact_binary
    mpu.initialize_model_parallel(1)
    box = [tempfile.mkdtemp() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    path = os.path.join(box[0], "train.bin")
    if rank == 0:
        rs = np.random.RandomState(0)
        write_compact_binary(path, [rs.randint(8192, 58192, rs.randint(2, 30)).tolist() for _ in range(40)],
                             rs.randint(0, 8192, (40, 1024)))
    dist.barrier()
    args = P.get_args(["--num-layers", "2", "--hidden-size", "128", "--num-attention-heads", "2", "--fp16", "--batch-size", "3",
                       "--train-data", path, "--num-workers", "0", "--lr", "2e-4", "--lr-decay-style", "cosine",
                       "--train-iters", "100", "--warmup", "0.1"])
    assert args.dynamic_loss_scale and args.deepspeed is False and args.max_position_embeddings == 1089

    class Opt:
        param_groups = [{"lr": 0.0}]
    sch = P.get_learning_rate_scheduler(Opt(), args)
    assert sch.warmup_iter == 10 and sch.end_iter == 100 and sch.decay_style == "cosine"
    it = P.make_data_iterator(args)
    first = [next(it) for _ in range(3)]
    assert first[0]["text"].shape == (3, 1089) and first[0]["loss_mask"].shape == (3, 1089)
    mine = torch.stack([b_["text"] for b_ in first])
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    assert not torch.equal(both[0], both[1])                   # the two ranks hold different slices of each global batch
    args.iteration = 2                                          # resume: the third batch again
    again = next(P.make_data_iterator(args))
    assert torch.equal(again["text"], first[2]["text"])
    # the global batch is what a single rank would draw: 6 consecutive virtual indices of the index-seeded mapping
    from cogview_amd.data_utils import RandomMappingDataset, get_dataset_by_type
    ds = RandomMappingDataset(get_dataset_by_type(args.dataset_type, path, args))
    want = torch.stack([torch.from_numpy(np.asarray(ds[i]["text"])) for i in range(rank * 3, rank * 3 + 3)])
    assert torch.equal(first[0]["text"], want)


# ------------------------------------------------------------------------------------------------ tests
def test_trainer_data_path_world2():
    _run("w_trainer_data_path", 2)


def test_checkpoint_files_world4_mp2():
    _run("w_checkpoint_mp2_dp2", 4)


def test_activation_partitioning_world2():
    _run("w_activation_partitioning", 2)


def test_shard_plan_world2():
    _run("w_shard_plan", 2)


def test_shard_plan_world4():
    _run("w_shard_plan", 4)


def test_topology_world4_mp2():
    _run("w_topology", 4)


def test_mappings_world2():
    _run("w_mappings", 2)


def test_broadcast_data_world4_mp2():
    _run("w_broadcast_data", 4)


def test_data_parallel_allreduce_world2():
    _run("w_ddp", 2)


def test_overflow_flag_sync_world2():
    _run("w_overflow_sync", 2)


def test_weight_sharding_rule_world2():
    _run("w_sharding_rule", 2)
