"""Model-/data-parallel process groups (same topology and API as mpu/initialize.py:30-135).

World of W ranks, model-parallel size p: model-parallel groups are runs of p ADJACENT ranks
[i*p, (i+1)*p) -- on MI355X adjacent ranks share a direct xGMI link, which is what the latency-bound
tensor-parallel all-reduces want -- and data-parallel groups are the strided sets {i, i+p, i+2p, ...}.
The backend is whatever torch.distributed was initialised with: "nccl" (= RCCL over xGMI) on the GPU box,
"gloo" in the CPU tests.
"""
import torch

from .utils import ensure_divisibility

_MODEL_PARALLEL_GROUP = None
_DATA_PARALLEL_GROUP = None


def initialize_model_parallel(model_parallel_size_):
    assert torch.distributed.is_initialized()
    world_size = torch.distributed.get_world_size()
    rank = torch.distributed.get_rank()
    if rank == 0:
        print('> initializing model parallel with size {}'.format(model_parallel_size_))
    mp = min(model_parallel_size_, world_size)
    ensure_divisibility(world_size, mp)
    global _DATA_PARALLEL_GROUP, _MODEL_PARALLEL_GROUP
    assert _DATA_PARALLEL_GROUP is None, 'data parallel group is already initialized'
    for i in range(mp):
        group = torch.distributed.new_group(range(i, world_size, mp))
        if i == rank % mp:
            _DATA_PARALLEL_GROUP = group
    assert _MODEL_PARALLEL_GROUP is None, 'model parallel group is already initialized'
    for i in range(world_size // mp):
        group = torch.distributed.new_group(range(i * mp, (i + 1) * mp))
        if i == rank // mp:
            _MODEL_PARALLEL_GROUP = group


def model_parallel_is_initialized():
    return _MODEL_PARALLEL_GROUP is not None and _DATA_PARALLEL_GROUP is not None


def get_model_parallel_group():
    assert _MODEL_PARALLEL_GROUP is not None, 'model parallel group is not initialized'
    return _MODEL_PARALLEL_GROUP


def get_data_parallel_group():
    assert _DATA_PARALLEL_GROUP is not None, 'data parallel group is not initialized'
    return _DATA_PARALLEL_GROUP


def get_model_parallel_world_size():
    return torch.distributed.get_world_size(group=get_model_parallel_group())


def get_model_parallel_rank():
    return torch.distributed.get_rank(group=get_model_parallel_group())


def get_model_parallel_src_rank():
    """Global rank of model-parallel rank 0 of the caller's group."""
    p = get_model_parallel_world_size()
    return (torch.distributed.get_rank() // p) * p


def get_data_parallel_world_size():
    return torch.distributed.get_world_size(group=get_data_parallel_group())


def get_data_parallel_rank():
    return torch.distributed.get_rank(group=get_data_parallel_group())


def destroy_model_parallel():
    global _MODEL_PARALLEL_GROUP, _DATA_PARALLEL_GROUP
    _MODEL_PARALLEL_GROUP = None
    _DATA_PARALLEL_GROUP = None


def mp_world_size_or_1():
    """Model-parallel size, 1 when no process group exists (single-process use of the layers)."""
    if _MODEL_PARALLEL_GROUP is None:
        return 1
    return get_model_parallel_world_size()


def mp_rank_or_0():
    if _MODEL_PARALLEL_GROUP is None:
        return 0
    return get_model_parallel_rank()
