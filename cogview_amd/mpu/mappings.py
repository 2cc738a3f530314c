"""Tensor-parallel region mappings over torch.distributed (RCCL on MI355X).

The reference defines four autograd functions (mpu/mappings.py:79-141); each is a (forward, backward) pair
drawn from three collective primitives on the LAST dimension.  They are generated here from one table:

    name      forward        backward       used by
    copy      identity       all-reduce     ColumnParallelLinear input, tied-logits input
    reduce    all-reduce     identity       RowParallelLinear / VocabParallelEmbedding output  (in place)
    scatter   keep my slice  all-gather     RowParallelLinear input when it is not already parallel
    gather    all-gather     keep my slice  ColumnParallelLinear output when gather_output=True
"""
import torch
import torch.distributed as dist

from .initialize import get_model_parallel_group, mp_world_size_or_1


def _identity(t):
    return t


def _all_reduce(t):
    """Sum over the model-parallel group, in place on `t` (the reference reduces in place, :31)."""
    if mp_world_size_or_1() > 1:
        dist.all_reduce(t, group=get_model_parallel_group())
    return t


def _my_slice(t):
    """This rank's contiguous chunk of the last dimension."""
    world = mp_world_size_or_1()
    if world == 1:
        return t
    rank = dist.get_rank(group=get_model_parallel_group())
    width = t.shape[-1]
    assert width % world == 0, '{} is not divisible by {}'.format(width, world)
    step = width // world
    return t[..., rank * step:(rank + 1) * step].contiguous()


def _all_gather_last(t):
    """Concatenate every rank's tensor along the last dimension (rank order)."""
    world = mp_world_size_or_1()
    if world == 1:
        return t
    group = get_model_parallel_group()
    mine = t.contiguous()
    parts = [mine if r == dist.get_rank(group=group) else torch.empty_like(mine) for r in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.cat(parts, dim=-1)


def _region_fn(name, fwd, bwd):
    class _Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return fwd(x)

        @staticmethod
        def backward(ctx, g):
            return bwd(g)
    _Fn.__name__ = _Fn.__qualname__ = name
    return _Fn


_CopyToModelParallelRegion = _region_fn('_CopyToModelParallelRegion', _identity, _all_reduce)
_ReduceFromModelParallelRegion = _region_fn('_ReduceFromModelParallelRegion', _all_reduce, _identity)
_ScatterToModelParallelRegion = _region_fn('_ScatterToModelParallelRegion', _my_slice, _all_gather_last)
_GatherFromModelParallelRegion = _region_fn('_GatherFromModelParallelRegion', _all_gather_last, _my_slice)

copy_to_model_parallel_region = _CopyToModelParallelRegion.apply
reduce_from_model_parallel_region = _ReduceFromModelParallelRegion.apply
scatter_to_model_parallel_region = _ScatterToModelParallelRegion.apply
gather_from_model_parallel_region = _GatherFromModelParallelRegion.apply

# private aliases some callers of the reference reach for
_reduce, _split, _gather = _all_reduce, _my_slice, _all_gather_last
