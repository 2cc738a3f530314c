"""Model-parallel linear / embedding layers with the reference's constructor signatures, parameter names,
shapes and `.model_parallel` attributes (mpu/layers.py:42-326); the arithmetic runs in the HIP library."""
import torch
import torch.nn.init as init
from torch.nn.parameter import Parameter

from .. import functional as F_
from .initialize import mp_rank_or_0, mp_world_size_or_1
from .mappings import (copy_to_model_parallel_region, gather_from_model_parallel_region,
                       reduce_from_model_parallel_region, scatter_to_model_parallel_region)
from .utils import VocabUtility, divide


def _initialize_affine_weight(weight, output_size, input_size, per_partition_size, partition_dim, init_method,
                              stride=1, return_master_weight=False):
    """Same sharding rule as mpu/layers.py:42-74: every rank draws the FULL master weight (so that the result
    does not depend on the partitioning), splits it into per_partition_size/stride slabs along partition_dim
    and keeps slabs rank, rank+p, rank+2p, ... (stride=3 keeps the local [q_r; k_r; v_r] layout of QKV)."""
    world = mp_world_size_or_1()
    if world == 1:
        init_method(weight)
        return weight if return_master_weight else None
    master = torch.empty(output_size, input_size, dtype=weight.dtype, requires_grad=False)
    init_method(master)
    slab = divide(per_partition_size, stride)
    pieces = torch.split(master, slab, dim=partition_dim)[mp_rank_or_0()::world]
    with torch.no_grad():
        weight.copy_(torch.cat(pieces, dim=partition_dim))
    return master if return_master_weight else None


class VocabParallelEmbedding(torch.nn.Module):
    """Embedding sharded along the vocabulary (mpu/layers.py:77-133)."""

    def __init__(self, num_embeddings, embedding_dim, init_method=init.xavier_normal_):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.padding_idx, self.max_norm, self.norm_type = None, None, 2.
        self.scale_grad_by_freq, self.sparse, self._weight = False, False, None
        self.vocab_start_index, self.vocab_end_index = VocabUtility.vocab_range_from_global_vocab_size(
            num_embeddings, mp_rank_or_0(), mp_world_size_or_1())
        self.num_embeddings_per_partition = self.vocab_end_index - self.vocab_start_index
        self.weight = Parameter(torch.empty(self.num_embeddings_per_partition, embedding_dim))
        self.weight.model_parallel = True
        _initialize_affine_weight(self.weight, num_embeddings, embedding_dim, self.num_embeddings_per_partition, 0,
                                  init_method)

    def forward(self, input_):
        return F_.embedding(input_, self.weight, self.vocab_start_index)


class ParallelEmbedding(torch.nn.Module):
    """Embedding sharded along the embedding dimension (mpu/layers.py:136-183)."""

    def __init__(self, num_embeddings, embedding_dim, init_method=init.xavier_normal_,
                 keep_master_weight_for_test=False):
        super().__init__()
        self.num_embeddings, self.embedding_dim = num_embeddings, embedding_dim
        self.padding_idx, self.max_norm, self.norm_type = None, None, 2.
        self.scale_grad_by_freq, self.sparse, self._weight = False, False, None
        self.embedding_dim_per_partition = divide(embedding_dim, mp_world_size_or_1())
        self.weight = Parameter(torch.empty(num_embeddings, self.embedding_dim_per_partition))
        self.weight.model_parallel = True
        _initialize_affine_weight(self.weight, num_embeddings, embedding_dim, self.embedding_dim_per_partition, 1,
                                  init_method, stride=1, return_master_weight=False)

    def forward(self, input_):
        out = F_.embedding(copy_to_model_parallel_region(input_), self.weight, 0)
        return gather_from_model_parallel_region(out)


class ColumnParallelLinear(torch.nn.Module):
    """Y = X A + b with A split along its columns (mpu/layers.py:186-249).  weight: [out/p, in]."""

    def __init__(self, input_size, output_size, bias=True, gather_output=True, init_method=init.xavier_normal_,
                 stride=1, keep_master_weight_for_test=False):
        super().__init__()
        self.input_size, self.output_size, self.gather_output = input_size, output_size, gather_output
        self.output_size_per_partition = divide(output_size, mp_world_size_or_1())
        self.weight = Parameter(torch.empty(self.output_size_per_partition, input_size))
        self.weight.model_parallel = True
        if bias:
            self.bias = Parameter(torch.zeros(self.output_size_per_partition))
            self.bias.model_parallel = True
        else:
            self.register_parameter('bias', None)
        self.master_weight = _initialize_affine_weight(
            self.weight, output_size, input_size, self.output_size_per_partition, 0, init_method, stride=stride,
            return_master_weight=keep_master_weight_for_test)

    def forward(self, input_):
        out = F_.linear(copy_to_model_parallel_region(input_), self.weight, self.bias)
        return gather_from_model_parallel_region(out) if self.gather_output else out


class RowParallelLinear(torch.nn.Module):
    """Y = X A + b with A split along its rows and X along its last dim (mpu/layers.py:252-326).
    weight: [out, in/p]; bias is NOT sharded and is added after the all-reduce."""

    def __init__(self, input_size, output_size, bias=True, input_is_parallel=False, init_method=init.xavier_normal_,
                 stride=1, keep_master_weight_for_test=False):
        super().__init__()
        self.input_size, self.output_size, self.input_is_parallel = input_size, output_size, input_is_parallel
        self.input_size_per_partition = divide(input_size, mp_world_size_or_1())
        self.weight = Parameter(torch.empty(output_size, self.input_size_per_partition))
        self.weight.model_parallel = True
        if bias:
            self.bias = Parameter(torch.zeros(output_size))
        else:
            self.register_parameter('bias', None)
        self.master_weight = _initialize_affine_weight(
            self.weight, output_size, input_size, self.input_size_per_partition, 1, init_method, stride=stride,
            return_master_weight=keep_master_weight_for_test)

    def forward(self, input_):
        x = input_ if self.input_is_parallel else scatter_to_model_parallel_region(input_)
        if mp_world_size_or_1() == 1:
            return F_.linear(x, self.weight, self.bias)          # bias fused into the GEMM epilogue
        # bias enters once (on model-parallel rank 0) before the all-reduce: same sum as adding it afterwards
        out = F_.linear(x, self.weight, self.bias if mp_rank_or_0() == 0 else None)
        out = reduce_from_model_parallel_region(out)
        if self.bias is not None and mp_rank_or_0() != 0 and self.bias.requires_grad:
            out = _BiasGradOnly.apply(out, self.bias)
        return out


class _BiasGradOnly(torch.autograd.Function):
    """Identity that still produces the (replicated) bias gradient on ranks whose GEMM did not add the bias."""

    @staticmethod
    def forward(ctx, x, bias):
        ctx.bias = bias
        return x

    @staticmethod
    def backward(ctx, dy):
        from .. import ops
        d2 = dy.reshape(-1, dy.shape[-1])
        ops.colsum(d2 if d2.is_contiguous() else d2.contiguous(), out=F_.grad_buffer(ctx.bias),
                   accumulate=F_.grad_accumulate(ctx.bias))
        return dy, None
