"""`mpu` -- the model-parallel toolkit surface of the reference (30 public names, mpu/__init__.py:18-51),
implemented on the MI355X HIP library.  `import cogview_amd.mpu as mpu` is a drop-in for `import mpu`."""
from . import data, grads, initialize, layers, mappings, random, transformer, utils  # noqa: F401
from .cross_entropy import vocab_parallel_cross_entropy
from .data import broadcast_data
from .grads import clip_grad_norm
from .initialize import (destroy_model_parallel, get_data_parallel_group, get_data_parallel_rank,
                         get_data_parallel_world_size, get_model_parallel_group, get_model_parallel_rank,
                         get_model_parallel_src_rank, get_model_parallel_world_size, initialize_model_parallel,
                         model_parallel_is_initialized)
from .layers import ColumnParallelLinear, ParallelEmbedding, RowParallelLinear, VocabParallelEmbedding
from .mappings import (copy_to_model_parallel_region, gather_from_model_parallel_region,
                       reduce_from_model_parallel_region, scatter_to_model_parallel_region)
from .random import (checkpoint, get_cuda_rng_tracker, model_parallel_cuda_manual_seed,
                     partition_activations_in_checkpoint)
from .transformer import GPT2ParallelTransformer, LayerNorm

__all__ = [
    'vocab_parallel_cross_entropy', 'broadcast_data', 'clip_grad_norm',
    'destroy_model_parallel', 'get_data_parallel_group', 'get_data_parallel_rank', 'get_data_parallel_world_size',
    'get_model_parallel_group', 'get_model_parallel_rank', 'get_model_parallel_src_rank',
    'get_model_parallel_world_size', 'initialize_model_parallel', 'model_parallel_is_initialized',
    'ColumnParallelLinear', 'ParallelEmbedding', 'RowParallelLinear', 'VocabParallelEmbedding',
    'copy_to_model_parallel_region', 'gather_from_model_parallel_region', 'reduce_from_model_parallel_region',
    'scatter_to_model_parallel_region', 'checkpoint', 'partition_activations_in_checkpoint',
    'get_cuda_rng_tracker', 'model_parallel_cuda_manual_seed', 'GPT2ParallelTransformer', 'LayerNorm',
]
