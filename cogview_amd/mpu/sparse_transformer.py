"""Import-path alias: the reference keeps its transformer in mpu/sparse_transformer.py (imported as
`from mpu.sparse_transformer import ...` by model/gpt2_modeling.py and the sparse-attention self test, :752-810)."""
from .transformer import *                                                    # noqa: F401,F403
from .transformer import (GPT2ParallelMLP, GPT2ParallelSelfAttention, GPT2ParallelTransformer,  # noqa: F401
                          GPT2ParallelTransformerLayer, LayerNorm, gelu, scaled_init_method, sparse_attention,
                          sparse_attention_inference, standard_attention, unscaled_init_method)
