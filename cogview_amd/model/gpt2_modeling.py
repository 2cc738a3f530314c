"""GPT2Model: constructor / forward signature and state_dict keys of model/gpt2_modeling.py:55-123.

Dataflow on MI355X (model-parallel size 1):
    ids --[1 kernel: gather word row + position row + embedding dropout + abs-max]--> h0
    h0  --[L fused layer Functions, 9 kernels forward / 15 backward each]--> hL
    hL  --[Sandwich-LN]--[MFMA GEMM against the embedding table (tied weights)]--> logits (16-bit)
"""
import torch

from .. import functional as F_
from .. import mpu


def init_method_normal(std=0.02):
    """N(0, std) initialiser used for the word embeddings (model/gpt2_modeling.py:24-32)."""
    return lambda tensor: torch.nn.init.normal_(tensor, mean=0.0, std=std)


def gpt2_get_params_for_weight_decay_optimization(module):
    """Optimizer groups of model/gpt2_modeling.py:35-52: (decay, no-decay).  A parameter is exempt from
    weight decay when it belongs to a LayerNorm or is named `bias`; parameter order inside each group follows
    module traversal order, as in the reference."""
    decay, no_decay = [], []
    for sub in module.modules():
        own = [(n, p) for n, p in sub._parameters.items() if p is not None]
        if isinstance(sub, (mpu.LayerNorm, torch.nn.LayerNorm)):
            no_decay += [p for _, p in own]
            continue
        decay += [p for n, p in own if n != 'bias']
        no_decay += [p for n, p in own if n == 'bias']
    return {'params': decay}, {'params': no_decay, 'weight_decay': 0.0}


class GPT2Model(torch.nn.Module):
    """GPT-2 style decoder over concatenated text + image tokens.  `forward` returns
    (logits, *memories); logits are vocabulary-parallel unless parallel_output=False."""

    # Every parameter gradient of this module is written by a fused backward kernel that asks functional.grad_accumulate
    # (or arena.ensure_zeroed) whether to overwrite or add: the declaration arena.ParamArena.zero_grad(lazy=True) needs.
    _cogv_lazy_zero_grad = True

    def __init__(self, num_layers, vocab_size, hidden_size, num_attention_heads, embedding_dropout_prob,
                 attention_dropout_prob, output_dropout_prob, max_sequence_length, max_memory_length,
                 checkpoint_activations, checkpoint_num_layers=1, parallel_output=True, query_window=128,
                 key_window_times=6, num_pivot=768, kv_cache=False):
        super().__init__()
        self.parallel_output = parallel_output
        self.word_embeddings = mpu.VocabParallelEmbedding(vocab_size, hidden_size,
                                                          init_method=init_method_normal(std=0.02))
        sparse_cfg = dict(query_window=query_window, key_window_times=key_window_times, num_pivot=num_pivot)
        self.transformer = mpu.GPT2ParallelTransformer(
            num_layers, hidden_size, num_attention_heads, max_sequence_length, max_memory_length,
            embedding_dropout_prob, attention_dropout_prob, output_dropout_prob, checkpoint_activations,
            checkpoint_num_layers, kv_cache=kv_cache, **sparse_cfg)

    def forward(self, input_ids, position_ids, attention_mask, txt_indices_bool, img_indices_bool, is_sparse, *mems):
        h0 = self.transformer.embed(input_ids, position_ids, self.word_embeddings)
        hL, *memories = self.transformer(h0, position_ids, attention_mask, txt_indices_bool, img_indices_bool,
                                         is_sparse, *mems, embedded=True)
        logits = F_.tied_logits(hL, self.word_embeddings.weight)      # copy-to-model-parallel folded into backward
        if not self.parallel_output:
            logits = mpu.gather_from_model_parallel_region(logits)
        return (logits, *memories)
