"""GPT-2 transformer with Sandwich-LN -- module surface of the reference's mpu/sparse_transformer.py
(class names, constructor arguments, parameter names and shapes, forward signatures), HIP arithmetic.

Forward paths:
  * training / evaluation on whole sequences (no memories): each layer is ONE fused autograd Function
    (cogview_amd.functional.transformer_layer);
  * incremental decoding with `mems` (generation/sampling.py:64-186 keeps LAYER INPUTS as memories): the same
    kernels composed op by op (no autograd needed there).
  * sparse attention (mpu/sparse_transformer.py:675-750): is_sparse = 1 (training:
    pivots + blocked window, forward and backward in "slot space", see csrc/attention.hip) and is_sparse = 2
    (generation: gathered keys), composed op by op around the sparse attention kernels.
"""
import math

import torch

from .. import functional as F_
from .. import ops
from .initialize import mp_world_size_or_1
from .layers import ColumnParallelLinear, RowParallelLinear
from .utils import divide, split_tensor_along_last_dim


class LayerNorm(torch.nn.Module):
    """Sandwich-LN primitive: LayerNorm(x / (max|x| / 8)) (mpu/sparse_transformer.py:40-44; the reference
    subclasses apex FusedLayerNorm -- parameters `weight`, `bias`, attribute `eps`)."""

    def __init__(self, normalized_shape, eps=1e-5, elementwise_affine=True):
        super().__init__()
        if isinstance(normalized_shape, int):
            normalized_shape = (normalized_shape,)
        assert len(normalized_shape) == 1 and elementwise_affine
        self.normalized_shape = tuple(normalized_shape)
        self.eps = eps
        self.elementwise_affine = True
        self.weight = torch.nn.Parameter(torch.ones(*normalized_shape))
        self.bias = torch.nn.Parameter(torch.zeros(*normalized_shape))

    def forward(self, x, residual=None):
        """residual (extension): returns residual + LN(x) -- in fp32 when `residual` is the fp32 residual stream."""
        return F_.sandwich_layer_norm(x, self.weight, self.bias, self.eps, residual=residual)


def gelu(x):
    """OpenAI tanh GeLU (mpu/sparse_transformer.py:172-179)."""
    return F_.gelu(x)


standard_attention = F_.standard_attention
sparse_attention = F_.sparse_attention                          # mpu/sparse_transformer.py:675-725
sparse_attention_inference = F_.sparse_attention_inference      # mpu/sparse_transformer.py:727-750 (forward only)


class _Dropout(torch.nn.Module):
    """torch.nn.Dropout stand-in (same `.p` / `.training` protocol) on the counter-based HIP dropout."""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, x):
        return F_.dropout(x, self.p, self.training)


class KVCacheSlot:
    """One layer's key/value cache travelling through the `*mems` arguments in K/V-cache mode (SURVEY section 8f item 2:
    "replace layer-input mems with a real K/V cache behind the same *mems signature").

    The memory tensor the caller sees is a [b, s_mem, 2 * hp] view (keys | values of the positions decoded so far)
    of a buffer with spare capacity; the view object carries the buffer as an attribute, so when the caller hands the
    same object back (generation/sampling.py does) the new keys/values are appended in place: O(new tokens) per step
    instead of the reference's QKV projection over the whole memory (mpu/sparse_transformer.py:135-140).  A memory that
    lost the attribute (expanded to more beams, re-indexed by shrink_beams) is copied into a fresh buffer once."""
    GROW = 256

    def __init__(self, mem, max_len=0):
        self.mem, self.max_len, self.out = mem, int(max_len), None

    def append(self, k_new, v_new):
        """k_new, v_new [b, sq, hp] -> (cache view [b, s_mem + sq, 2 hp], keys view, values view)."""
        b, sq, hp = k_new.shape
        mem = self.mem
        s_mem = 0 if mem is None else mem.size(1)
        buf = getattr(mem, "_cogv_kv_buf", None) if mem is not None else None
        need = s_mem + sq
        if buf is None or buf.size(0) != b or buf.size(1) < need or mem.data_ptr() != buf.data_ptr():
            cap = ((need + self.GROW - 1) // self.GROW + 1) * self.GROW
            buf = torch.empty((b, cap, 2 * hp), dtype=k_new.dtype, device=k_new.device)
            if s_mem:
                buf[:, :s_mem].copy_(mem)
        buf[:, s_mem:need, :hp].copy_(k_new)
        buf[:, s_mem:need, hp:].copy_(v_new)
        view = buf[:, :need]
        view._cogv_kv_buf = buf
        self.out = view
        if self.max_len > 0 and need > self.max_len:
            # memory window (max_memory_length, mpu/sparse_transformer.py:615-626): this step still attends everything it
            # was given; the memory handed back keeps the last max_len positions, at the front of a fresh buffer
            nb = torch.empty_like(buf)
            nb[:, :self.max_len].copy_(buf[:, need - self.max_len:need])
            self.out = nb[:, :self.max_len]
            self.out._cogv_kv_buf = nb
        return view, view[:, :, :hp], view[:, :, hp:]


class StaticKVSlot(KVCacheSlot):
    """A layer's key/value cache with FIXED capacity and address for captured decode steps (generation/decoder.py): the
    new keys / values are written at the device-side position `pos_index`, attention runs over all `cap` slots through
    the gathered form, whose index table flags the slots not written yet -- no launch parameter depends on the length."""

    def __init__(self, cache, pos_index, table):
        super().__init__(None, 0)
        self.cache, self.pos_index, self.table = cache, pos_index, table

    def append(self, k_new, v_new):
        hp = k_new.shape[-1]
        self.cache.index_copy_(1, self.pos_index, torch.cat((k_new, v_new), -1))
        self.out = self.cache
        return self.cache, self.cache[:, :, :hp], self.cache[:, :, hp:]


class GPT2ParallelSelfAttention(torch.nn.Module):
    """mpu/sparse_transformer.py:46-169."""

    def __init__(self, hidden_size, num_attention_heads, attention_dropout_prob, output_dropout_prob, init_method,
                 output_layer_init_method=None, query_window=128, key_window_times=6):
        super().__init__()
        if output_layer_init_method is None:
            output_layer_init_method = init_method
        world_size = mp_world_size_or_1()
        self.hidden_size_per_partition = divide(hidden_size, world_size)
        self.hidden_size_per_attention_head = divide(hidden_size, num_attention_heads)
        self.num_attention_heads_per_partition = divide(num_attention_heads, world_size)
        if self.hidden_size_per_attention_head != 64:
            raise NotImplementedError("the HIP attention kernels are built for head dim 64 (every CogView config)")
        self.query_window, self.key_window_times = query_window, key_window_times
        self.query_key_value = ColumnParallelLinear(hidden_size, 3 * hidden_size, stride=3, gather_output=False,
                                                    init_method=init_method)
        self.attention_dropout = _Dropout(attention_dropout_prob)
        self.dense = RowParallelLinear(hidden_size, hidden_size, input_is_parallel=True,
                                       init_method=output_layer_init_method)
        self.output_dropout = _Dropout(output_dropout_prob)

    def _transpose_for_scores(self, tensor):
        shape = tensor.size()[:-1] + (self.num_attention_heads_per_partition, self.hidden_size_per_attention_head)
        return tensor.view(*shape).permute(0, 2, 1, 3)

    def forward(self, hidden_states, ltor_mask, pivot_idx=None, is_sparse=0, mem=None):
        query_length = hidden_states.size(1)
        if isinstance(mem, KVCacheSlot):
            # K/V-cache mode: project only the new positions, append their keys / values to the layer's cache and
            # attend over the cache (strided views: the attention kernels take any row stride)
            assert int(is_sparse) == 0
            mixed = self.query_key_value(hidden_states)
            q, k, v = split_tensor_along_last_dim(mixed, 3)
            _, k, v = mem.append(k, v)
        else:
            src = hidden_states if mem is None else torch.cat((mem, hidden_states), 1)
            mixed = self.query_key_value(src)
            q, k, v = split_tensor_along_last_dim(mixed, 3)
            if mem is not None:
                q = q[:, -query_length:]
        if isinstance(mem, StaticKVSlot):   # captured decode step: every slot of the fixed-capacity cache, unwritten ones flagged
            o, _ = ops.attention_fwd(F_._as_bshd(self._transpose_for_scores(q)), F_._as_bshd(self._transpose_for_scores(k)),
                                     F_._as_bshd(self._transpose_for_scores(v)), kv_index=mem.table)
            ctx = o.permute(0, 2, 1, 3)
        elif int(is_sparse) == 1:        # mpu/sparse_transformer.py:147-148: ltor_mask carries the pivot attention mask
            assert mem is None
            ctx = sparse_attention(self._transpose_for_scores(q), self._transpose_for_scores(k),
                                   self._transpose_for_scores(v), pivot_idx, ltor_mask, self.query_window,
                                   self.key_window_times, self.attention_dropout)
        elif int(is_sparse) == 2:        # mpu/sparse_transformer.py:149-150: pivot_idx carries pivots + trailing window
            ctx = sparse_attention_inference(self._transpose_for_scores(q), self._transpose_for_scores(k),
                                             self._transpose_for_scores(v), pivot_idx)
        else:
            ctx = standard_attention(self._transpose_for_scores(q), self._transpose_for_scores(k),
                                     self._transpose_for_scores(v), ltor_mask, self.attention_dropout)
        ctx = ctx.permute(0, 2, 1, 3).contiguous()
        ctx = ctx.view(*ctx.size()[:-2], self.hidden_size_per_partition)
        return self.output_dropout(self.dense(ctx))


class GPT2ParallelMLP(torch.nn.Module):
    """mpu/sparse_transformer.py:189-234."""

    def __init__(self, hidden_size, output_dropout_prob, init_method, output_layer_init_method=None):
        super().__init__()
        if output_layer_init_method is None:
            output_layer_init_method = init_method
        self.dense_h_to_4h = ColumnParallelLinear(hidden_size, 4 * hidden_size, gather_output=False,
                                                  init_method=init_method)
        self.dense_4h_to_h = RowParallelLinear(4 * hidden_size, hidden_size, input_is_parallel=True,
                                               init_method=output_layer_init_method)
        self.dropout = _Dropout(output_dropout_prob)

    def forward(self, hidden_states):
        return self.dropout(self.dense_4h_to_h(gelu(self.dense_h_to_4h(hidden_states))))


class GPT2ParallelTransformerLayer(torch.nn.Module):
    """mpu/sparse_transformer.py:237-342 (Sandwich-LN: four LayerNorms per layer)."""

    def __init__(self, hidden_size, num_attention_heads, attention_dropout_prob, output_dropout_prob,
                 layernorm_epsilon, init_method, output_layer_init_method=None, query_window=128, key_window_times=6,
                 scale_normalization=True):
        super().__init__()
        if output_layer_init_method is None:
            output_layer_init_method = init_method
        self.input_layernorm = LayerNorm(hidden_size, eps=layernorm_epsilon)
        self.attention = GPT2ParallelSelfAttention(hidden_size, num_attention_heads, attention_dropout_prob,
                                                   output_dropout_prob, init_method,
                                                   output_layer_init_method=output_layer_init_method,
                                                   query_window=query_window, key_window_times=key_window_times)
        self.post_attention_layernorm = LayerNorm(hidden_size, eps=layernorm_epsilon)
        self.scale_normalization = scale_normalization
        if scale_normalization:
            self.third_layernorm = LayerNorm(hidden_size, eps=layernorm_epsilon)
            self.fourth_layernorm = LayerNorm(hidden_size, eps=layernorm_epsilon)
        self.mlp = GPT2ParallelMLP(hidden_size, output_dropout_prob, init_method,
                                   output_layer_init_method=output_layer_init_method)

    def forward(self, hidden_states, ltor_mask, pivot_idx=None, is_sparse=0, mem=None, recompute=False,
                on_backward_done=None):
        is_sparse = int(is_sparse)
        # The hidden state that runs through the layer loop is the fp32 RESIDUAL STREAM (csrc/common.cuh); a 16-bit
        # input (a layer used on its own, as the reference's modules allow) is widened here and the result narrowed back
        in_dtype = hidden_states.dtype
        if in_dtype != torch.float32:
            out = self.forward(hidden_states.float(), ltor_mask, pivot_idx, is_sparse, mem, recompute, on_backward_done)
            return out.to(in_dtype)
        if mem is None and self.scale_normalization and is_sparse == 0:
            s = hidden_states.size(1)
            sep = F_.mask_to_sep(ltor_mask, s, s)
            if sep is not None:          # (an arbitrary mask tensor takes the op-by-op composition below)
                return F_.transformer_layer(self, hidden_states, getattr(hidden_states, "_cogv_absmax", None), sep,
                                            self.training, recompute, on_backward_done)
        if isinstance(mem, KVCacheSlot) and self.scale_normalization and is_sparse == 0 and not torch.is_grad_enabled():
            sep = ltor_mask if isinstance(ltor_mask, int) else F_.mask_to_sep(ltor_mask, hidden_states.size(1),
                                                                              hidden_states.size(1))
            if sep is None:
                raise NotImplementedError("key/value-cache decoding takes the left-to-right mask (int `sep` or its tensor form)")
            return F_.transformer_layer_kv(self, hidden_states, getattr(hidden_states, "_cogv_absmax", None), sep, mem)
        # op-by-op composition (memories / no Sandwich-LN), exactly the reference's dataflow
        a = self.input_layernorm(hidden_states)
        if mem is not None and not isinstance(mem, KVCacheSlot):
            mem = self.input_layernorm(mem.float() if mem.dtype != torch.float32 else mem)
        att = self.attention(a, ltor_mask, pivot_idx, is_sparse, mem)
        if self.scale_normalization:
            y = self.third_layernorm(att, residual=hidden_states)        # x + LN3(att), summed in fp32
        else:
            y = F_.add(hidden_states, att)
        m = self.mlp(self.post_attention_layernorm(y))
        if self.scale_normalization:
            return self.fourth_layernorm(m, residual=y)
        return F_.add(y, m)


def unscaled_init_method(sigma):
    def init_(tensor):
        return torch.nn.init.normal_(tensor, mean=0.0, std=sigma)
    return init_


def scaled_init_method(sigma, num_layers):
    std = sigma / math.sqrt(2.0 * num_layers)

    def init_(tensor):
        return torch.nn.init.normal_(tensor, mean=0.0, std=std)
    return init_


class GPT2ParallelTransformer(torch.nn.Module):
    """mpu/sparse_transformer.py:361-626."""

    def __init__(self, num_layers, hidden_size, num_attention_heads, max_sequence_length, max_memory_length,
                 embedding_dropout_prob, attention_dropout_prob, output_dropout_prob, checkpoint_activations,
                 checkpoint_num_layers=1, layernorm_epsilon=1.0e-5, init_method_std=0.02,
                 use_scaled_init_for_output_weights=True, query_window=128, key_window_times=6, num_pivot=768,
                 kv_cache=False):
        super().__init__()
        # kv_cache (extension): the memories returned / accepted by forward are per-layer
        # key/value caches [b, s_mem, 2 * hidden/p] instead of the reference's layer inputs [b, s_mem, hidden]
        self.kv_cache = bool(kv_cache)
        self.checkpoint_activations = checkpoint_activations
        self.checkpoint_num_layers = checkpoint_num_layers
        self.max_memory_length = max_memory_length
        self.max_sequence_length = max_sequence_length
        output_layer_init_method = None
        if use_scaled_init_for_output_weights:
            output_layer_init_method = scaled_init_method(init_method_std, num_layers)
        self.embedding_dropout = _Dropout(embedding_dropout_prob)
        self.position_embeddings = torch.nn.Embedding(max_sequence_length, hidden_size)
        torch.nn.init.normal_(self.position_embeddings.weight, mean=0.0, std=init_method_std)
        self.query_window, self.key_window_times, self.num_pivot = query_window, key_window_times, num_pivot
        self.layers = torch.nn.ModuleList([
            GPT2ParallelTransformerLayer(hidden_size, num_attention_heads, attention_dropout_prob, output_dropout_prob,
                                         layernorm_epsilon, unscaled_init_method(init_method_std),
                                         output_layer_init_method=output_layer_init_method, query_window=query_window,
                                         key_window_times=key_window_times, scale_normalization=True)
            for _ in range(num_layers)])
        for i, layer in enumerate(self.layers):
            layer._cogv_index = i              # backward defers weight gradients to groups of layers (functional.py)
        self.final_layernorm = LayerNorm(hidden_size, eps=layernorm_epsilon)
        self.rmask = None
        self.on_layer_backward_done = None      # set by the data-parallel wrapper to overlap the all-reduce
        self.on_layer_forward_start = None      # set by the data-parallel wrapper: parameters of layer i must have arrived

    def embed(self, input_ids, position_ids, word_embeddings):
        """word + position embedding + embedding dropout in ONE kernel (mpu/layers.py:117-133 and
        mpu/sparse_transformer.py:522-524); returns hidden states carrying their abs-max slot."""
        drop = F_._drop(self.embedding_dropout.p, self.training)
        return F_.embedding(input_ids, word_embeddings.weight, word_embeddings.vocab_start_index, position_ids,
                            self.position_embeddings.weight, drop, stream=True)

    def forward(self, hidden_states, position_ids, attention_mask, txt_indices_bool, img_indices_bool, is_sparse=0,
                *mems, embedded=False):
        is_sparse = int(is_sparse)
        batch_size, query_length = hidden_states.size()[:2]
        memory_length = mems[0].size(1) if mems else 0
        key_length = query_length + memory_length
        if is_sparse == 1:
            sep = 0                              # the sparse training form has its own mask rule (rmask + window)
        elif isinstance(attention_mask, torch.Tensor) and attention_mask.numel() > 1:
            sep = F_.mask_to_sep(attention_mask, query_length, key_length)
            if sep is None:
                sep = attention_mask     # an arbitrary mask tensor: handed to the layers as it is (general-mask attention path)
        else:
            sep = int(attention_mask) if not isinstance(attention_mask, torch.Tensor) else int(attention_mask.item())
        if not embedded:
            # generic entry (hidden_states = word embeddings): position add + dropout, op by op
            pos = F_.embedding(position_ids, self.position_embeddings.weight, 0)
            hidden_states = F_.add(hidden_states, pos.expand_as(hidden_states).contiguous())
            hidden_states = F_.dropout(hidden_states, self.embedding_dropout.p, self.training)
        if hidden_states.dtype != torch.float32:
            hidden_states = hidden_states.float()        # from here on: the fp32 residual stream
        mem_layers = [hidden_states.detach()] if self.max_memory_length > 0 else []
        recompute = bool(self.checkpoint_activations) and torch.is_grad_enabled()
        if is_sparse == 1:
            # sparse training (mpu/sparse_transformer.py:492-505, 553-570): one pivot draw -- all text positions plus a
            # random sample of the image positions -- per chunk of checkpoint_num_layers layers.  The reference REQUIRES
            # activation checkpointing here to fit its gathered copies; these layers keep their activations instead
            # (no gathered copies exist on this path, and the dense fused layer is the one that recomputes).
            import random
            assert key_length == query_length and not mems
            if query_length % self.query_window:
                raise ValueError("sparse attention training needs the sequence length to be a multiple of query_window")
            img_indices = [img_indices_bool[i].nonzero(as_tuple=False).view(-1) for i in range(batch_size)]
            txt_indices = [txt_indices_bool[i].nonzero(as_tuple=False).view(-1) for i in range(batch_size)]
            num_pivot = self.num_pivot
        if is_sparse == 2:
            # sparse inference (mpu/sparse_transformer.py:497-499, 511-518, 586-600): every layer draws its own pivots --
            # all text positions plus a random sample of the image positions left of the trailing window
            import random
            left = max(0, key_length - self.key_window_times * self.query_window)
            window_idx = torch.arange(left, key_length, device=hidden_states.device, dtype=torch.long).expand(batch_size, -1)
            img_indices = [img_indices_bool[i][:left].nonzero(as_tuple=False).view(-1) for i in range(batch_size)]
            txt_indices = [txt_indices_bool[i][:left].nonzero(as_tuple=False).view(-1) for i in range(batch_size)]
            ratio = self.num_pivot / self.max_sequence_length
            max_text_num = max(len(t) for t in txt_indices)
            num_pivot = max_text_num + int((left - max_text_num) * ratio)
        kv_mode = self.kv_cache and self.max_memory_length > 0 and is_sparse == 0 and not torch.is_grad_enabled()
        if kv_mode:
            mem_layers = []
        for i, layer in enumerate(self.layers):
            mem_i = mems[i] if mems else None
            if self.on_layer_forward_start is not None:
                self.on_layer_forward_start(i)
            if kv_mode:
                slot = KVCacheSlot(mem_i, self.max_memory_length)
                hidden_states = layer(hidden_states, sep, mem=slot)
                mem_layers.append(slot.out)
                continue
            if is_sparse == 1:
                if i % max(1, int(self.checkpoint_num_layers)) == 0:
                    pivot_idx = torch.stack([
                        torch.cat((text_idx, img_indices[j][torch.tensor(
                            random.sample(range(len(img_indices[j])), k=num_pivot - len(text_idx)), dtype=torch.long,
                            device=text_idx.device)]), dim=0)
                        for j, text_idx in enumerate(txt_indices)])
                    plan = F_.sparse_pivot_plan(pivot_idx, query_length, self.query_window, self.key_window_times)
                hidden_states = layer(hidden_states, None, plan, is_sparse)
                continue
            if is_sparse == 2:
                pivot_idx = torch.stack([
                    torch.cat((text_idx, img_indices[j][torch.tensor(
                        random.sample(range(len(img_indices[j])), k=num_pivot - len(text_idx)), dtype=torch.long,
                        device=text_idx.device)]), dim=0)
                    for j, text_idx in enumerate(txt_indices)])
                pw_idx = torch.cat((pivot_idx, window_idx), dim=-1)
                hidden_states = layer(hidden_states, sep, pw_idx, is_sparse, mem=mem_i)
                if self.max_memory_length > 0:
                    mem_layers.append(hidden_states.detach())
                continue
            hidden_states = layer(hidden_states, sep, mem=mem_i, recompute=recompute,
                                  on_backward_done=self.on_layer_backward_done)
            if self.max_memory_length > 0:
                mem_layers.append(hidden_states.detach())
        output = self.final_layernorm(hidden_states)
        if kv_mode:       # same count as the reference's memories (layer inputs + final output): the last one is empty
            mem_layers.append(mem_layers[0].new_empty((mem_layers[0].size(0), mem_layers[0].size(1), 0)))
        elif self.max_memory_length > 0:
            mem_layers = self.update_mems(mem_layers, mems)
        return (output, *mem_layers)

    def update_mems(self, hiddens, mems):
        """mpu/sparse_transformer.py:615-626."""
        memory_length = mems[0].size(1) if mems else 0
        query_length = hiddens[0].size(1)
        new_memory_length = min(self.max_memory_length, memory_length + query_length)
        new_mems = []
        with torch.no_grad():
            for i in range(len(hiddens)):
                if new_memory_length <= query_length:
                    new_mems.append(hiddens[i][:, -new_memory_length:])
                else:
                    new_mems.append(torch.cat((mems[i][:, -new_memory_length + query_length:], hiddens[i]), dim=1))
        return new_mems
