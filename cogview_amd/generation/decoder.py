"""Single-token decode steps as ONE captured HIP graph.

An eager decode step of the 4B model is ~900 kernel launches issued from Python: 16.6 ms per token on an MI355X whose
weight-read floor is 1.6 ms (tools/mb_decode.py) -- the step is launch bound, which is what graph capture is for.  To make
a step replayable nothing in it may depend on the current length:
  * the key/value caches have a fixed capacity; the new keys / values land at a device-side position (index_copy_);
  * attention runs over ALL slots through the gathered form (cogv_attn_desc.kv_index); slots not written yet carry the
    masked flag (bit 31) in the index table, which is device data updated between replays;
  * Sandwich-LN's abs-max slots come from a slab at fixed addresses that the graph's first node clears;
  * token and position are read from static device buffers.
Reference path: generation/sampling.py:139-148 (one model call per generated token, layer-input memories)."""
import torch

from .. import functional as F_
from .. import ops
from ..mpu.transformer import StaticKVSlot


class GraphDecoder:
    def __init__(self, model, batch=1, capacity=1152):
        """model: GPT2Model (optionally inside FP16_Module) in eval mode, dense attention; capacity: slots per cache
        (<= 4096, the gathered form's limit)."""
        m = model
        while hasattr(m, "module"):
            m = m.module
        self.gpt, tr = m, m.transformer
        assert capacity <= 4096
        p0 = tr.layers[0].attention.query_key_value.weight
        hp = tr.layers[0].attention.hidden_size_per_partition
        dev, dt = p0.device, p0.dtype
        self.batch, self.cap, self.length = batch, capacity, 0
        self.tok = torch.zeros((batch, 1), dtype=torch.long, device=dev)
        self.pos = torch.zeros((batch, 1), dtype=torch.long, device=dev)
        self.pos_index = torch.zeros(1, dtype=torch.long, device=dev)
        self.table = torch.arange(capacity, dtype=torch.int64, device=dev)
        self.table = ((self.table | (1 << 31)) - (1 << 32)).to(torch.int32).unsqueeze(0).repeat(batch, 1).contiguous()
        self.caches = [torch.zeros((batch, capacity, 2 * hp), dtype=dt, device=dev) for _ in tr.layers]
        self.slots = [StaticKVSlot(c, self.pos_index, self.table) for c in self.caches]
        self.slab = torch.zeros(8 * len(tr.layers) + 16, dtype=torch.float32, device=dev)
        self.graph, self.logits = None, None
        self.fused = True            # False: layer-by-layer path (every LayerNorm its own launch) -- kept for comparison

    # ------------------------------------------------------------------ one decode step, eager (also what gets captured)
    def _step(self):
        tr = self.gpt.transformer
        self.slab.zero_()
        with ops.scalar_slab(self.slab):
            h = tr.embed(self.tok, self.pos, self.gpt.word_embeddings)
            if self.fused and F_.decode_chain_supported(tr, self.batch):
                # five launches per layer: the LayerNorms ride as prologues of the GEMVs, the cache append inside the
                # decode attention kernel (functional.decode_chain)
                return F_.decode_chain(tr, h, h._cogv_absmax, self.slots, self.gpt.word_embeddings.weight)
            for layer, slot in zip(tr.layers, self.slots):
                h = layer(h, 0, mem=slot)
            out = tr.final_layernorm(h)
            return F_.tied_logits(out, self.gpt.word_embeddings.weight)

    def capture(self):
        """Warm up on a side stream (first-use allocations inside the library and the caching allocator), then capture."""
        assert 0 < self.length < self.cap, "prefill first: the warm-up steps write at the current position"
        self.pos_index.fill_(self.length)     # warm-up and capture runs scribble on the next (still masked) slot only
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(3):
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.logits = self._step()

    # ------------------------------------------------------------------ cache management (outside the graph)
    @torch.no_grad()
    def prefill(self, tokens, position_ids, attention_mask=0):
        """Run the context through the model once (eager, K/V-cache memories) and load the caches.  Returns its logits."""
        assert tokens.shape[0] == self.batch and tokens.shape[1] < self.cap
        tr = self.gpt.transformer
        kv_flag, tr.kv_cache = tr.kv_cache, True
        max_mem, tr.max_memory_length = tr.max_memory_length, max(tr.max_memory_length, self.cap)
        try:
            logits, *mems = self.gpt(tokens, position_ids, attention_mask, None, None, 0)
        finally:
            tr.kv_cache, tr.max_memory_length = kv_flag, max_mem
        n = tokens.shape[1]
        for c, mem in zip(self.caches, mems):
            c[:, :n].copy_(mem)
        self.table[:, :n] = torch.arange(n, dtype=torch.int32, device=self.table.device)
        self.length = n
        return logits

    @torch.no_grad()
    def step(self, token, position):
        """token, position: [batch, 1] (or [batch]) device tensors for the next input.  Returns logits [batch, 1, vocab]
        (a static buffer when the step is captured: consume it before the next call)."""
        assert self.length < self.cap, "key/value cache capacity exhausted"
        self.tok.copy_(token.view(self.batch, 1))
        self.pos.copy_(position.view(self.batch, 1))
        self.pos_index.fill_(self.length)
        self.table[:, self.length] = self.length          # this step's own slot becomes visible
        if self.graph is not None:
            self.graph.replay()
            logits = self.logits
        else:
            logits = self._step()
        self.length += 1
        return logits
