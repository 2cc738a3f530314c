"""Data-parallel wrappers with the reference's names (model/distributed.py:26-101).

Both classes are the same MI355X-native implementation: the module's parameters are moved into a flat
arena, so the gradient exchange is a handful of large all-reduces on contiguous slices (RCCL over xGMI is
per-link bound: few, large messages), issued on a side stream AS EACH LAYER'S BACKWARD FINISHES (the fused
layer Function reports completion), i.e. overlapped with the rest of backward exactly like torch-DDP's
buckets in the reference's default configuration -- without autograd hooks or bucket copies.

    PyTorchDistributedDataParallel(module, device_ids=..., output_device=..., process_group=...)
        reference: subclass of torch DDP with an unwrapped state_dict (model/distributed.py:26-32)
    DistributedDataParallel(module)
        reference: broadcast parameters at construction, explicit .allreduce_params(reduce_after, no_scale,
        fp32_allreduce) and .needs_reduction (model/distributed.py:35-76)
"""
import weakref

import torch
import torch.distributed as dist
from torch.nn.modules import Module

from .. import mpu
from ..arena import arena_of, flatten_module


class DistributedDataParallel(Module):

    def __init__(self, module, device_ids=None, output_device=None, process_group=None, overlap=True,
                 bucket_layers=4, force_collectives=False, shard_optimizer=False):
        super().__init__()
        self.module = module
        self.data_parallel_group = process_group if process_group is not None else mpu.get_data_parallel_group()
        self.world = dist.get_world_size(group=self.data_parallel_group)
        params = [p for p in module.parameters()]
        self.arena = arena_of(params)
        if self.arena is None and params and params[0].is_cuda:
            self.arena = flatten_module(module)
        # every replica starts from data-parallel rank 0's parameters (model/distributed.py:43-46)
        src = dist.get_global_rank(self.data_parallel_group, 0) if hasattr(dist, "get_global_rank") else \
            mpu.get_model_parallel_rank()
        if self.world > 1:
            if self.arena is not None:
                dist.broadcast(self.arena.data, src, group=self.data_parallel_group)
            else:
                for p in params:
                    dist.broadcast(p.data, src, group=self.data_parallel_group)
        self.needs_reduction = False
        # torch-DDP's contract -- the exchange finishes by itself at the end of backward -- for a wrapper constructed the way the
        # reference constructs torch's DDP (pretrain_gpt2.py:100-103: device_ids=[i], output_device=i); see
        # PyTorchDistributedDataParallel.  A native construction (this package's training step, bench.py, the tests) finishes
        # the exchange itself, at a point of its choosing, and the end of its backward pass is left alone.
        self.auto_sync = bool(type(self).auto_sync_when_constructed_like_torch_ddp and device_ids is not None)
        self._sync_consumer = False          # an FP16_Optimizer that finishes the exchange in update_master_grads() (attach_data_parallel)
        if self.auto_sync and self.arena is not None:
            self.arena.data_parallel_wrapper = weakref.ref(self)      # FP16_Optimizer.__init__ attaches itself to it
        # force_collectives: run the full bucketed / overlapped exchange even in a group of one rank (single-GPU
        # test of the RCCL plumbing; a one-rank mean all-reduce is the identity)
        self.force = bool(force_collectives)
        self.overlap = overlap and self.arena is not None and (self.world > 1 or self.force)
        self.bucket_layers = max(1, bucket_layers)
        self._comm_stream = None
        self._pending, self._reduced_upto, self._layers_done = [], None, 0
        self._layer_bucket_end = {}
        self._buckets = self._plan_buckets() if self.overlap else []
        # shard_optimizer: the exchange becomes reduce-scatter (gradients) + all-gather (updated 16-bit parameters)
        # and every rank runs the optimizer on its 1/N of each region only -- see the class docstring of ShardPlan
        self.shard = ShardPlan(self) if (shard_optimizer and self.arena is not None and (self.world > 1 or self.force)) else None
        tr = self._transformer()
        if self.overlap and tr is not None:
            tr.on_layer_backward_done = self._on_layer_done
        if self.shard is not None and tr is not None:
            tr.on_layer_forward_start = self._on_layer_forward_start

    # ------------------------------------------------------------------ bucket plan (layer order, reversed)
    def _transformer(self):
        m = self.module
        while hasattr(m, "module"):
            m = m.module
        return getattr(m, "transformer", None)

    def _plan_buckets(self):
        tr = self._transformer()
        if tr is None:
            return []
        layers = list(tr.layers)
        buckets = []
        self._layer_bucket_end = {}
        for hi in range(len(layers), 0, -self.bucket_layers):
            lo = max(0, hi - self.bucket_layers)
            ps = [p for l in layers[lo:hi] for p in l.parameters()]
            buckets.append((lo, self.arena.slice_of(ps)))
            for i in range(lo, hi):
                self._layer_bucket_end[i] = buckets[-1][1][1]
        return buckets      # bucket k becomes ready when layer index `lo` has finished its backward

    def _on_layer_done(self, layer):
        if not self.needs_reduction:
            return
        tr = self._transformer()
        self._layers_done += 1
        done_index = len(tr.layers) - self._layers_done          # backward visits layers in reverse
        for lo, (s, e) in self._buckets:
            if lo == done_index:
                self._launch(s, e)

    def _on_layer_forward_start(self, i):
        """Sharded exchange: the all-gather of the updated parameters runs on the side stream in forward order; layer i
        may start once the region that holds it has arrived (the last layer also waits for the final LayerNorm)."""
        if not self.shard.pending():
            return
        n = len(self._transformer().layers)
        end = self.arena.total if i == n - 1 else self._layer_bucket_end.get(i, self.arena.total)
        self.shard.wait_upto(end)

    def _launch(self, s, e):
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        self._comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._comm_stream):
            if self.shard is not None:
                self.shard.reduce_region(s, e)
            else:
                self._allreduce_mean(self.arena.grad[s:e])
        self._pending.append((s, e))

    def _allreduce_mean(self, g):
        """Mean over the data-parallel group.  RCCL averages inside the collective (no extra pass over the
        gradients); gloo (CPU tests) has no AVG, so pre-divide there -- which is also the reference's
        `reduce_after=False` order (model/distributed.py:69-71)."""
        if dist.get_backend(self.data_parallel_group) == "nccl":
            dist.all_reduce(g, op=dist.ReduceOp.AVG, group=self.data_parallel_group)
        else:
            g.div_(self.world)
            dist.all_reduce(g, group=self.data_parallel_group)

    # ------------------------------------------------------------------ reference API
    def allreduce_params(self, reduce_after=True, no_scale=False, fp32_allreduce=False):
        """Finish the gradient exchange: slices already reduced during backward are skipped, the rest
        (embeddings, final LayerNorm, or everything when overlap is off) is reduced now."""
        if not self.needs_reduction:
            return
        self.needs_reduction = False
        self._layers_done = 0
        if self.arena is not None:
            self.arena.finish_lazy()          # gradients no kernel wrote since a lazy zero_grad become real zeros
        if self.world == 1 and not self.force:
            return
        if self.arena is None:
            for p in self.module.parameters():
                if p.requires_grad and p.grad is not None:
                    if not no_scale and not reduce_after:
                        p.grad.data.div_(self.world)
                    dist.all_reduce(p.grad.data, group=self.data_parallel_group)
                    if not no_scale and reduce_after:
                        p.grad.data.div_(self.world)
            return
        # the exchange below moves slices of arena.grad: every parameter's .grad must still be its view of that buffer
        # (an optimizer that set .grad = None makes backward allocate loose tensors the exchange would never see)
        esz = self.arena.grad.element_size()
        base = self.arena.grad.data_ptr()
        # (772 parameters at 48 layers: ~0.2 ms of host pointer compares per step, hidden behind the queued backward)
        for p, off in zip(self.arena.params, self.arena.offsets):
            if p.grad is not None and p.grad.data_ptr() != base + off * esz:
                self.arena.grad[off:off + p.numel()].view(p.shape).copy_(p.grad)
                p.grad = self.arena.grad[off:off + p.numel()].view(p.shape)
                if self._pending:
                    raise RuntimeError("a parameter gradient left the flat arena while overlapped gradient slices were "
                                       "already reduced: zero gradients with arena.zero_grad(), not by setting .grad = None")
        done = sorted(self._pending)
        self._pending = []
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)
        cur, rest = 0, []
        for s, e in done:
            if s > cur:
                rest.append((cur, s))
            cur = max(cur, e)
        if cur < self.arena.total:
            rest.append((cur, self.arena.total))
        if self.shard is not None:
            if getattr(self, "_shard_consumer", None) is None:
                # after a reduce-scatter only the owner's slice of each region holds the mean gradient and the parameters
                # only become consistent again through the optimizer's all-gather: a plain optimizer (or mpu.clip_grad_norm
                # over model.parameters()) would read partial sums and the replicas would silently diverge
                raise RuntimeError("DistributedDataParallel(shard_optimizer=True) needs an FP16_Optimizer attached with "
                                   "optimizer.attach_data_parallel(ddp) before the first gradient exchange")
            if fp32_allreduce or no_scale:
                raise NotImplementedError("shard_optimizer exchanges mean gradients in their 16-bit storage type")
            for s, e in rest:
                for rs, re_ in self.shard.split(s, e):
                    self.shard.reduce_region(rs, re_)
            return
        for s, e in rest:
            g = self.arena.grad[s:e]
            if not fp32_allreduce and not no_scale:
                self._allreduce_mean(g)
                continue
            buf = g.float() if fp32_allreduce else g
            if not no_scale and not reduce_after:
                buf.div_(self.world)
            dist.all_reduce(buf, group=self.data_parallel_group)
            if not no_scale and reduce_after:
                buf.div_(self.world)
            if fp32_allreduce:
                g.copy_(buf)

    auto_sync_when_constructed_like_torch_ddp = False       # PyTorchDistributedDataParallel: True

    def forward(self, *inputs, **kwargs):
        self.needs_reduction = True
        self._layers_done = 0
        if self.shard is not None and self.shard.pending():
            # embeddings (everything in front of the first layer bucket) now, the layers as the forward reaches them
            first = min((s for _, (s, _e) in self._buckets), default=self.arena.total)
            self.shard.wait_upto(first if self._transformer() is not None else self.arena.total)
            out = self.module(*inputs, **kwargs)
            self.shard.wait_upto(self.arena.total)
        else:
            out = self.module(*inputs, **kwargs)
        if self.auto_sync and not self._sync_consumer and torch.is_grad_enabled() and (self.world > 1 or self.force):
            self._arm_end_of_backward(out)   # nobody downstream will finish the exchange (fp32 training: plain loss.backward())
        return out

    def _arm_end_of_backward(self, out):
        """A hook on the first output that carries a gradient: when the backward pass reaches it (its very beginning), the end of
        that pass is given the callback that finishes the gradient exchange."""
        outs = out if isinstance(out, (tuple, list)) else (out,)
        first = next((t for t in outs if isinstance(t, torch.Tensor) and t.requires_grad), None)
        if first is None:
            return

        def on_backward_start(grad):
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
            return grad

        first.register_hook(on_backward_start)

    def _end_of_backward(self):
        from .. import functional as F_
        F_.flush_weight_grads(final=True)            # no weight gradient may still be queued when its region is exchanged
        self.allreduce_params(reduce_after=False)    # a no-op if the caller has already finished the exchange itself

    def state_dict(self, destination=None, prefix='', keep_vars=False):
        if self.shard is not None:
            self.shard.wait_upto(self.arena.total)
        return self.module.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars)

    def load_state_dict(self, state_dict, strict=True):
        self.module.load_state_dict(state_dict, strict=strict)


class ShardPlan:
    """Data-parallel exchange in the reduce-scatter / all-gather form (what DeepSpeed's ZeRO stage 1 does for the
    reference, scripts/ds_config_zero.json:6-14), laid out for the flat arena and for xGMI:

      * the arena is cut into REGIONS -- the layer buckets of the overlapped exchange plus the gaps between them
        (embeddings, final LayerNorm); rank r of the N data-parallel ranks OWNS the r-th 1/N of every region (cut at
        multiples of 128 elements; the < 128 N leftover elements of a region belong to the last rank);
      * backward: each region is REDUCE-SCATTERED (mean) as soon as it is complete -- one in-place RCCL call per region,
        (N-1)/N of the bytes of an all-reduce's first half and no second half;
      * the fused overflow / norm pass and the AdamW pass touch the owned slices only (1/N of the optimizer's 30 B per
        parameter of HBM traffic); the two statistics are summed over the ranks in one 16-byte all-reduce;
      * the updated 16-bit parameters are ALL-GATHERED region by region in forward order on the side stream.

    The fp32 master / moment buffers stay allocated full-size (16 B per parameter = 63 GB at 4B: a rounding error in 288
    GB of HBM3E, and it keeps `state_dict()` in the reference's layout); each rank simply never reads or writes the
    slices it does not own, and `FP16_Optimizer.consolidate_state()` (a collective, called by `utils.save_checkpoint` on
    every rank) refreshes them with one all-gather per buffer before saving."""

    ALIGN = 128

    def __init__(self, ddp):
        self.ddp, self.world = ddp, ddp.world
        self.group = ddp.data_parallel_group
        self.rank = dist.get_rank(group=self.group)
        total = ddp.arena.total
        cuts = sorted({0, total} | {x for _, (s, e) in ddp._buckets for x in (s, e)})
        self.regions = [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]
        self._events = []                    # (region end, event) of all-gathers the compute stream has not waited for

    def split(self, s, e):
        """The regions that tile [s, e) (a gap of the bucket plan may span several)."""
        out = [(a, b) for a, b in self.regions if a >= s and b <= e]
        assert out and out[0][0] == s and out[-1][1] == e, "range is not a union of regions"
        return out

    def _cut(self, s, e):
        per = (e - s) // (self.world * self.ALIGN) * self.ALIGN          # owned elements per rank in the main part
        return per, s + per * self.world                                  # (per-rank length, start of the leftover)

    def owned(self, rank=None):
        """Sorted, disjoint element ranges of the arena this rank owns."""
        r = self.rank if rank is None else rank
        out = []
        for s, e in self.regions:
            per, tail = self._cut(s, e)
            if per:
                out.append((s + r * per, s + (r + 1) * per))
            if tail < e and r == self.world - 1:
                out.append((tail, e))
        merged = []
        for a, b in out:
            if merged and merged[-1][1] == a:
                merged[-1] = (merged[-1][0], b)
            else:
                merged.append((a, b))
        return merged

    def reduce_region(self, s, e):
        """Mean gradient of region [s, e): afterwards the owner's slice holds the reduced values (other slices hold
        partial data and are never read)."""
        g = self.ddp.arena.grad
        per, tail = self._cut(s, e)
        nccl = dist.get_backend(self.group) == "nccl"
        if per:
            if nccl:
                dist.reduce_scatter_tensor(g[s + self.rank * per:s + (self.rank + 1) * per], g[s:tail],
                                           op=dist.ReduceOp.AVG, group=self.group)
            else:                                   # gloo (CPU tests, two ranks on one GPU) has no reduce-scatter
                g[s:tail].div_(self.world)
                dist.all_reduce(g[s:tail], group=self.group)
        if tail < e:
            self.ddp._allreduce_mean(g[tail:e])

    def gather_params(self):
        """All-gather the updated 16-bit parameters, region by region in forward order, on the side stream.  The compute
        stream does not wait here: wait_upto() is called as the next forward reaches each region, so the exchange of the
        later layers overlaps the forward pass of the earlier ones."""
        ddp, w = self.ddp, self.ddp.arena.data
        if ddp._comm_stream is None:
            ddp._comm_stream = torch.cuda.Stream()
        ddp._comm_stream.wait_stream(torch.cuda.current_stream())
        nccl = dist.get_backend(self.group) == "nccl"
        last = dist.get_global_rank(self.group, self.world - 1) if hasattr(dist, "get_global_rank") else self.world - 1
        with torch.cuda.stream(ddp._comm_stream):
            for s, e in self.regions:
                per, tail = self._cut(s, e)
                if per:
                    mine = w[s + self.rank * per:s + (self.rank + 1) * per]
                    if nccl:
                        dist.all_gather_into_tensor(w[s:tail], mine, group=self.group)
                    else:
                        parts = [w[s + r * per:s + (r + 1) * per] for r in range(self.world)]
                        dist.all_gather(parts, mine.clone(), group=self.group)
                if tail < e:
                    dist.broadcast(w[tail:e], last, group=self.group)
                ev = torch.cuda.Event()
                ev.record(ddp._comm_stream)
                self._events.append((e, ev))

    def pending(self):
        return bool(self._events)

    def wait_upto(self, end):
        """The compute stream waits for the all-gathers of every region that starts before element `end`."""
        cur = torch.cuda.current_stream()
        while self._events and self._events[0][0] <= end:
            cur.wait_event(self._events.pop(0)[1])

    def gather_state(self, flat):
        """Make a full-size fp32 state buffer consistent on every rank (checkpointing): owners broadcast their slices."""
        nccl = dist.get_backend(self.group) == "nccl"
        last = dist.get_global_rank(self.group, self.world - 1) if hasattr(dist, "get_global_rank") else self.world - 1
        for s, e in self.regions:
            per, tail = self._cut(s, e)
            if per:
                mine = flat[s + self.rank * per:s + (self.rank + 1) * per]
                if nccl:
                    dist.all_gather_into_tensor(flat[s:tail], mine, group=self.group)
                else:
                    parts = [flat[s + r * per:s + (r + 1) * per] for r in range(self.world)]
                    dist.all_gather(parts, mine.clone(), group=self.group)
            if tail < e:
                dist.broadcast(flat[tail:e], last, group=self.group)


class PyTorchDistributedDataParallel(DistributedDataParallel):
    """Same engine.  Constructed the way the reference constructs torch's DistributedDataParallel -- `DDP(model, device_ids=[i],
    output_device=i, process_group=...)`, pretrain_gpt2.py:100-103 -- the wrapper takes over torch-DDP's contract: the gradient
    exchange finishes WITHOUT anybody calling allreduce_params (with USE_TORCH_DDP = True pretrain_gpt2.backward_step :344-391
    never does).  Two mechanisms:
      * an FP16_Optimizer built afterwards on the same flat arena attaches itself (fp16.py), and its update_master_grads() --
        which backward_step calls right after backward -- finishes the exchange in the calling thread: the same code path as
        training.backward_step / attach_data_parallel(), which the two-process GPU tests exercise;
      * without such an optimizer (fp32 training: plain loss.backward()) a callback queued on the autograd engine finishes it
        at the end of every backward pass that reaches the module's output.
    Until round 5 only this package's own training.backward_step or an explicitly attached optimizer finished the exchange: the
    reference's train_step on two ranks left the embeddings' gradients unreduced and the replicas diverged
    (tests/ref_drivers/drive_pretrain_gpt2_dp2.py: two gloo ranks on the CPU-emulated ops; the reference-style construction has
    NOT yet run on GPUs -- the round's GPU budget was spent when this was found).
    Constructed natively (no device_ids: training.train_step, bench.py, the GPU tests) nothing changes."""
    auto_sync_when_constructed_like_torch_ddp = True

    def finish_gradient_sync(self):
        self.allreduce_params(reduce_after=False)
