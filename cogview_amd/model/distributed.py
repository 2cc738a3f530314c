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
import torch
import torch.distributed as dist
from torch.nn.modules import Module

from .. import mpu
from ..arena import arena_of, flatten_module


class DistributedDataParallel(Module):

    def __init__(self, module, device_ids=None, output_device=None, process_group=None, overlap=True,
                 bucket_layers=4, force_collectives=False):
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
        # force_collectives: run the full bucketed / overlapped exchange even in a group of one rank (single-GPU
        # test of the RCCL plumbing; a one-rank mean all-reduce is the identity)
        self.force = bool(force_collectives)
        self.overlap = overlap and self.arena is not None and (self.world > 1 or self.force)
        self.bucket_layers = max(1, bucket_layers)
        self._comm_stream = None
        self._pending, self._reduced_upto, self._layers_done = [], None, 0
        self._buckets = self._plan_buckets() if self.overlap else []
        tr = self._transformer()
        if self.overlap and tr is not None:
            tr.on_layer_backward_done = self._on_layer_done

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
        for hi in range(len(layers), 0, -self.bucket_layers):
            lo = max(0, hi - self.bucket_layers)
            ps = [p for l in layers[lo:hi] for p in l.parameters()]
            buckets.append((lo, self.arena.slice_of(ps)))
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

    def _launch(self, s, e):
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream()
        self._comm_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self._comm_stream):
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

    def forward(self, *inputs, **kwargs):
        self.needs_reduction = True
        self._layers_done = 0
        return self.module(*inputs, **kwargs)

    def state_dict(self, destination=None, prefix='', keep_vars=False):
        return self.module.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars)

    def load_state_dict(self, state_dict, strict=True):
        self.module.load_state_dict(state_dict, strict=strict)


class PyTorchDistributedDataParallel(DistributedDataParallel):
    """Same engine; with this class the exchange is finished automatically at the end of backward by
    `finish_gradient_sync()` (called by FP16_Optimizer.update_master_grads / the training step), mirroring
    torch-DDP's implicit synchronisation in the reference's default (USE_TORCH_DDP) configuration."""

    def finish_gradient_sync(self):
        self.allreduce_params(reduce_after=False)
