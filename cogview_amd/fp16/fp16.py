"""FP16_Module / FP16_Optimizer with the reference's API (fp16/fp16.py:59-629).

bf16 is accepted everywhere fp16 is (the reference raises TypeError on anything but HalfTensor,
fp16/fp16.py:194-216): `FP16_Module(module, dtype=torch.bfloat16)`.

FP16_Optimizer on MI355X: when the 16-bit model parameters live in a flat arena (FP16_Module puts them
there) and the wrapped optimizer is cogview_amd.optim.FusedAdam, the reference's five per-tensor passes
    overflow check | fp16->fp32 grad copy | /loss_scale | clip | Adam | fp32->fp16 param copy
(fp16/fp16.py:312-334,399-453,556-567; 388 / 772 tensors, two host syncs per tensor) collapse into
    cogv_grad_stats   one pass over the flat gradients: inf/nan flag + sum of squares
    cogv_adamw_step   one pass: unscale, clip coefficient from the device-side norm, AdamW on fp32 masters,
                      16-bit parameter write
with ONE host read per step (the overflow flag + norm, needed by the loss-scale state machine).
Any other combination (loose parameters, another inner optimizer) takes the generic path that mirrors the
reference step by step.
"""
import warnings

import torch
import torch.nn as nn

from .. import mpu
from .. import ops
from ..arena import arena_of, flatten_module
from ..optim import FusedAdam
from .fp16util import clip_grad_norm, master_params_to_model_params, model_grads_to_master_grads
from .loss_scaler import DynamicLossScaler, LossScaler

_HALF = (torch.float16, torch.bfloat16)


def conversion_helper(val, conversion):
    if not isinstance(val, (tuple, list)):
        return conversion(val)
    rtn = [conversion_helper(v, conversion) for v in val]
    return tuple(rtn) if isinstance(val, tuple) else rtn


def fp32_to_fp16(val, dtype=torch.float16):
    def conv(v):
        return v.to(dtype) if isinstance(v, torch.Tensor) and v.dtype == torch.float32 else v
    return conversion_helper(val, conv)


def fp16_to_fp32(val):
    def conv(v):
        return v.float() if isinstance(v, torch.Tensor) and v.dtype in _HALF else v
    return conversion_helper(val, conv)


class FP16_Module(nn.Module):
    """Casts the module to 16 bits, float inputs -> 16 bits, 16-bit outputs -> float (fp16/fp16.py:59-71).
    `keep_half_outputs=True` skips the output up-cast (the fused cross-entropy reads 16-bit logits and
    computes in fp32, so the 2x larger fp32 logits tensor never needs to exist)."""

    def __init__(self, module, dtype=torch.float16, keep_half_outputs=False, flatten=True):
        super().__init__()
        self.dtype = dtype
        self.keep_half_outputs = keep_half_outputs
        self.add_module('module', module.to(dtype))
        if flatten and any(p.is_cuda for p in module.parameters()):
            flatten_module(self.module)

    def forward(self, *inputs, **kwargs):
        out = self.module(*(fp32_to_fp16(inputs, self.dtype)), **kwargs)
        return out if self.keep_half_outputs else fp16_to_fp32(out)

    def state_dict(self, destination=None, prefix='', keep_vars=False):
        return self.module.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars)

    def load_state_dict(self, state_dict, strict=True):
        self.module.load_state_dict(state_dict, strict=strict)


class FP16_Optimizer(object):
    def __init__(self, init_optimizer, static_loss_scale=1.0, dynamic_loss_scale=False, dynamic_loss_args=None,
                 verbose=False):
        self.verbose = verbose
        self.optimizer = init_optimizer
        self.fp16_groups, self.fp32_from_fp16_groups, self.fp32_from_fp32_groups = [], [], []
        half_params = [p for g in self.optimizer.param_groups for p in g['params']
                       if p.requires_grad and p.dtype in _HALF]
        arena = arena_of(half_params) if half_params else None
        fused = (arena is not None and isinstance(self.optimizer, FusedAdam) and len(self.optimizer.param_groups) <= 8
                 and all(p.dtype in _HALF for g in self.optimizer.param_groups for p in g['params'] if p.requires_grad)
                 and len({id(p) for p in half_params}) == len(arena.params))
        self._arena = arena if fused else None
        if not fused and half_params:
            # never silent: the generic path keeps the reference's per-tensor semantics (fp16/fp16.py:322-...), with the norm /
            # overflow reductions on cogv_grad_stats but per-tensor copies and the inner optimizer's own update
            why = ("the 16-bit parameters are not views of one flat arena (build the model through FP16_Module / "
                   "arena.flatten_module)" if arena is None else
                   "the inner optimizer is %s, not cogview_amd's FusedAdam" % type(self.optimizer).__name__
                   if not isinstance(self.optimizer, FusedAdam) else
                   "more than 8 parameter groups" if len(self.optimizer.param_groups) > 8 else
                   "the optimizer mixes fp32 parameters with the 16-bit ones, or holds only part of the arena")
            warnings.warn("FP16_Optimizer: per-tensor path instead of the fused flat one (one statistics launch + one AdamW "
                          "launch per step): " + why, RuntimeWarning, stacklevel=2)
        if fused:
            self._master_flat = torch.empty(arena.total, dtype=torch.float32, device=arena.data.device)
            ops.cast_flat(arena.data, self._master_flat)
            self._m_flat = torch.zeros_like(self._master_flat)
            self._v_flat = torch.zeros_like(self._master_flat)
            self._stats = torch.zeros(2, dtype=torch.float64, device=arena.data.device)
            self._step_count = 0
        offsets = {id(p): off for p, off in zip(arena.params, arena.offsets)} if fused else {}
        group_of = {}
        for gi, param_group in enumerate(self.optimizer.param_groups):
            self.maybe_print("FP16_Optimizer processing param group {}:".format(gi))
            fp16_params, fp32_params, fp32_from_fp16 = [], [], []
            for i, param in enumerate(param_group['params']):
                if not param.requires_grad:
                    continue
                if param.dtype in _HALF:
                    if not param.is_cuda:
                        raise TypeError("FP16_Optimizer needs GPU parameters, got {}".format(param.type()))
                    fp16_params.append(param)
                    if fused:
                        off = offsets[id(param)]
                        master = self._master_flat[off:off + param.numel()].view(param.shape)
                        master.requires_grad = True
                        self.optimizer.state[master] = {
                            'exp_avg': self._m_flat[off:off + param.numel()].view(param.shape),
                            'exp_avg_sq': self._v_flat[off:off + param.numel()].view(param.shape)}
                        group_of[id(param)] = gi
                    else:
                        master = param.detach().clone().float()
                        master.requires_grad = True
                        if param in self.optimizer.state:
                            self.optimizer.state[master] = self.optimizer.state.pop(param)
                    master.model_parallel = getattr(param, 'model_parallel', False)
                    param_group['params'][i] = master
                    fp32_from_fp16.append(master)
                elif param.dtype == torch.float32:
                    fp32_params.append(param)
                else:
                    raise TypeError("Wrapped parameters must be float16, bfloat16 or float32 tensors. "
                                    "Received {}".format(param.type()))
            self.fp16_groups.append(fp16_params)
            self.fp32_from_fp16_groups.append(fp32_from_fp16)
            self.fp32_from_fp32_groups.append(fp32_params)
        if fused:
            mp_rank0 = (not mpu.model_parallel_is_initialized()) or mpu.get_model_parallel_rank() == 0
            self._group_of = lambda p: group_of[id(p)]
            self._norm_of = lambda p: bool(getattr(p, 'model_parallel', False)) or mp_rank0      # mpu/grads.py:61
            self._tables = arena.chunk_table(self._group_of, self._norm_of)
        else:
            self.optimizer.load_state_dict(self.optimizer.state_dict())
        if dynamic_loss_scale:
            self.dynamic_loss_scale = True
            self.loss_scaler = DynamicLossScaler(**dynamic_loss_args) if dynamic_loss_args is not None \
                else DynamicLossScaler()
        else:
            self.dynamic_loss_scale = False
            self.loss_scaler = LossScaler(static_loss_scale)
        self.overflow = False
        self._fwd_nan_dev, self._fwd_nan_host = None, None
        self.first_closure_call_this_step = True
        self.clip_grad_norm = clip_grad_norm
        self._clip, self._stats_valid, self._ddp, self._shard = 0.0, False, None, None
        # A data-parallel wrapper constructed the way the reference constructs torch's DDP (pretrain_gpt2.py:100-103) has
        # announced itself on the arena: the reference never introduces the optimizer to it (torch's DDP needs no introduction),
        # so it is attached here -- update_master_grads() then finishes the gradient exchange, in the calling thread, exactly
        # as after an explicit attach_data_parallel()
        wrapper = getattr(arena, 'data_parallel_wrapper', None) if fused else None
        wrapper = wrapper() if wrapper is not None else None
        if wrapper is not None and wrapper.auto_sync:
            self.attach_data_parallel(wrapper)

    # ---------------------------------------------------------------------------------------------- misc
    def maybe_print(self, msg):
        if self.verbose:
            print(msg)

    def attach_data_parallel(self, ddp):
        """Let update_master_grads() finish the overlapped gradient exchange first.  With a sharded exchange
        (DistributedDataParallel(shard_optimizer=True)) the statistics and AdamW passes are restricted to this rank's
        slices of the flat buffers and step() all-gathers the updated parameters."""
        self._ddp = ddp
        ddp._sync_consumer = True            # this optimizer finishes the exchange: the wrapper need not (model/distributed.py)
        shard = getattr(ddp, 'shard', None)
        if shard is not None:
            assert self._arena is not None and self._arena is ddp.arena, "sharding needs the fused flat optimizer path"
            self._shard = shard
            self._tables = self._arena.chunk_table(self._group_of, self._norm_of, owned=shard.owned())
            ddp._shard_consumer = self       # allreduce_params refuses to scatter gradients nobody will gather back

    def __getstate__(self):
        raise RuntimeError("FP16_Optimizer should be serialized using state_dict().")

    def __setstate__(self, state):
        raise RuntimeError("FP16_Optimizer should be deserialized using load_state_dict().")

    @property
    def lazy_zero_grad_ok(self):
        """Opt-in declared by the model (arena.lazy_ok <- module._cogv_lazy_zero_grad), not a property of the arena."""
        return self._arena is not None and self._arena.lazy_ok

    def finish_lazy_zero_grad(self):
        if self._arena is not None:
            self._arena.finish_lazy()

    def zero_grad(self, set_grads_to_None=False, lazy=False):
        """lazy (fused flat path only): no memset, the next backward pass overwrites instead of accumulating
        (arena.ParamArena.zero_grad) -- for callers that run backward immediately, like training.backward_step."""
        if self._arena is not None:
            self._arena.zero_grad(lazy=lazy)         # one memset (or none); param.grad stay views of the flat buffer
            self._stats_valid = False
            return
        for group in self.optimizer.param_groups:
            for p in group['params']:
                if set_grads_to_None:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.detach_()
                    p.grad.zero_()
        for fp16_group in self.fp16_groups:
            for param in fp16_group:
                if set_grads_to_None:
                    param.grad = None
                elif param.grad is not None:
                    param.grad.detach_()
                    param.grad.zero_()

    # ---------------------------------------------------------------------------------------------- fused pieces
    def _compute_stats(self):
        self._arena.finish_lazy()
        self._stats.zero_()
        ops.grad_stats(self._arena.grad, self._tables[0], self._tables[1], self._tables[3], self._stats)
        if self._shard is not None:          # every rank saw its own slices: sum of squares and overflow count add up
            torch.distributed.all_reduce(self._stats, group=self._shard.group)
        if mpu.model_parallel_is_initialized() and mpu.get_model_parallel_world_size() > 1:
            g = mpu.get_model_parallel_group()
            st = self._stats.clone()
            torch.distributed.all_reduce(st[0:1], group=g)                                    # mpu/grads.py:66
            torch.distributed.all_reduce(st[1:2], op=torch.distributed.ReduceOp.MAX, group=g)  # loss_scaler.py:119
            self._stats.copy_(st)
        if self._fwd_nan_dev is not None:            # the forward NaN guard's flag rides in the same read (training.train_step)
            both = torch.cat((self._stats, self._fwd_nan_dev)).tolist()
            self._host_stats, self._fwd_nan_host, self._fwd_nan_dev = both[:2], both[2] != 0.0, None
        else:
            self._host_stats = self._stats.tolist()  # the single host read of the step
        self._stats_valid = True

    def defer_forward_nan_flag(self, total_loss):
        """training.train_step(check_forward_nan=True): remember `not isfinite(total_loss)` as a device scalar; it reaches the
        host together with the gradient statistics (arena path) or at forward_was_nan()."""
        self._fwd_nan_dev = (~torch.isfinite(total_loss.detach()).all()).to(torch.float64).view(1)
        self._fwd_nan_host = None

    def forward_was_nan(self):
        if self._fwd_nan_dev is not None:            # not consumed by a statistics pass (no arena / static loss scale)
            self._fwd_nan_host, self._fwd_nan_dev = bool(self._fwd_nan_dev.item() != 0.0), None
        return bool(self._fwd_nan_host)

    def _check_overflow(self):
        if self._arena is not None:
            if not self._stats_valid:
                self._compute_stats()
            self.overflow = self._host_stats[1] != 0.0
            return
        params = [p for group in self.fp16_groups for p in group] + \
                 [p for group in self.fp32_from_fp32_groups for p in group]
        self.overflow = self.loss_scaler.has_overflow(params)

    def _update_scale(self, has_overflow=False):
        self.loss_scaler.update_scale(has_overflow)

    def _master_params_to_model_params(self):
        if self._arena is not None:
            ops.cast_flat_back(self._master_flat, self._arena.data)
            return
        for fp16_group, fp32_group in zip(self.fp16_groups, self.fp32_from_fp16_groups):
            master_params_to_model_params(fp16_group, fp32_group)

    def _model_params_to_master_params(self):
        if self._arena is not None:
            ops.cast_flat(self._arena.data, self._master_flat)
            return
        for fp16_group, fp32_group in zip(self.fp16_groups, self.fp32_from_fp16_groups):
            master_params_to_model_params(fp32_group, fp16_group)

    def _model_grads_to_master_grads(self):
        for fp16_group, fp32_group in zip(self.fp16_groups, self.fp32_from_fp16_groups):
            model_grads_to_master_grads(fp16_group, fp32_group)

    def _downscale_master(self):
        if self.loss_scale != 1.0:
            for group in self.optimizer.param_groups:
                for param in group['params']:
                    if param.grad is not None:
                        param.grad.data.mul_(1. / self.loss_scale)

    # ---------------------------------------------------------------------------------------------- reference API
    def clip_master_grads(self, max_norm, norm_type=2):
        """Returns the global gradient norm, or -1 after an overflow (fp16/fp16.py:312-334)."""
        if self.overflow:
            return -1
        if self._arena is not None:
            if float(norm_type) != 2.0:
                # other norms (fp16/fp16.py:312-334 hands norm_type through to clip_grad_norm): the unfused way -- the norm
                # of the unscaled gradients by mpu.clip_grad_norm's rule, the coefficient folded into the 16-bit gradients;
                # the fused step then runs without its own (2-norm) clipping
                if self._shard is not None:
                    # after the reduce-scatter only the owned slices hold mean gradients: a per-tensor norm would read
                    # partial sums.  (The fused 2-norm path restricts its chunk table to the owned ranges instead.)
                    raise NotImplementedError("norm_type != 2 is not available with the sharded optimizer exchange")
                self._arena.finish_lazy()
                params = [p for p in self._arena.params if p.grad is not None]      # the 16-bit model parameters (arena views)
                total = mpu.clip_grad_norm(params, float(max_norm) * self.loss_scale, norm_type) / self.loss_scale
                self._clip = 0.0
                self._stats_valid = False
                return total
            if not self._stats_valid:
                self._compute_stats()
            self._clip = float(max_norm)             # applied inside cogv_adamw_step from the device-side norm
            return (self._host_stats[0] ** 0.5) / self.loss_scale
        fp32_params = [p for g in self.optimizer.param_groups for p in g['params']]
        return self.clip_grad_norm(fp32_params, max_norm, norm_type)

    def consolidate_state(self):
        """COLLECTIVE over the data-parallel group (every rank must call it): with a sharded exchange each rank updates
        only its slices of the fp32 master / moment buffers; this refreshes the slices the other ranks own, after which
        state_dict() is the full, reference-layout state on every rank.  A no-op otherwise.  utils.save_checkpoint calls
        it on all ranks before data-parallel rank 0 writes the file."""
        if self._shard is not None:
            for flat in (self._master_flat, self._m_flat, self._v_flat):
                self._shard.gather_state(flat)

    def state_dict(self):
        if self._arena is not None:
            for g in self.optimizer.param_groups:     # where the generic path / apex FusedAdam keep the step
                g['step'] = self._step_count
        return {'loss_scaler': self.loss_scaler, 'dynamic_loss_scale': self.dynamic_loss_scale,
                'overflow': self.overflow, 'first_closure_call_this_step': self.first_closure_call_this_step,
                'optimizer_state_dict': self.optimizer.state_dict(), 'fp32_from_fp16': self.fp32_from_fp16_groups,
                'cogv_step_count': getattr(self, '_step_count', None)}

    def load_state_dict(self, state_dict):
        self.loss_scaler = state_dict['loss_scaler']
        self.dynamic_loss_scale = state_dict['dynamic_loss_scale']
        self.overflow = state_dict['overflow']
        self.first_closure_call_this_step = state_dict['first_closure_call_this_step']
        if self._arena is None:
            self.optimizer.load_state_dict(state_dict['optimizer_state_dict'])
        else:
            saved = state_dict['optimizer_state_dict']
            flat_params = [p for g in self.optimizer.param_groups for p in g['params']]
            ids = [i for g in saved['param_groups'] for i in g['params']]
            for p, i in zip(flat_params, ids):
                st = saved['state'].get(i)
                if st is not None:
                    self.optimizer.state[p]['exp_avg'].copy_(st['exp_avg'])
                    self.optimizer.state[p]['exp_avg_sq'].copy_(st['exp_avg_sq'])
            for g, sg in zip(self.optimizer.param_groups, saved['param_groups']):
                for k, v in sg.items():
                    if k != 'params':
                        g[k] = v
            if state_dict.get('cogv_step_count') is not None:
                self._step_count = state_dict['cogv_step_count']
            else:
                # a checkpoint written by the generic path or by the reference (apex FusedAdam): the step lives in
                # the param groups, or (older apex / torch.optim.Adam) in the per-parameter state
                step = saved['param_groups'][0].get('step') if saved['param_groups'] else None
                if step is None:
                    steps = [st['step'] for st in saved['state'].values() if isinstance(st, dict) and 'step' in st]
                    step = steps[0] if steps else 0
                self._step_count = int(step)
        for current_group, saved_group in zip(self.fp32_from_fp16_groups, state_dict['fp32_from_fp16']):
            for current, saved in zip(current_group, saved_group):
                current.data.copy_(saved.data)

    def step(self, closure=None):
        """fp16/fp16.py:399-453: update the loss scale, skip on overflow, else optimizer step + master->model."""
        scale = self.loss_scaler.loss_scale
        self._update_scale(self.overflow)
        if self.overflow:
            self.maybe_print("OVERFLOW! Skipping step. Attempted loss scale: {}, reducing to {}".format(
                scale, self.loss_scale))
            return
        if self._arena is not None:
            retval = None
            if closure is not None:
                # fp16/fp16.py:431-453 for an inner optimizer that evaluates its closure once per step (Adam): the closure
                # (zero_grad + forward + self.backward(loss)) is evaluated here, again with a reduced scale while its
                # gradients overflow; the fused update then uses the scale that last evaluation ran with
                self.first_closure_call_this_step = False
                retval = closure()
                while self.overflow:
                    bad = self.loss_scaler.loss_scale
                    self._update_scale(self.overflow)
                    self.maybe_print("OVERFLOW within closure! Skipping step. Attempted loss scale: {}, reducing to "
                                     "{}".format(bad, self.loss_scale))
                    retval = closure()
                self.first_closure_call_this_step = True
                scale = self.loss_scaler.loss_scale
            if not self._stats_valid:
                self._compute_stats()
            groups = self.optimizer.param_groups
            beta1, beta2 = groups[0]['betas']
            self._step_count += 1
            ops.adamw_step(self._arena.data, self._arena.grad, self._master_flat, self._m_flat, self._v_flat,
                           self._tables[0], self._tables[1], self._tables[2],
                           [g['lr'] for g in groups], [g['weight_decay'] for g in groups], beta1, beta2,
                           groups[0]['eps'], self._step_count, inv_loss_scale=1.0 / scale,
                           max_grad_norm=self._clip, stats=self._stats,
                           bias_correction=bool(groups[0].get('bias_correction', True)),
                           adam_w_mode=bool(getattr(self.optimizer, 'adam_w_mode', 1)))
            self._stats_valid = False
            if self._shard is not None:
                self._shard.gather_params()
            return retval
        retval = self._step_with_closure(closure) if closure is not None else self.optimizer.step()
        self._master_params_to_model_params()
        return retval

    def _step_with_closure(self, closure):
        def wrapped_closure():
            if self.first_closure_call_this_step:
                self.first_closure_call_this_step = False
            else:
                self._master_params_to_model_params()
            temp_loss = closure()
            while self.overflow:
                scale = self.loss_scaler.loss_scale
                self._update_scale(self.overflow)
                self.maybe_print("OVERFLOW within closure! Skipping step. Attempted loss scale: {}, reducing to "
                                 "{}".format(scale, self.loss_scale))
                temp_loss = closure()
            return temp_loss
        retval = self.optimizer.step(wrapped_closure)
        self.first_closure_call_this_step = True
        return retval

    def backward(self, loss, update_master_grads=True, retain_graph=False):
        self.loss_scaler.backward(loss.float(), retain_graph=retain_graph)
        if update_master_grads:
            self.update_master_grads()

    def update_master_grads(self):
        """fp16/fp16.py:556-567.  Fused path: nothing is copied -- the statistics pass runs here and the
        unscale happens inside the Adam kernel."""
        if self._ddp is not None:
            self._ddp.allreduce_params(reduce_after=False)
        if self._arena is not None:
            self._stats_valid = False
            self._clip = 0.0
            if self.dynamic_loss_scale:
                self._check_overflow()
            else:
                self.overflow = False
            return
        if self.dynamic_loss_scale:
            self._check_overflow()
            if self.overflow:
                return
        self._model_grads_to_master_grads()
        self._downscale_master()

    def inspect_master_grad_data(self):
        if self.overflow:
            print("Warning:  calling FP16_Optimizer.inspect_master_grad_data while in an overflow state.  "
                  "Gradients are currently invalid (may be inf, nan, or stale).  Returning None.")
            return None
        if self._arena is not None:
            inv = 1.0 / self.loss_scale
            return [[p.grad.float() * inv if p.grad is not None else None for p in group] for group in self.fp16_groups]
        return [[p.grad.data if p.grad is not None else None for p in g['params']] for g in self.optimizer.param_groups]

    def _get_loss_scale(self):
        return self.loss_scaler.loss_scale

    def _set_loss_scale(self, value):
        self.loss_scaler.cur_scale = value

    loss_scale = property(_get_loss_scale, _set_loss_scale)

    def _get_state(self):
        return self.optimizer.state

    def _set_state(self, value):
        self.optimizer.state = value

    state = property(_get_state, _set_state)

    def _get_param_groups(self):
        return self.optimizer.param_groups

    def _set_param_groups(self, value):
        self.optimizer.param_groups = value

    param_groups = property(_get_param_groups, _set_param_groups)
