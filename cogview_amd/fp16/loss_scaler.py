"""Loss scalers with the state machine of fp16/loss_scaler.py:26-183 (bit-identical scale trajectories;
tests/test_host_logic.py replays the reference's own trajectories).

Overflow detection differs in mechanism, not in meaning: the reference sums every gradient tensor on the
device and pulls each sum to the host (one sync per tensor); here a single kernel over the flat gradient
arena sets a device flag (cogv_grad_stats) that is read once."""
import torch

from .. import mpu
from .. import ops
from ..mpu.grads import _chunk_tables


def to_python_float(t):
    return t.item() if hasattr(t, 'item') else t[0]


def _device_overflow_flag(params):
    """1.0 if any gradient among `params` holds inf/nan (device tensor [2] float64: sumsq, flag)."""
    params = [p for p in params if p.grad is not None]
    if not params:
        return None
    dev = params[0].grad.device
    stats = torch.zeros(2, dtype=torch.float64, device=dev)
    # one cogv_grad_stats launch per underlying storage (ONE for gradients that live in a flat arena), 16-bit and fp32 alike
    for flat, cs, cl, cn in _chunk_tables([p.grad.data for p in params]):
        ops.grad_stats(flat, cs, cl, cn, stats)
    return stats


class LossScaler:
    """Static loss scale (fp16/loss_scaler.py:26-60)."""

    def __init__(self, scale=1):
        self.cur_scale = scale

    def has_overflow(self, params):
        return False

    def _has_inf_or_nan(x):
        return False

    def update_scale(self, overflow):
        pass

    @property
    def loss_scale(self):
        return self.cur_scale

    def scale_gradient(self, module, grad_in, grad_out):
        return tuple(self.loss_scale * g for g in grad_in)

    def backward(self, loss, retain_graph=False):
        (loss * self.loss_scale).backward(retain_graph=retain_graph)


class DynamicLossScaler:
    """fp16/loss_scaler.py:63-183: start at init_scale (2**32), on overflow divide by scale_factor (after
    `delayed_shift` hysteresis), multiply by scale_factor after every `scale_window` clean iterations,
    never below min_scale."""

    def __init__(self, init_scale=2 ** 32, scale_factor=2., scale_window=1000, min_scale=1, delayed_shift=1,
                 consecutive_hysteresis=False):
        self.cur_scale = init_scale
        self.cur_iter = 0
        self.last_overflow_iter = -1
        self.scale_factor = scale_factor
        self.scale_window = scale_window
        self.min_scale = min_scale
        self.delayed_shift = delayed_shift
        self.cur_hysteresis = delayed_shift
        self.consecutive_hysteresis = consecutive_hysteresis

    def has_overflow_serial(self, params):
        stats = _device_overflow_flag(params)
        return bool(stats is not None and stats[1].item() != 0.0)

    def has_overflow(self, params):
        overflow = self.has_overflow_serial(params)
        return self.sync_overflow(overflow, params)

    @staticmethod
    def sync_overflow(overflow, params=None):
        """MAX over the model-parallel group (fp16/loss_scaler.py:115-122): each rank holds a model shard."""
        if mpu.model_parallel_is_initialized() and mpu.get_model_parallel_world_size() > 1:
            dev = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else 'cpu'
            flag = torch.tensor([1.0 if overflow else 0.0], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX, group=mpu.get_model_parallel_group())
            overflow = bool(flag.item())
        return bool(overflow)

    @staticmethod
    def _has_inf_or_nan(x):
        s = float(x.float().sum())
        return s == float('inf') or s == -float('inf') or s != s

    def update_scale(self, overflow):
        if overflow:
            if self.delayed_shift == 1 or self.cur_hysteresis == 1:
                self.cur_scale = max(self.cur_scale / self.scale_factor, self.min_scale)
            else:
                self.cur_hysteresis -= 1
            self.last_overflow_iter = self.cur_iter
        else:
            if self.consecutive_hysteresis:
                self.cur_hysteresis = self.delayed_shift
            if (self.cur_iter - self.last_overflow_iter) % self.scale_window == 0:
                if not self.consecutive_hysteresis:
                    self.cur_hysteresis = self.delayed_shift
                self.cur_scale *= self.scale_factor
        self.cur_iter += 1

    @property
    def loss_scale(self):
        return self.cur_scale

    def scale_gradient(self, module, grad_in, grad_out):
        return tuple(self.loss_scale * g for g in grad_in)

    def backward(self, loss, retain_graph=False):
        (loss * self.loss_scale).backward(retain_graph=retain_graph)
