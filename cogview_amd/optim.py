"""FusedAdam -- stand-in for apex.optimizers.FusedAdam (reference call site pretrain_gpt2.py:43,139-140):
Adam with decoupled weight decay (adam_w_mode=True), bias correction, betas (0.9, 0.999), eps 1e-8.

Inside FP16_Optimizer with arena-backed parameters the whole update is ONE HIP kernel over flat buffers
(unscale + clip + AdamW + cast to the 16-bit model parameters: cogv_adamw_step).  Stand-alone use on
arbitrary fp32 parameters takes the same update rule through torch tensor ops (API completeness, not the
hot path)."""
import torch


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, bias_correction=True, betas=(0.9, 0.999), eps=1e-8, adam_w_mode=True,
                 weight_decay=0., amsgrad=False, set_grad_none=True):
        if amsgrad:
            raise RuntimeError('FusedAdam does not support the AMSGrad variant.')
        defaults = dict(lr=lr, bias_correction=bias_correction, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.adam_w_mode = 1 if adam_w_mode else 0
        self.set_grad_none = set_grad_none

    def _arenas(self):
        out = {}
        for group in self.param_groups:
            for p in group['params']:
                e = getattr(p, '_cogv_arena', None)
                if e is not None:
                    out[id(e[0])] = e[0]
        return list(out.values())

    @property
    def lazy_zero_grad_ok(self):
        return all(getattr(p, '_cogv_arena', None) is not None and p._cogv_arena[0].lazy_ok
                   for group in self.param_groups for p in group['params'])

    def finish_lazy_zero_grad(self):
        for a in self._arenas():
            a.finish_lazy()

    def zero_grad(self, set_to_none=None, lazy=False):
        """Arena-backed parameters keep `param.grad` as a view of the flat gradient buffer (one memset): the
        data-parallel exchange reduces slices of that buffer, so dropping the views would silently stop the gradient
        synchronisation.  Loose parameters follow apex's set_grad_none default."""
        if set_to_none is None:
            set_to_none = self.set_grad_none
        arenas, loose = {}, False
        for group in self.param_groups:
            for p in group['params']:
                e = getattr(p, '_cogv_arena', None)
                if e is not None:
                    arenas[id(e[0])] = e[0]
                else:
                    loose = True
        for a in arenas.values():
            a.zero_grad(lazy=lazy and not loose)
        if not loose:
            return
        for group in self.param_groups:
            for p in group['params']:
                if getattr(p, '_cogv_arena', None) is not None or p.grad is None:
                    continue
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.detach_()
                    p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.finish_lazy_zero_grad()
        for group in self.param_groups:
            beta1, beta2 = group['betas']
            group['step'] = group.get('step', 0) + 1
            step = group['step']
            bc1 = 1 - beta1 ** step if group['bias_correction'] else 1.0
            bc2 = 1 - beta2 ** step if group['bias_correction'] else 1.0
            for p in group['params']:
                if p.grad is None:
                    continue
                g = p.grad.float()
                st = self.state[p]
                if len(st) == 0:
                    st['exp_avg'] = torch.zeros_like(p, dtype=torch.float32)
                    st['exp_avg_sq'] = torch.zeros_like(p, dtype=torch.float32)
                m, v = st['exp_avg'], st['exp_avg_sq']
                if not self.adam_w_mode:
                    g = g.add(p.float(), alpha=group['weight_decay'])
                m.mul_(beta1).add_(g, alpha=1 - beta1)
                v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
                upd = (m / bc1) / ((v / bc2).sqrt() + group['eps'])
                if self.adam_w_mode:
                    upd.add_(p.float(), alpha=group['weight_decay'])
                p.add_(upd.to(p.dtype), alpha=-group['lr'])
        return loss
