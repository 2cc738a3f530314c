"""Learning-rate schedule of the training script (learning_rates.py:21-83): linear warm-up to `start_lr`, then a linear or
cosine decay over `num_iters`; the cosine ends at start_lr * decay_ratio (the constructor stores 1 / decay_ratio, as the
reference does, and that is what travels in checkpoints).  Works on any object with `.param_groups` -- torch optimizers,
cogview_amd.optim.FusedAdam, FP16_Optimizer."""
import math


class AnnealingLR:
    DECAY_STYLES = ['linear', 'cosine', 'exponential', 'constant', 'None']

    def __init__(self, optimizer, start_lr, warmup_iter, num_iters, decay_style=None, last_iter=-1, decay_ratio=0.5):
        assert warmup_iter <= num_iters
        self.optimizer, self.start_lr, self.warmup_iter = optimizer, start_lr, warmup_iter
        self.num_iters, self.end_iter = last_iter + 1, num_iters
        self.decay_style = decay_style.lower() if isinstance(decay_style, str) else None
        self.decay_ratio = 1 / decay_ratio
        self.step(self.num_iters)

    def get_lr(self):
        n = self.num_iters
        if self.warmup_iter > 0 and n <= self.warmup_iter:
            return float(self.start_lr) * n / self.warmup_iter
        if self.decay_style == 'linear':
            return self.start_lr * ((self.end_iter - (n - self.warmup_iter)) / self.end_iter)
        if self.decay_style == 'cosine':
            frac = min(1.0, (n - self.warmup_iter) / self.end_iter)
            return self.start_lr / self.decay_ratio * ((math.cos(math.pi * frac) + 1) * (self.decay_ratio - 1) / 2 + 1)
        return self.start_lr                 # 'exponential' is a stub in the reference too; constant / None

    def step(self, step_num=None):
        self.num_iters = self.num_iters + 1 if step_num is None else step_num
        lr = self.get_lr()
        for group in self.optimizer.param_groups:
            group['lr'] = lr

    def state_dict(self):
        return {'warmup_iter': self.warmup_iter, 'num_iters': self.num_iters, 'decay_style': self.decay_style,
                'end_iter': self.end_iter, 'decay_ratio': self.decay_ratio}

    def load_state_dict(self, sd):
        self.warmup_iter, self.num_iters, self.decay_style = sd['warmup_iter'], sd['num_iters'], sd['decay_style']
        if 'decay_ratio' in sd:
            self.decay_ratio = sd['decay_ratio']
        self.step(self.num_iters)
