"""PolyLR with the reference's semantics (utils/scheduler.py:3-11): lr = max(base*(1-it/max)^p, min_lr),
stepped once per iteration AFTER optimizer.step() (main_embedding.py:507)."""
from torch.optim.lr_scheduler import _LRScheduler


class PolyLR(_LRScheduler):
    def __init__(self, optimizer, max_iters, power=0.9, last_epoch=-1, min_lr=1e-6):
        self.power, self.max_iters, self.min_lr = power, max_iters, min_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        frac = 1 - self.last_epoch / self.max_iters
        return [max(base * frac ** self.power, self.min_lr) for base in self.base_lrs]
