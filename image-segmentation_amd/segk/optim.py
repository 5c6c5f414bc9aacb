"""Adam on the GPU in one launch (torch.optim.Adam semantics; the reference builds Adam at
perlosstrain/naive_train.py:397-398 and pl_utils/pl_mod.py:109-110)."""
import torch

from . import ops
from .flat import FlatModel


class Adam:
    def __init__(self, model_or_flat, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat = model_or_flat if isinstance(model_or_flat, FlatModel) else FlatModel(model_or_flat)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(self.flat.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat.flat_param)
        self.steps = 0

    def zero_grad(self, set_to_none=False):
        self.flat.flat_grad.zero_()

    def step(self):
        self.steps += 1
        ops.adam_step(self.flat.flat_param, self.flat.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr, self.betas[0],
                      self.betas[1], self.eps, self.weight_decay, self.steps)
