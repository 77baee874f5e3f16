"""The knowledge-distillation training step (teacher forward, student forward, CE + T-softmax KL +
feature MSE, student backward, gradient all-reduce, fused AdamW) as one callable.

The reference trains one model with plain CE (trainer.py:86-90) and has no KD code; this step is
what BASELINE.json's north_star adds on top of the reference's `return_intermediates` affordance
(fusion_module.py:234,260-262).  Defaults: T=4, alpha=beta=1."""
from __future__ import annotations

from typing import Optional

import torch

from . import gradsink
from .ddp import BucketedAllReduce
from .losses import kd_objective
from .optim import FusedAdamW


class KDStep:
    def __init__(self, student, teacher, optimizer: FusedAdamW, class_weights: Optional[torch.Tensor] = None,
                 T: float = 4.0, alpha: float = 1.0, beta: float = 1.0, ignore_index: int = -1,
                 reducer: Optional[BucketedAllReduce] = None):
        self.student, self.teacher, self.opt = student, teacher, optimizer
        self.cw, self.T, self.alpha, self.beta, self.ignore_index = class_weights, T, alpha, beta, ignore_index
        self.reducer = reducer
        self.sink = gradsink.install(optimizer.flat, reducer)      # backward kernels write into the flat grad buffer
        self.teacher.eval()
        for p in self.teacher.parameters():
            p.requires_grad_(False)

    def __call__(self, images, points, labels):
        with torch.no_grad():
            zt, mt = self.teacher(images, points, return_intermediates=True)
        gradsink.active = self.sink
        self.sink.begin_step()
        self.opt.zero_grad()
        zs, ms = self.student(images, points, return_intermediates=True)
        total, parts = kd_objective(zs, ms, zt, mt, labels, self.cw, self.T, self.alpha, self.beta, self.ignore_index)
        total.backward()
        self.opt.grad_scale = self.reducer.finish() if self.reducer is not None else 1.0
        self.opt.step()
        parts["total"] = total.detach()
        parts["logits"] = zs.detach()
        return parts
