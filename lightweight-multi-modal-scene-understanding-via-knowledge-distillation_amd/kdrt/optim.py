"""Fused AdamW over one flat parameter buffer (torch.optim.AdamW math, trainer.py:56).

All parameters are re-homed as views of one contiguous fp32 buffer, gradients likewise, so the
optimiser step is ONE kernel launch and the DDP all-reduce works on contiguous bucket slices.
`state_dict()` / `load_state_dict()` keep torch.optim.AdamW's layout (per-parameter `step`,
`exp_avg`, `exp_avg_sq`) so reference checkpoints (`optimizer_state`, trainer.py:116-142) load.
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from .lib import lib
from .ops import P, stream


class FlatParams:
    """Re-homes `params` (in the given order) into one flat buffer; p.data and p.grad become views."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        self.offsets = [0]
        for s in sizes:
            self.offsets.append(self.offsets[-1] + ((s + 3) // 4) * 4)      # keep 16-byte alignment per tensor
        self.numel = self.offsets[-1]
        self.data = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.data[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.data[o:o + n].view(p.shape)
            p.grad = self.grad[o:o + n].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):      # re-attach views if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.flat = FlatParams(params)
        self.exp_avg = torch.zeros_like(self.flat.data)
        self.exp_avg_sq = torch.zeros_like(self.flat.data)
        self._step = 0
        self.epoch = 0                   # bumped on every device update: caches keyed on parameter contents read it
        for q in self.flat.params:       # (ops.owner_epoch -- the update goes through raw pointers, not tensor versions)
            q._kd_owner = self
        self.grad_scale = 1.0            # set to 1/world_size by the DDP wrapper (sum all-reduce)
        # step-to-step state lives on the device so a captured hipGraph of the step replays correctly:
        # dev_state = [lr, step, 1-beta1^step, sqrt(1-beta2^step)]   (kd_adamw_step_dev)
        self.dev_state = torch.zeros(4, device=self.flat.data.device, dtype=torch.float32)
        self._dev_lr = None
        for p, o in zip(self.flat.params, self.flat.offsets):
            n = p.numel()
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": self.exp_avg[o:o + n].view(p.shape),
                             "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape)}

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()

    def sync_lr(self):
        """Push the current learning rate to the device state (call after a scheduler step; cheap no-op otherwise)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._dev_lr:
            self.dev_state[0:1].fill_(lr)
            self._dev_lr = lr

    def enqueue_update(self):
        """The device part of a step (two kernel launches, graph-capturable)."""
        self.epoch += 1
        g = self.param_groups[0]
        lib.call("kd_adamw_step_dev", P(self.flat.data), P(self.flat.grad), P(self.exp_avg), P(self.exp_avg_sq),
                 self.flat.numel, P(self.dev_state), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                 float(g["weight_decay"]), float(self.grad_scale), stream())

    def note_steps(self, k: int = 1):
        """Host-side bookkeeping for k device steps (state_dict compatibility with torch.optim.AdamW)."""
        self._step += k
        self.epoch += 1
        t = torch.tensor(float(self._step))
        for st in self.state.values():
            st["step"] = t

    @torch.no_grad()
    def step(self, closure=None):
        self.sync_lr()
        self.enqueue_update()
        self.note_steps(1)

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # re-home the loaded moments into the flat buffers.  torch.optim.AdamW creates per-parameter state lazily, so a
        # reference checkpoint may lack it for a parameter that never received a gradient: zeros / step 0 then.
        steps = []
        for p, o in zip(self.flat.params, self.flat.offsets):
            st = self.state[p]
            n = p.numel()
            if "exp_avg" not in st or "exp_avg_sq" not in st:
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                st.setdefault("step", torch.tensor(0.0))
            self.exp_avg[o:o + n].copy_(st["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            st["exp_avg"] = self.exp_avg[o:o + n].view(p.shape)
            st["exp_avg_sq"] = self.exp_avg_sq[o:o + n].view(p.shape)
            steps.append(int(float(st["step"])))
        self._step = max(steps) if steps else 0
        self.epoch += 1
        self.dev_state[1:2].fill_(float(self._step))
        self._dev_lr = None
