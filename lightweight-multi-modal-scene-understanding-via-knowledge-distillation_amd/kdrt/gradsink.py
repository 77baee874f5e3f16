"""Direct gradient sink: backward kernels write parameter gradients straight into the optimiser's
flat gradient buffer instead of returning fresh tensors for autograd to `+=` into `p.grad`
(96 tiny add kernels + copies per step otherwise).

Contract while a sink is installed: every registered parameter receives at most ONE gradient per
step (no weight sharing, no multi-backward accumulation) -- a second write in the same step raises.
Parameters the sink does not know go through autograd as usual."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

active: Optional["GradSink"] = None


class GradSink:
    def __init__(self, flat, reducer=None):
        self.flat, self.reducer = flat, reducer
        self.views: Dict[int, Tuple[torch.nn.Parameter, torch.Tensor]] = {}
        for p, o in zip(flat.params, flat.offsets):
            self.views[id(p)] = (p, flat.grad[o:o + p.numel()].view(p.shape))
        self.written = set()

    def begin_step(self):
        self.written.clear()

    def buffer(self, p) -> Optional[torch.Tensor]:
        e = self.views.get(id(p))
        if e is None or not p.requires_grad:
            return None
        if id(p) in self.written:
            raise RuntimeError("GradSink: a parameter received two gradients in one step (weight sharing / "
                               "gradient accumulation is not supported on the direct path)")
        return e[1]

    def done(self, p):
        self.written.add(id(p))
        if self.reducer is not None:
            self.reducer.notify(p)


def install(flat, reducer=None) -> GradSink:
    global active
    active = GradSink(flat, reducer)
    return active


def uninstall():
    global active
    active = None


def out_for(p: torch.Tensor):
    """-> (tensor to write the gradient of `p` into, direct?)"""
    if active is not None:
        b = active.buffer(p)
        if b is not None:
            return b, True
    return torch.empty_like(p), False


def finish(p: torch.Tensor, buf: torch.Tensor, direct: bool):
    """What the autograd Function must return for this parameter."""
    if direct:
        active.done(p)
        return None
    return buf


def deliver(p: torch.Tensor, value: torch.Tensor):
    """For gradients produced inside a packed buffer: copy into the sink (tiny) or hand to autograd."""
    if active is not None:
        b = active.buffer(p)
        if b is not None:
            b.copy_(value.reshape(b.shape))
            active.done(p)
            return None
    return value.reshape(p.shape)
