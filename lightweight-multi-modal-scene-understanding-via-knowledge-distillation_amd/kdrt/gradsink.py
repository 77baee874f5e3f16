"""Direct gradient sink: backward kernels write parameter gradients straight into the optimiser's
flat gradient buffer instead of returning fresh tensors for autograd to `+=` into `p.grad`
(96 tiny add kernels + copies per step otherwise).

Contract while a sink is installed: every registered parameter receives at most ONE gradient per
step (no weight sharing, no multi-backward accumulation) -- a second write in the same step raises.
Parameters the sink does not know go through autograd as usual."""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch

from . import ops

active: Optional["GradSink"] = None


class GradSink:
    def __init__(self, flat, reducer=None):
        self.flat, self.reducer = flat, reducer
        self.views: Dict[int, Tuple[torch.nn.Parameter, torch.Tensor]] = {}
        for p, o in zip(flat.params, flat.offsets):
            self.views[id(p)] = (p, flat.grad[o:o + p.numel()].view(p.shape))
        self.written = set()
        self.step_open = False         # True between begin_step() and end_step(): only then may feature gradients be routed

    def begin_step(self):
        """Start of a training step that will check `pending()` after its backward pass (KDStep, Trainer._step).  Feature-gradient
        routing (units.grad_routing) is enabled only while such a step is open: a plain `model(...); loss.backward()` loop that
        merely inherited an installed sink never deposits addends nobody checks."""
        self.written.clear()
        drop_pending()                 # nothing deposited by a step that failed half-way may leak into this one
        self.step_open = True
        ops.TRANSPOSES.refresh()       # the parameters are final for this step: every W^T the data gradients need, one launch

    def end_step(self):
        self.step_open = False

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


# Feature-map gradient addends.  A loss that reads an intermediate feature map (the KD step's two feature MSEs) deposits its
# gradient here, keyed by the map's [M, C] matrix; the fusion block that consumed the same map hands it to the data-gradient
# kernel of its projection as `addend`, so the sum of the two gradient paths is formed inside that kernel instead of by an
# autograd accumulation pass over the map.  The depositor must check `pending()` is empty after backward (KDStep does): an
# addend nobody collected would be a silently missing gradient term.
_addends: Dict[tuple, tuple] = {}


def _key(mat: torch.Tensor):
    # address AND shape: an address the caching allocator handed to an unrelated tensor after a failed step does not match by accident
    return (mat.data_ptr(), tuple(mat.shape))


def deposit(mat: torch.Tensor, grad: torch.Tensor, folded_residual: Optional[torch.Tensor] = None):
    """`folded_residual`: a gradient that is already part of `grad` and that the collecting residual block would otherwise add
    for its own skip connection (FPNFn folds d(stage-5 output) into its stage-4 deposit); the collector compares it with the
    gradient it actually received and corrects the sum if they differ (a third consumer of that output)."""
    if mat.shape != grad.shape or not grad.is_contiguous():
        raise RuntimeError("gradsink.deposit: the addend must be a contiguous tensor shaped like the feature matrix")
    _addends[_key(mat)] = (grad, folded_residual)


def collect(mat: torch.Tensor) -> Optional[torch.Tensor]:
    e = _addends.pop(_key(mat), None) if _addends else None
    return None if e is None else e[0]


def collect_flagged(mat: torch.Tensor):
    """-> (gradient, folded residual gradient or None) or None"""
    return _addends.pop(_key(mat), None) if _addends else None


def pending() -> int:
    return len(_addends)


def drop_pending():
    _addends.clear()


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


def deliver_many(pairs):
    """deliver() for up to four (parameter, packed gradient) pairs with ONE copy launch when all of them go to the sink;
    returns the list deliver() would have returned for each pair."""
    if active is not None and 1 < len(pairs) <= 4:
        bufs = [active.buffer(p) for p, _ in pairs]
        if all(b is not None for b in bufs):
            from .lib import lib
            a = []
            for b, (_, v) in zip(bufs, pairs):
                v = v.reshape(-1)
                if not (v.is_contiguous() and b.is_contiguous() and v.numel() == b.numel()):
                    break
                a += [ops.P(v), ops.P(b), v.numel()]
            else:
                a += [None, None, 0] * (4 - len(pairs))
                lib.call("kd_copy_segments", *a, ops.stream())
                for p, _ in pairs:
                    active.done(p)
                return [None] * len(pairs)
    return [deliver(p, v) for p, v in pairs]


def deliver(p: torch.Tensor, value: torch.Tensor):
    """For gradients produced inside a packed buffer: copy into the sink (tiny) or hand to autograd."""
    if active is not None:
        b = active.buffer(p)
        if b is not None:
            b.copy_(value.reshape(b.shape))
            active.done(p)
            return None
    return value.reshape(p.shape)
