"""Data-parallel gradient exchange: bucketed asynchronous all-reduce over RCCL (xGMI), overlapped
with the rest of backward.

The reference is single-process (no torch.distributed anywhere), so this is new
work defined by BASELINE.json's north_star: one process per GPU, the batch sharded over ranks,
student gradients summed with `all_reduce` and divided by world size inside the fused AdamW kernel.
Semantics == k independent reference replicas on k micro-batches with averaged gradients
(per-rank BatchNorm statistics; the reference has no SyncBN).

Gradients live in ONE flat buffer (optim.FlatParams); a bucket is a contiguous slice of it.  The
payload is ~2.1 MB fp32 -- latency-bound on xGMI -- so there are only a few buckets, cut at
module boundaries in backward-completion order (head -> fusion -> FPN/LiDAR -> camera stages), and
each is launched from an autograd hook the moment its last gradient has been accumulated, so the
reduce of the late layers flies under the camera encoder's backward.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from .optim import FlatParams


def broadcast_module(module: torch.nn.Module, src: int = 0, group=None):
    """One-time parameter + buffer broadcast from rank `src` (like DDP's init)."""
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)
    from . import ops
    ops.bump_global_epoch()                # `.data` writes do not bump tensor versions: invalidate content-keyed caches


def broadcast_buffers(module: torch.nn.Module, src: int = 0, group=None):
    """BatchNorm running statistics are per replica during training (the reference has no SyncBN); before a validation
    pass every rank takes rank `src`'s buffers so that all ranks evaluate -- and rank 0 saves -- the same model."""
    for t in module.buffers():
        dist.broadcast(t.data, src=src, group=group)
    from . import ops
    ops.bump_global_epoch()


def distributed() -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class BucketedAllReduce:
    """`force=True` (or KD_FORCE_REDUCER=1): issue the collectives even in a world of ONE rank.  A one-rank sum is the
    identity, so the step must stay bit-identical to the reducer-less step -- which is how the whole RCCL path
    (communicator, ProcessGroupNCCL's side stream and events, async work handles, `wait()` ordering before AdamW) is
    executed and checked on a 1-GPU box (tests/test_gpu_rccl_world1.py, `bench.py --force-reducer`).
    `enabled=False` silences the hooks (a second, reducer-less KDStep over the same parameters)."""

    def __init__(self, flat: FlatParams, names: Optional[Sequence[str]] = None, n_buckets: int = 3, group=None,
                 force: Optional[bool] = None):
        self.flat, self.group = flat, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.force = (os.environ.get("KD_FORCE_REDUCER") == "1") if force is None else bool(force)
        if self.force and not dist.is_initialized():
            raise RuntimeError("BucketedAllReduce(force=True) needs an initialised process group (a world of one rank is fine)")
        self.enabled = True
        self.collectives_issued = 0       # all-reduce calls handed to torch.distributed so far (tests / bench read it)
        # set to a list to measure the EXPOSED all-reduce time: finish() then records a HIP-event pair on the compute stream
        # around its waits (the stream sits between them exactly as long as a reduction is still running); bench.py's `comm`
        self.exposed: Optional[list] = None
        n = len(flat.params)
        # bucket boundaries (parameter indices), contiguous in registration order; cut where the
        # top-level module name changes, then merged down to n_buckets of similar byte size
        cuts = [0]
        if names is not None:
            tops = [nm.split(".")[0] for nm in names]
            cuts += [i for i in range(1, n) if tops[i] != tops[i - 1]]
        cuts.append(n)
        spans = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1) if cuts[i + 1] > cuts[i]]
        while len(spans) > n_buckets:                   # merge the smallest neighbouring pair
            sizes = [flat.offsets[b] - flat.offsets[a] for a, b in spans]
            j = min(range(len(spans) - 1), key=lambda i: sizes[i] + sizes[i + 1])
            spans[j:j + 2] = [(spans[j][0], spans[j + 1][1])]
        self.spans = spans
        self.bucket_of = [0] * n
        for b, (a, e) in enumerate(spans):
            for i in range(a, e):
                self.bucket_of[i] = b
        self.views = [flat.grad[flat.offsets[a]:flat.offsets[e]] for a, e in spans]
        self.pending = [0] * len(spans)
        self.handles: List = []
        self.index_of = {id(p): i for i, p in enumerate(flat.params)}
        for i, p in enumerate(flat.params):
            p.register_post_accumulate_grad_hook(self._make_hook(i))
        self.launch_order: List[int] = []
        self.reset()

    def reset(self):
        for b, (a, e) in enumerate(self.spans):
            self.pending[b] = e - a
        self.seen = [False] * len(self.flat.params)
        self.handles = []
        self.launch_order = []

    def _make_hook(self, i):
        # A parameter counts ONCE per step, whichever path reports it first.  With the direct gradient sink the backward
        # Function reports the write itself (GradSink.done -> notify) and returns None to autograd -- and autograd still
        # runs the parameter's AccumulateGrad node afterwards and fires this hook a second time (observed on torch 2.10:
        # every bucket was launched when HALF of its gradients had landed, and the ranks diverged).
        def hook(_p):
            if not self.enabled or self.seen[i]:
                return
            self.seen[i] = True
            b = self.bucket_of[i]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def notify(self, p):
        """Called by the direct gradient sink (kdrt.gradsink) when p's gradient has been written."""
        self._make_hook(self.index_of[id(p)])(p)

    def _launch(self, b):
        self.launch_order.append(b)
        if self.world == 1 and not self.force:
            return
        # Canonical async collective: ProcessGroupNCCL (RCCL) runs it on its own internal stream, ordered
        # after everything already enqueued on the current (compute) stream -- i.e. after this bucket's
        # gradient kernels -- and `work.wait()` later makes the compute stream wait for it.  The launch
        # order is the (deterministic) order in which backward finishes the buckets, identical on all ranks.
        h = dist.all_reduce(self.views[b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.collectives_issued += 1
        self.handles.append(h)

    def finish(self):
        """Call after backward: launches any bucket whose hooks did not all fire (unused parameters)
        and makes the compute stream wait for the reductions (no host synchronisation with RCCL)."""
        for b in range(len(self.spans)):
            if self.pending[b] > 0:
                self._launch(b)
        if self.exposed is not None and self.handles and torch.cuda.is_available():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for h in self.handles:
                h.wait()
            e1.record()
            self.exposed.append((e0, e1))
        else:
            for h in self.handles:
                h.wait()
        self.reset()
        return 1.0 / self.world           # the factor the optimiser applies to the summed gradients
