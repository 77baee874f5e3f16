"""The knowledge-distillation training step (teacher forward, student forward, CE + T-softmax KL +
feature MSE, student backward, gradient all-reduce, fused AdamW) as one callable.

The reference trains one model with plain CE (trainer.py:86-90) and has no KD code; this step is
what BASELINE.json's north_star adds on top of the reference's `return_intermediates` affordance
(fusion_module.py:234,260-262).  Defaults: T=4, alpha=beta=1."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import gradsink, ops, units
from .ddp import BucketedAllReduce
from .losses import kd_objective, kd_objective_backward
from .optim import FusedAdamW


KD_FEATURES = ("camera_feat", "lidar_feat", "logits")       # what the objective reads of each model's intermediates


class KDStep:
    def __init__(self, student, teacher, optimizer: FusedAdamW, class_weights: Optional[torch.Tensor] = None,
                 T: float = 4.0, alpha: float = 1.0, beta: float = 1.0, ignore_index: int = -1,
                 reducer: Optional[BucketedAllReduce] = None, teacher_storage: str = "fp32",
                 fused_objective: Optional[bool] = None):
        if teacher_storage not in ("fp32", "bf16"):
            raise ValueError(f"teacher_storage must be 'fp32' or 'bf16', got {teacher_storage!r}")
        # fused objective (default): loss values and loss gradients from the same kernel passes, feature-MSE gradients added
        # inside the fusion block's data-gradient GEMMs (losses.kd_objective_backward); False / KD_FUSED_OBJECTIVE=0 keeps the
        # autograd formulation kd_objective(...).backward() -- same bits, more passes (tests compare the two)
        self.fused_objective = (os.environ.get("KD_FUSED_OBJECTIVE", "1") != "0") if fused_objective is None else bool(fused_objective)
        # "bf16": the frozen teacher runs kdrt.bf16.forward_bf16 (bf16 activations in HBM, fp32 accumulate); a second,
        # separately gated mode -- the default keeps every tensor of the step fp32
        self.teacher_storage = teacher_storage
        self.student, self.teacher, self.opt = student, teacher, optimizer
        self.cw, self.T, self.alpha, self.beta, self.ignore_index = class_weights, T, alpha, beta, ignore_index
        self.reducer = reducer
        self.sink = gradsink.install(optimizer.flat, reducer)      # backward kernels write into the flat grad buffer
        self.teacher.eval()
        for p in self.teacher.parameters():
            p.requires_grad_(False)

    def teacher_forward(self, images, points):
        with torch.no_grad():
            if self.teacher_storage == "bf16":
                from .bf16 import forward_bf16
                return forward_bf16(self.teacher, images, points, return_intermediates=KD_FEATURES)
            return self.teacher(images, points, return_intermediates=KD_FEATURES)

    def objective_backward(self, zs, ms, zt, mt, labels):
        """Loss values + the student's backward pass -> (total, parts of detached device scalars)."""
        a = (zs, ms, zt, mt, labels, self.cw, self.T, self.alpha, self.beta, self.ignore_index)
        if self.fused_objective:
            return kd_objective_backward(*a)
        total, parts = kd_objective(*a)
        gradsink.drop_pending()
        total.backward()
        if gradsink.pending():
            gradsink.drop_pending()
            raise RuntimeError("a deposited feature gradient was not collected (kdrt.gradsink): set KD_GRAD_ROUTING=0")
        return total.detach(), parts

    def __call__(self, images, points, labels):
        units.share_point_bins(True)       # teacher and student of THIS step sort the same points once ...
        try:
            zt, mt = self.teacher_forward(images, points)
            gradsink.active = self.sink
            self.sink.begin_step()
            self.opt.zero_grad()
            zs, ms = self.student(images, points, return_intermediates=KD_FEATURES)
        finally:
            units.share_point_bins(False)  # ... and nothing of it outlives the two forward passes
        total, parts = self.objective_backward(zs, ms, zt, mt, labels)
        self.sink.end_step()
        self.opt.grad_scale = self.reducer.finish() if self.reducer is not None else 1.0
        self.opt.step()
        parts["total"] = total
        parts["logits"] = zs.detach()
        return parts


class GraphedKDStep:
    """The whole KD step captured ONCE into a hipGraph (through torch.cuda.CUDAGraph) and replayed:
    ~600 kernel launches become one graph launch, so small per-GPU batches (the reference trains at
    B=4) are no longer bound by the ~5.8 ms of host launch work per step.  Everything that varies
    between steps lives on the device: the batch is copied into static input buffers, the AdamW
    step counter / bias corrections / learning rate are device state (kd_adamw_step_dev), BatchNorm
    running statistics are updated by the kernels themselves.

    With a gradient reducer (data parallel) the bucketed all-reduces are captured INSIDE the graph: ProcessGroupNCCL
    forks its collective stream off the capturing stream with an event and `work.wait()` joins it back, so the replayed
    graph holds backward kernels -> RCCL all-reduce per bucket -> AdamW with the same dependencies as the eager step.
    The capture then runs in "thread_local" error mode (the process group's watchdog thread polls events of earlier
    collectives from another thread, which the default "global" mode would treat as a capture violation).

    Restrictions: fixed batch shape; `optimizer.sync_lr()` happens automatically before each replay; more than one rank is
    an explicit opt-in (see __init__)."""

    def __init__(self, step: KDStep, images, points, labels, warmup: int = 3, allow_multi_rank: bool = False):
        # Captured collectives have only ever executed in a world of ONE rank here (tests/test_gpu_rccl_world1.py: a real RCCL
        # communicator, forced buckets); a multi-rank capture or replay problem would show as a hang in user training, so a
        # world of more than one rank is refused unless the caller opts in (`allow_multi_rank=True` / KD_GRAPH_MULTI_RANK=1)
        # -- to be lifted once a multi-GPU run has executed the captured path.
        world = step.reducer.world if step.reducer is not None else 1
        if world > 1 and not (allow_multi_rank or os.environ.get("KD_GRAPH_MULTI_RANK") == "1"):
            raise RuntimeError(f"GraphedKDStep: capturing the gradient all-reduces of a {world}-rank job has not run on hardware "
                               "yet; pass allow_multi_rank=True (or KD_GRAPH_MULTI_RANK=1) to try it, or use the eager KDStep")
        self.step = step
        self.images, self.points, self.labels = images.clone(), points.clone(), labels.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                       # warm-up on a side stream (allocator, workspaces, caches)
            for _ in range(warmup):
                self.out = step(self.images, self.points, self.labels)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        step.opt.sync_lr()
        mode = {} if step.reducer is None else {"capture_error_mode": "thread_local"}
        with torch.cuda.graph(self.graph, **mode):
            self.out = self._body()
        self.replays = 0

    def _body(self):
        s = self.step
        zt, mt = s.teacher_forward(self.images, self.points)
        gradsink.active = s.sink
        s.sink.begin_step()
        s.opt.zero_grad()
        zs, ms = s.student(self.images, self.points, return_intermediates=KD_FEATURES)
        total, parts = s.objective_backward(zs, ms, zt, mt, self.labels)
        s.sink.end_step()
        s.opt.grad_scale = s.reducer.finish() if s.reducer is not None else 1.0
        s.opt.enqueue_update()
        parts["total"] = total
        parts["logits"] = zs.detach()
        return parts

    def __call__(self, images=None, points=None, labels=None):
        if images is not None:
            self.images.copy_(images, non_blocking=True)
            self.points.copy_(points, non_blocking=True)
            self.labels.copy_(labels, non_blocking=True)
        self.step.opt.sync_lr()
        self.graph.replay()
        ops.bump_global_epoch()            # the replay rewrote parameters and BatchNorm buffers behind torch's back
        self.step.opt.note_steps(1)
        self.replays += 1
        return self.out
