"""Loss / metric Functions of the KD step (device-side values, no host sync).

  seg_loss      : weighted CE with ignore_index (trainer.py:55,88), optionally + alpha*T^2*KL to a
                  teacher's logits -- value and dL/dlogits produced by one fused HIP call
  feature_mse   : F.mse_loss forward + gradient
  kd_objective  : CE + alpha*T^2*KL + beta*(MSE(cam) + MSE(lidar))   (the
                  reference has no KD code -- this definition is the build's specification)
  kd_objective_backward : the same objective AND its backward pass for the training step: every loss kernel
                  produces its value and its gradient in one pass (the root's upstream gradient is 1), the
                  feature-MSE gradients ride into the fusion block's data-gradient GEMMs as addends
  confusion     : argmax + confusion matrix of SegmentationMetrics.update (trainer.py:18-26)
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

import numpy as np

from . import gradsink, ops
from .lib import KDError, lib
from .ops import P, stream


def _check_target(logits, target):
    """The loss kernels index the target as [B, H, W] of the logits: anything else would read past its end."""
    B, _, H, W = logits.shape
    if tuple(target.shape) != (B, H, W) or target.device != logits.device:
        raise KDError(f"segmentation target {tuple(target.shape)} on {target.device} does not match the logits' "
                      f"[B, H, W] = {(B, H, W)} on {logits.device}")


class _SegLossFn(torch.autograd.Function):
    """forward: loss values only; backward: one more fused call that writes dL/dlogits scaled by the
    upstream gradient read from device memory (no host sync, no extra elementwise pass)."""

    @staticmethod
    def _call(zs, zt, target, class_w, ignore_index, T, alpha, gdev, losses, dzs):
        B, NC, H, W = zs.shape
        nbytes = lib.kd_seg_loss_ws_bytes(B * H * W)
        ws = ops.workspace(nbytes, zs.device)
        lib.call("kd_seg_loss_fwd_bwd", P(zs), P(zt), P(target), P(class_w), int(ignore_index), float(T), float(alpha),
                 1.0, P(gdev), P(losses), P(dzs), B, NC, H * W, P(ws), nbytes, stream())

    @staticmethod
    def forward(ctx, zs, zt, target, class_w, ignore_index, T, alpha):
        ops.require_gpu_tensor(zs, "seg_loss")
        zs_c = zs.detach().contiguous()
        zt_c = None if zt is None else zt.detach().contiguous()
        target = target.contiguous()
        if target.dtype != torch.int64:
            raise KDError("segmentation target must be int64")
        _check_target(zs_c, target)
        losses = torch.empty(4, device=zs.device, dtype=torch.float32)
        _SegLossFn._call(zs_c, zt_c, target, class_w, ignore_index, T, alpha, None, losses, None)
        ctx.args = (zs_c, zt_c, target, class_w, ignore_index, T, alpha)
        kl = losses[1]
        ctx.mark_non_differentiable(kl)
        return losses[0], kl

    @staticmethod
    def backward(ctx, g_ce, _g_kl):
        # the gradient through `ce` is d(CE + alpha*T^2*KL)/dzs (kd_objective adds KL's value separately)
        zs_c, zt_c, target, class_w, ignore_index, T, alpha = ctx.args
        losses = torch.empty(4, device=zs_c.device, dtype=torch.float32)
        dzs = torch.empty_like(zs_c)
        g = g_ce.contiguous().view(1)
        _SegLossFn._call(zs_c, zt_c, target, class_w, ignore_index, T, alpha, g, losses, dzs)
        return dzs, None, None, None, None, None, None


def seg_loss(logits, target, class_weights: Optional[torch.Tensor] = None, ignore_index: int = -1,
             teacher_logits: Optional[torch.Tensor] = None, T: float = 4.0, alpha: float = 1.0):
    """-> (ce, kl).  The gradient that flows back through `ce` is that of ce + alpha*T^2*kl
    (kl is returned for logging / for adding its VALUE to the total)."""
    if teacher_logits is None:
        alpha = 0.0
    return _SegLossFn.apply(logits, teacher_logits, target, class_weights, ignore_index, T, alpha)


class _MSEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ops.require_gpu_tensor(a, "feature_mse")
        am, _ = ops.nhwc_view(a.detach())
        bm, _ = ops.nhwc_view(b.detach())
        n = am.numel()
        loss = torch.empty(1, device=a.device, dtype=torch.float32)
        nbytes = lib.kd_mse_ws_bytes(n)
        ws = ops.workspace(nbytes, a.device)
        lib.call("kd_mse_fwd_bwd", P(am), P(bm), n, 0.0, None, P(loss), None, P(ws), nbytes, stream())
        ctx.am, ctx.bm = am, bm
        ctx.geom = (a.shape[0], a.shape[2], a.shape[3])
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        am, bm = ctx.am, ctx.bm
        n = am.numel()
        da = torch.empty_like(am)
        nbytes = lib.kd_mse_ws_bytes(n)
        ws = ops.workspace(nbytes, am.device)
        lib.call("kd_mse_fwd_bwd", P(am), P(bm), n, 2.0 / n, P(g.contiguous().view(1)), None, P(da), P(ws), nbytes, stream())
        return ops.nchw_from_matrix(da, ctx.geom), None


def feature_mse(student_feat, teacher_feat):
    return _MSEFn.apply(student_feat, teacher_feat)


def kd_objective(student_logits, student_mids: Dict[str, torch.Tensor], teacher_logits, teacher_mids, target,
                 class_weights=None, T: float = 4.0, alpha: float = 1.0, beta: float = 1.0, ignore_index: int = -1):
    """total = CE + alpha*T^2*KL + beta*(MSE(camera_feat) + MSE(lidar_feat)); returns (total, parts)."""
    ce, kl = seg_loss(student_logits, target, class_weights, ignore_index, teacher_logits, T, alpha)
    mse_c = feature_mse(student_mids["camera_feat"], teacher_mids["camera_feat"])
    mse_l = feature_mse(student_mids["lidar_feat"], teacher_mids["lidar_feat"])
    # `ce` carries the CE+KL gradient; add KL's value without a second gradient path
    total = ce + (alpha * T * T) * kl.detach() + beta * (mse_c + mse_l)
    return total, {"ce": ce.detach(), "kl": kl.detach(), "mse_cam": mse_c.detach(), "mse_lidar": mse_l.detach()}


def kd_objective_backward(student_logits, student_mids: Dict[str, torch.Tensor], teacher_logits, teacher_mids, target,
                          class_weights=None, T: float = 4.0, alpha: float = 1.0, beta: float = 1.0, ignore_index: int = -1):
    """kd_objective(...)[0].backward() in one: returns (total, parts) with the student's gradients already propagated.

    Same values and the same gradient bits as the autograd formulation, fewer passes: the segmentation-loss call writes
    dL/dlogits together with CE / KL; each feature MSE writes its gradient in the pass that sums its value, and that
    gradient is deposited (gradsink.deposit) for the fusion block's projection of the same map, whose data-gradient GEMM
    adds it in its epilogue -- no gradient-accumulation pass over the [B, 128, H, W] maps, no scalar-arithmetic kernels."""
    zs = student_logits
    ops.require_gpu_tensor(zs, "kd_objective_backward")
    if not zs.requires_grad:
        raise KDError("kd_objective_backward: the student logits carry no autograd graph (was the forward run under no_grad?)")
    dev = zs.device
    zs_c = zs.detach().contiguous()
    zt_c = teacher_logits.detach().contiguous()
    target = target.contiguous()
    if target.dtype != torch.int64:
        raise KDError("segmentation target must be int64")
    _check_target(zs_c, target)
    B, NC, H, W = zs_c.shape
    vals = torch.empty(8, device=dev, dtype=torch.float32)        # [0:4] seg-loss values, [4] mse_cam, [5] mse_lidar, [6] total
    dzs = torch.empty_like(zs_c)
    nbytes = lib.kd_seg_loss_ws_bytes(B * H * W)
    ws = ops.workspace(nbytes, dev)
    lib.call("kd_seg_loss_fwd_bwd", P(zs_c), P(zt_c), P(target), P(class_weights), int(ignore_index), float(T), float(alpha),
             1.0, None, P(vals), P(dzs), B, NC, H * W, P(ws), nbytes, stream())
    roots, grads = [zs], [dzs]
    gradsink.drop_pending()
    slabs, counts = [], []
    for key in ("camera_feat", "lidar_feat"):
        a, b = student_mids[key], teacher_mids[key]
        am, _ = ops.nhwc_view(a.detach())
        bm, _ = ops.nhwc_view(b.detach())
        n = am.numel()
        want = beta != 0.0 and a.requires_grad
        da = torch.empty_like(am) if want else None
        slab = torch.empty(lib.kd_mse_slab_blocks(n), device=dev, dtype=torch.float32)
        gcoef = float(np.float32(2.0 / n) * np.float32(beta))      # the product autograd's fp32 chain rule forms
        lib.call("kd_mse_partial", P(am), P(bm), n, gcoef, P(da), P(slab), stream())
        slabs.append(slab)
        counts.append(n)
        if want:
            gradsink.deposit(am, da)
    # the two MSE values and the total in one launch (vals[4], vals[5], vals[6])
    lib.call("kd_kd_objective_final", P(vals), P(slabs[0]), counts[0], P(slabs[1]), counts[1], float(alpha * T * T), float(beta), P(vals[4:]),
             stream())
    torch.autograd.backward(roots, grads)
    if gradsink.pending():
        gradsink.drop_pending()
        raise KDError("kd_objective_backward: a feature-map gradient was not collected by the fusion block's backward "
                      "(unsupported model structure for the fused objective; use kd_objective(...).backward())")
    return vals[6], {"ce": vals[0], "kl": vals[1], "mse_cam": vals[4], "mse_lidar": vals[5]}


def confusion(logits, target, num_classes: int = 2, ignore_index: int = -1, out: Optional[torch.Tensor] = None):
    """Accumulates into (or creates) an int64 [C, C] device confusion matrix; returns (conf, argmax)."""
    ops.require_gpu_tensor(logits, "confusion")
    z = logits.detach().contiguous()
    B, NC, H, W = z.shape
    _check_target(z, target)
    if out is None:
        out = torch.zeros(num_classes, num_classes, device=z.device, dtype=torch.int64)
    pred = torch.empty(B, H, W, device=z.device, dtype=torch.int64)
    lib.call("kd_argmax_confusion", P(z), P(target.contiguous()), int(ignore_index), P(out), P(pred), B, NC, H * W, stream())
    return out, pred
