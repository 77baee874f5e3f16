"""CPU oracle for the camera+LiDAR KD training path -- TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) *restatement* of the algorithm the
reference implements in `src/models/*.py` and `src/training/trainer.py`.  It is
written functionally over a flat ``state`` dict (same keys as the reference's
``state_dict()``), not as nn.Modules, so it shares no code with the reference
or with the product package.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module.  The product path (the package
``lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd``)
never imports it and fails loudly when the HIP library is missing.

Pinning: ``oracle/make_golden.py`` imports the real reference (in the build
container only) and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks every function below against those vectors.  The KD loss (``kd_loss``)
has NO reference implementation: for that one
function parity is unpinned and the definition below *is* the specification.

Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]

BN_EPS = 1e-5       # nn.BatchNorm default (camera_encoder.py:25, lidar_encoder.py:27)
BN_MOMENTUM = 0.1   # nn.BatchNorm default


# --------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------
def _bn(x: torch.Tensor, st: State, key: str, training: bool) -> torch.Tensor:
    """BatchNorm (1d or 2d) with the nn.BatchNorm semantics the reference relies on:
    batch statistics + running-stat update (momentum 0.1, unbiased var) in training,
    running statistics in eval.  `num_batches_tracked` is incremented in training."""
    rm, rv = st[key + ".running_mean"], st[key + ".running_var"]
    if training:
        nbt = st.get(key + ".num_batches_tracked")
        if nbt is not None:
            nbt += 1
    return F.batch_norm(x, rm, rv, st[key + ".weight"], st[key + ".bias"],
                        training=training, momentum=BN_MOMENTUM, eps=BN_EPS)


def _act(x: torch.Tensor, kind: str) -> torch.Tensor:
    if kind == "relu6":
        return torch.clamp(x, 0.0, 6.0)      # nn.ReLU6 == hardtanh(0, 6)
    if kind == "relu":
        return torch.clamp_min(x, 0.0)
    if kind == "none":
        return x
    raise ValueError(kind)


def _pw_bn_act(x, st, conv, bn, act, training):
    """1x1 conv (+optional bias) -> BN -> act.  fusion_module.py:8-17 (Conv1x1),
    camera_encoder.py:22-27 / :38-41 (expand / project)."""
    y = F.conv2d(x, st[conv + ".weight"], st.get(conv + ".bias"))
    return _act(_bn(y, st, bn, training), act)


def _dw_bn_act(x, st, conv, bn, act, stride, training):
    """depthwise 3x3 pad 1 -> BN -> act.  camera_encoder.py:30-35, fusion_module.py:25-27."""
    w = st[conv + ".weight"]
    y = F.conv2d(x, w, None, stride=stride, padding=1, groups=w.shape[0])
    return _act(_bn(y, st, bn, training), act)


# --------------------------------------------------------------------------
# camera encoder   (src/models/camera_encoder.py)
# --------------------------------------------------------------------------
def inverted_residual(x, st: State, p: str, cin: int, cout: int, stride: int,
                      expansion: int, training: bool):
    """camera_encoder.py:9-51.  Sequential indices: with expansion != 1 the layout is
    0 conv,1 bn,2 act | 3 dw,4 bn,5 act | 6 conv,7 bn ; with expansion == 1 the first
    three are absent (0 dw,1 bn,2 act | 3 conv,4 bn)."""
    h = x
    i = 0
    if expansion != 1:
        h = _pw_bn_act(h, st, f"{p}.conv.0", f"{p}.conv.1", "relu6", training)
        i = 3
    h = _dw_bn_act(h, st, f"{p}.conv.{i}", f"{p}.conv.{i+1}", "relu6", stride, training)
    h = _pw_bn_act(h, st, f"{p}.conv.{i+3}", f"{p}.conv.{i+4}", "none", training)
    if stride == 1 and cin == cout:          # camera_encoder.py:14,48-49
        h = x + h
    return h


def twinlite_encoder(x, st: State, p: str = "", training: bool = False,
                     return_multiscale: bool = False, base: int = 32):
    """camera_encoder.py:56-115.  `p` is the key prefix ('' or 'camera_encoder.')."""
    y = F.conv2d(x, st[p + "stem.0.weight"], None, stride=2, padding=1)   # :63-67
    y = _act(_bn(y, st, p + "stem.1", training), "relu6")
    x1 = inverted_residual(y, st, p + "stage1", base, base, 1, 1, training)
    x2 = inverted_residual(x1, st, p + "stage2", base, 2 * base, 2, 6, training)
    x3 = inverted_residual(x2, st, p + "stage3", 2 * base, 2 * base, 1, 6, training)
    x4 = inverted_residual(x3, st, p + "stage4", 2 * base, 4 * base, 2, 6, training)
    x5 = inverted_residual(x4, st, p + "stage5", 4 * base, 4 * base, 1, 6, training)
    if return_multiscale:
        return {"stage2": x2, "stage3": x3, "stage4": x4, "stage5": x5}
    return x5


# --------------------------------------------------------------------------
# LiDAR encoder   (src/models/lidar_encoder.py)
# --------------------------------------------------------------------------
def points_to_bev(points, x_range=(-50.0, 50.0), y_range=(-50.0, 50.0)):
    """lidar_encoder.py:42-55: normalised coords and the inclusive validity mask."""
    xn = (points[..., 0] - x_range[0]) / (x_range[1] - x_range[0])
    yn = (points[..., 1] - y_range[0]) / (y_range[1] - y_range[0])
    valid = (xn >= 0) & (xn <= 1) & (yn >= 0) & (yn <= 1)
    return xn, yn, valid


def bev_cell_index(points, grid: Tuple[int, int], x_range=(-50.0, 50.0), y_range=(-50.0, 50.0)):
    """lidar_encoder.py:69-79: flat cell index b*H*W + iy*W + ix (int64) and valid mask.
    `.long()` truncates toward zero; clamp to [0, W-1] / [0, H-1]."""
    H, W = grid
    B, N = points.shape[:2]
    xn, yn, valid = points_to_bev(points, x_range, y_range)
    ix = (xn * float(W - 1)).to(torch.int64).clamp(0, W - 1)
    iy = (yn * float(H - 1)).to(torch.int64).clamp(0, H - 1)
    b = torch.arange(B).view(B, 1).expand(B, N)
    return b * (H * W) + iy * W + ix, valid


def binning_stable_points(points, grid: Tuple[int, int]):
    """Test-input helper (no reference counterpart): zero the few points whose BEV cell or validity differs between an
    fp32 and an fp64 evaluation of `bev_cell_index` (a coordinate within one fp32 rounding of a cell edge; about one point
    in 10^5), so that a float64 run of the same model is a usable ground truth for the fp32 one -- one re-binned point
    moves a logit by 1e-2.  A zero point is what the dataset pads with (pandaset_dataset.py:124-126)."""
    f32, v32 = bev_cell_index(points.float(), grid)
    f64, v64 = bev_cell_index(points.double(), grid)
    bad = (v32 != v64) | ((f32 != f64) & v32)
    out = points.clone()
    out[bad] = 0.0
    return out, int(bad.sum())


class _ScatterMaxZeroInit(torch.autograd.Function):
    """out[c, :] = max over rows r with idx[r]==c of src[r, :]; cells without a source stay 0.
    Forward == zeros.scatter_reduce_(0, idx, src, 'amax', include_self=False)
    (lidar_encoder.py:85-96).  Backward == ATen's amax derivative: the
    cell gradient is split evenly among all sources equal to the result, and -- because the
    zero-initialised destination slot takes part in the tie count -- a result of exactly 0
    counts one extra (phantom) tie."""

    @staticmethod
    def forward(ctx, src, idx, n_cells):
        C = src.shape[1]
        out = torch.full((n_cells, C), float("-inf"), dtype=src.dtype)
        out = out.index_reduce(0, idx, src, "amax", include_self=True)
        touched = torch.zeros(n_cells, dtype=torch.bool)
        touched[idx] = True
        out = torch.where(touched.unsqueeze(1), out, torch.zeros_like(out))
        ctx.save_for_backward(src, idx, out)
        return out

    @staticmethod
    def backward(ctx, g):
        src, idx, out = ctx.saved_tensors
        hit = (src == out[idx]).to(src.dtype)                      # [R, C]
        cnt = torch.zeros_like(out).index_add_(0, idx, hit)        # ties per cell/channel
        cnt = cnt + (out == 0).to(src.dtype)                       # phantom self slot
        gs = hit * (g / cnt.clamp_min(1.0))[idx]
        return gs, None, None


def spatial_lidar_encoder(points, st: State, p: str, grid: Tuple[int, int], training: bool,
                          x_range=(-50.0, 50.0), y_range=(-50.0, 50.0)):
    """lidar_encoder.py:57-99 (forward_vectorized).  `p` is e.g. 'lidar_encoder.encoder.'.
    The point MLP runs on ALL points (invalid ones take part in the BN1d batch stats)."""
    B, N, _ = points.shape
    H, W = grid
    h = points.transpose(1, 2)                                     # [B, 4, N]
    for i in (0, 3, 6):                                            # Conv1d k=1, BN1d, ReLU
        h = F.conv1d(h, st[f"{p}point_mlp.{i}.weight"], st[f"{p}point_mlp.{i}.bias"])
        h = _act(_bn(h, st, f"{p}point_mlp.{i+1}", training), "relu")
    C = h.shape[1]
    flat, valid = bev_cell_index(points, grid, x_range, y_range)
    feats = h.permute(0, 2, 1)[valid]                              # [n_valid, C]
    if feats.shape[0] == 0:
        fmap = torch.zeros(B * H * W, C, dtype=points.dtype)
    else:
        fmap = _ScatterMaxZeroInit.apply(feats, flat[valid], B * H * W)
    return fmap.view(B, H, W, C).permute(0, 3, 1, 2)               # NCHW view of NHWC memory


def spatial_lidar_encoder_iterative(points, st: State, p: str, grid: Tuple[int, int], training: bool,
                                    x_range=(-50.0, 50.0), y_range=(-50.0, 50.0)):
    """lidar_encoder.py:101-143 (forward_iterative, `use_vectorized=False`): the same point MLP and binning, then a
    running `maximum` into a zero-initialised map, one valid point at a time in point order.  Pure-Python loop: small
    inputs only.  Forward values are bit-identical to `spatial_lidar_encoder`; autograd differs only in how TIED maxima
    share a cell's gradient (a chain of torch.maximum halves it at every tie: 1/2 to the later point) -- identical
    points carry identical features, so the PARAMETER gradients still agree (tests/test_oracle_golden.py)."""
    B, N, _ = points.shape
    H, W = grid
    h = points.transpose(1, 2)
    for i in (0, 3, 6):
        h = F.conv1d(h, st[f"{p}point_mlp.{i}.weight"], st[f"{p}point_mlp.{i}.bias"])
        h = _act(_bn(h, st, f"{p}point_mlp.{i+1}", training), "relu")
    C = h.shape[1]
    xn, yn, valid = points_to_bev(points, x_range, y_range)
    ix = (xn * float(W - 1)).to(torch.int64).clamp(0, W - 1)
    iy = (yn * float(H - 1)).to(torch.int64).clamp(0, H - 1)
    cells = [[None] * (H * W) for _ in range(B)]                   # running maximum per touched cell
    for b in range(B):
        for n in torch.nonzero(valid[b]).flatten().tolist():
            c = int(iy[b, n]) * W + int(ix[b, n])
            cur = cells[b][c] if cells[b][c] is not None else torch.zeros(C, dtype=points.dtype)
            cells[b][c] = torch.maximum(cur, h[b, :, n])
    zero = torch.zeros(C, dtype=points.dtype)
    fmap = torch.stack([torch.stack([v if v is not None else zero for v in cells[b]]) for b in range(B)])   # [B, H*W, C]
    return fmap.view(B, H, W, C).permute(0, 3, 1, 2)


# --------------------------------------------------------------------------
# fusion / heads / full model   (src/models/fusion_module.py)
# --------------------------------------------------------------------------
def conv1x1_block(x, st, p, training):
    """Conv1x1: conv -> BN -> ReLU.  fusion_module.py:8-17."""
    return _pw_bn_act(x, st, p + ".conv.0", p + ".conv.1", "relu", training)


def dwsep_block(x, st, p, training, stride=1):
    """DWSeparableConv: dw3x3 -> BN -> ReLU -> pw -> BN -> ReLU (plain ReLU).  :20-34."""
    h = _dw_bn_act(x, st, p + ".net.0", p + ".net.1", "relu", stride, training)
    return _pw_bn_act(h, st, p + ".net.3", p + ".net.4", "relu", training)


def camera_fpn(feats: Dict[str, torch.Tensor], st, p, stages: Sequence[str], training):
    """CameraFPNLite.forward, fusion_module.py:51-64 (target_size=None)."""
    H, W = max((feats[s].shape[-2:] for s in stages), key=lambda hw: hw[0] * hw[1])
    acc = None
    for s in stages:
        x = conv1x1_block(feats[s], st, f"{p}.laterals.{s}", training)
        if tuple(x.shape[-2:]) != (H, W):
            x = F.interpolate(x, size=(H, W), mode="bilinear", align_corners=False)
        acc = x if acc is None else acc + x
    return dwsep_block(acc, st, p + ".post", training)


def seg_head_same(x, st, p, training):
    """SameResolutionSegmentationHead, fusion_module.py:162-173."""
    h = dwsep_block(x, st, p + ".block.0", training)
    h = dwsep_block(h, st, p + ".block.1", training)
    return F.conv2d(h, st[p + ".cls.weight"], st[p + ".cls.bias"])


def seg_head_x4(x, st, p, training):
    """LightweightSegmentationHead, fusion_module.py:142-159."""
    h = F.conv_transpose2d(x, st[p + ".up1.0.weight"], None, stride=2, padding=1)
    h = _act(_bn(h, st, p + ".up1.1", training), "relu")
    h = F.conv_transpose2d(h, st[p + ".up2.0.weight"], None, stride=2, padding=1)
    h = _act(_bn(h, st, p + ".up2.1", training), "relu")
    return F.conv2d(h, st[p + ".cls.weight"], st[p + ".cls.bias"], padding=1)


def complete_model(images, points, st: State, *, fusion_type: str, grid: Tuple[int, int],
                   fpn_stages: Optional[Sequence[str]] = ("stage3", "stage4", "stage5"),
                   multiscale: bool = True, output_mode: str = "same", training: bool = False):
    """CompleteSegmentationModel.forward(..., return_intermediates=True),
    fusion_module.py:234-263.  Returns (logits, intermediates)."""
    cam_raw = twinlite_encoder(images, st, "camera_encoder.", training, multiscale)
    if multiscale:
        stages = list(fpn_stages) if fpn_stages else ["stage2", "stage3", "stage4", "stage5"]
        cam = camera_fpn(cam_raw, st, "camera_fpn", stages, training)
    else:
        cam = cam_raw
    lid = spatial_lidar_encoder(points, st, "lidar_encoder.encoder.", grid, training)
    if cam.shape[-2:] != lid.shape[-2:]:
        lid = F.interpolate(lid, size=cam.shape[-2:], mode="bilinear", align_corners=False)
    if fusion_type == "concat":                                    # :242-246
        cp = conv1x1_block(cam, st, "fusion.camera_proj", training)
        lp = conv1x1_block(lid, st, "fusion.lidar_proj", training)
        pre = torch.cat([cp, lp], dim=1)
        h = _dw_bn_act(pre, st, "fusion.fuse.0", "fusion.fuse.1", "relu", 1, training)
        fused = _pw_bn_act(h, st, "fusion.fuse.3", "fusion.fuse.4", "relu", training)
    elif fusion_type in ("minimal", "weighted"):                   # :247-256
        cp = conv1x1_block(cam, st, "fusion.cam_proj", training)
        lp = conv1x1_block(lid, st, "fusion.lidar_proj", training)
        if fusion_type == "weighted":
            cat = torch.cat([cp, lp], dim=1)
            a = F.conv2d(cat, st["fusion.attention.0.weight"], st["fusion.attention.0.bias"])
            a = F.conv2d(torch.clamp_min(a, 0.0), st["fusion.attention.2.weight"],
                         st["fusion.attention.2.bias"])
            w = torch.softmax(a, dim=1)
            pre = cp * w[:, 0:1] + lp * w[:, 1:2]
        else:
            pre = cp + lp
        fused = pre
    else:
        raise ValueError(f"Unknown fusion_type: {fusion_type}")
    if output_mode == "same":
        logits = seg_head_same(fused, st, "head", training)
    elif output_mode == "x4":
        logits = seg_head_x4(fused, st, "head", training)
    else:
        raise ValueError(f"Unknown output_mode: {output_mode}")
    return logits, {"camera_feat": cam, "lidar_feat": lid, "pre_fusion": pre,
                    "post_fusion": fused, "logits": logits}


# --------------------------------------------------------------------------
# losses / metrics / optimiser   (src/training/trainer.py + build-defined KD)
# --------------------------------------------------------------------------
def weighted_ce(logits, target, class_weights: Optional[torch.Tensor], ignore_index: int = -1):
    """nn.CrossEntropyLoss(ignore_index=-1, weight=w), trainer.py:55:
    sum_i w[y_i] * nll_i / sum_i w[y_i] over non-ignored pixels."""
    lsm = torch.log_softmax(logits, dim=1)
    keep = target != ignore_index
    tgt = torch.where(keep, target, torch.zeros_like(target))
    nll = -lsm.gather(1, tgt.unsqueeze(1)).squeeze(1)
    w = class_weights[tgt] if class_weights is not None else torch.ones_like(nll)
    w = w * keep.to(w.dtype)
    return (w * nll).sum() / w.sum()


def kd_loss(student_logits, student_mids, teacher_logits, teacher_mids, target,
            class_weights=None, T: float = 4.0, alpha: float = 1.0, beta: float = 1.0,
            ignore_index: int = -1):
    """Build-defined KD objective (NOT in the reference, parity unpinned):
        L = CE_w(z_s, y) + alpha * T^2 * KL(softmax(z_t/T) || softmax(z_s/T))   [mean over B*h*w]
              + beta * (MSE(cam_s, cam_t) + MSE(lidar_s, lidar_t))              [mean over elems]
    Returns (total, dict of the parts)."""
    ce = weighted_ce(student_logits, target, class_weights, ignore_index)
    ps_log = torch.log_softmax(student_logits / T, dim=1)
    pt = torch.softmax(teacher_logits / T, dim=1)
    pt_log = torch.log_softmax(teacher_logits / T, dim=1)
    npix = student_logits.shape[0] * student_logits.shape[2] * student_logits.shape[3]
    kl = (pt * (pt_log - ps_log)).sum() / npix
    mse_c = F.mse_loss(student_mids["camera_feat"], teacher_mids["camera_feat"])
    mse_l = F.mse_loss(student_mids["lidar_feat"], teacher_mids["lidar_feat"])
    total = ce + alpha * (T * T) * kl + beta * (mse_c + mse_l)
    return total, {"ce": ce, "kl": kl, "mse_cam": mse_c, "mse_lidar": mse_l}


def confusion_matrix(logits, target, num_classes: int = 2, ignore_index: int = -1):
    """SegmentationMetrics.update, trainer.py:18-26, vectorised: argmax(dim=1) then
    confusion[t, p] += 1 for non-ignored pixels with 0 <= t,p < num_classes (int64)."""
    pred = torch.argmax(logits, dim=1).reshape(-1)
    tgt = target.reshape(-1)
    keep = (tgt != ignore_index) & (tgt >= 0) & (tgt < num_classes) & (pred < num_classes)
    flat = tgt[keep] * num_classes + pred[keep]
    return torch.bincount(flat, minlength=num_classes * num_classes).view(num_classes, num_classes)


def miou_from_confusion(conf):
    """SegmentationMetrics.compute, trainer.py:28-37."""
    ious = []
    for i in range(conf.shape[0]):
        tp = int(conf[i, i]); fp = int(conf[:, i].sum()) - tp; fn = int(conf[i, :].sum()) - tp
        d = tp + fp + fn
        ious.append(tp / d if d > 0 else 0.0)
    return ious, float(sum(ious) / len(ious))


def adamw_step(params: List[torch.Tensor], grads: List[torch.Tensor], exp_avg, exp_avg_sq,
               step: int, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-3):
    """torch.optim.AdamW single step (trainer.py:56 defaults), `step` is the 1-based count."""
    b1, b2 = betas
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for p, g, m, v in zip(params, grads, exp_avg, exp_avg_sq):
        p.mul_(1.0 - lr * weight_decay)
        m.mul_(b1).add_(g, alpha=1.0 - b1)
        v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def cosine_lr(base_lr: float, epoch: int, t_max: int, eta_min: float = 1e-5) -> float:
    """CosineAnnealingLR closed form (trainer.py:59-61), lr after `epoch` scheduler steps."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * epoch / t_max)) / 2.0


# --------------------------------------------------------------------------
# state helpers (shared by the golden generator and the tests)
# --------------------------------------------------------------------------
def trainable_keys(st: State) -> List[str]:
    """Keys of learnable tensors in registration order (everything that is not a buffer)."""
    out = []
    for k in st:
        leaf = k.rsplit(".", 1)[-1]
        if leaf in ("running_mean", "running_var", "num_batches_tracked", "x_range", "y_range",
                    "grid_tensor"):
            continue
        out.append(k)
    return out


def randomize_state(st: State, seed: int) -> State:
    """Deterministic, structure-independent re-initialisation keyed by tensor NAME, so the
    golden generator (reference modules) and the tests (product modules / oracle) build
    bit-identical weights without shipping them.  BN affine and running stats are made
    non-trivial on purpose (default init gamma=1, beta=0, mean=0, var=1 hides bugs)."""
    import zlib
    out: State = {}
    for k, v in st.items():
        g = torch.Generator().manual_seed((zlib.crc32(k.encode()) + 7919 * seed) & 0x7FFFFFFF)
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[k] = torch.zeros_like(v)
        elif leaf in ("x_range", "y_range", "grid_tensor"):
            out[k] = v.clone()
        elif leaf == "running_mean":
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif leaf == "running_var":
            out[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif v.dim() == 1 and leaf == "weight":                    # BN gamma
            out[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif v.dim() == 1:                                         # conv / BN bias
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        else:                                                      # conv weights
            fan_in = int(v[0].numel())
            out[k] = torch.randn(v.shape, generator=g) * math.sqrt(2.0 / fan_in)
    return out


def clone_state(st: State, requires_grad: bool = False) -> State:
    out = {}
    tk = set(trainable_keys(st))
    for k, v in st.items():
        t = v.detach().clone()
        if requires_grad and k in tk and t.is_floating_point():
            t.requires_grad_(True)
        out[k] = t
    return out


def make_inputs(B: int, HW: int, N: int, grid: int, seed: int, pad_tail: int = 0,
                num_classes: int = 2):
    """Synthetic PandaSet-shaped batch: image ~ U[0,1), points in the recipe of
    lidar_encoder.py:227-234, labels uniform with a few ignore_index pixels."""
    g = torch.Generator().manual_seed(1000 + seed)
    images = torch.rand(B, 3, HW, HW, generator=g)
    pts = torch.randn(B, N, 4, generator=g)
    pts[..., 0] *= 40.0
    pts[..., 1] *= 40.0
    pts[..., 2] = pts[..., 2] * 4.0 - 1.0
    pts[..., 3] = torch.sigmoid(pts[..., 3])
    if pad_tail:
        pts[:, N - pad_tail:, :] = 0.0                             # pandaset_dataset.py:124-126
    labels = torch.randint(0, num_classes, (B, grid, grid), generator=g)
    labels[:, 0, :3] = -1
    return images, pts, labels
