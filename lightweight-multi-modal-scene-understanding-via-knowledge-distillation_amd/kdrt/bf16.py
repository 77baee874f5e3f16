"""bf16-storage inference path (BASELINE.json configs[1]: camera + LiDAR concat-fusion forward in bf16).

A second, separately gated mode beside the fp32 contract: eval-mode forward only.  Activations live in HBM as bf16 NHWC
matrices, already normalised and activated (in eval mode every unit is ONE kernel: conv -> BatchNorm -> activation
[+ residual] -> bf16); accumulation is fp32; parameters and BatchNorm buffers stay the model's fp32 tensors.  The kernels are
csrc/kd_bf16.hip (C ABI `kd_bf16_*`).  Supported: the multiscale TwinLite encoder + FPN, the spatial LiDAR encoder, concat
minimal and weighted fusion, the same-resolution head -- what `train_with_fusion_ablation.py` builds.  A LiDAR grid
that differs from the camera map is resized by the fp32 bilinear kernel (the BEV grid stays fp32 in this mode).  Accuracy is that of 8-bit-mantissa activations; tests/test_gpu_bf16.py states the
measured logit error and argmax agreement against the fp32 path.

The KD step can run its frozen TEACHER through this path (`KDStep(..., teacher_storage="bf16")`): the teacher only supplies
targets (softened logits, two feature maps), no gradient flows through it, so its activations do not need fp32 storage;
the student's forward, backward and optimiser stay fp32.
"""
from __future__ import annotations

import torch

from . import ops, units
from .lib import KDError, lib
from .ops import ACT_RELU, ACT_RELU6, P, stream


import os as _os

_LIDAR_ONE_KERNEL = _os.environ.get("KD_BF16_LIDAR_ONE_KERNEL", "1") != "0"


def _coef(spec):
    C = spec.bn.running_mean.numel()
    return units._coeffs(spec, None, 0, C, 0, False, None, spec.bn.running_mean.device)


_IDENT = {}


def _identity(C, dev):
    """scale = 1, shift = 0: the epilogue coefficients of a convolution that has no BatchNorm (attention.0)."""
    key = (C, dev)
    if key not in _IDENT:
        _IDENT[key] = (torch.ones(C, device=dev), torch.zeros(C, device=dev))
    return _IDENT[key]


def _pw(x, spec, M, res=None, out=None, a_kind=0, m_dev=None):
    """x: bf16 [M, K] (or fp32 when a_kind == 1) -> bf16 [M, N] = act(bn(x . W^T + b)) (+ res)."""
    w, b = spec.conv.weight, spec.conv.bias
    N, K = w.shape[0], w.shape[1]
    bnc = _coef(spec)
    if out is None:
        out = torch.empty(M, N, device=w.device, dtype=torch.bfloat16)
    e0 = ops._prof_begin()
    lib.call("kd_bf16_pwconv", P(x), x.stride(0), a_kind, P(w), P(b), P(bnc.scale), P(bnc.shift), spec.act, P(out), out.stride(0),
             P(res), res.stride(0) if res is not None else 0, 0, M, K, N, P(m_dev), None, None, None, None, 0, None, None, 0, stream())
    ops._prof_end(e0, "bf16_pw", 2.0 * M * N * K, M * K * (4.0 if a_kind == 1 else 2.0) + M * N * (4.0 if res is not None else 2.0) + 4.0 * N * K)
    return out


def _dw(x, spec, geom):
    B, H, W = geom
    C = x.shape[1]
    s = spec.stride
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    bnc = _coef(spec)
    y = torch.empty(B * Ho * Wo, C, device=x.device, dtype=torch.bfloat16)
    e0 = ops._prof_begin()
    lib.call("kd_bf16_dwconv3x3", P(x), P(spec.conv.weight), P(bnc.scale), P(bnc.shift), spec.act, P(y), B, H, W, C, s, stream())
    ops._prof_end(e0, "bf16_dw", 18.0 * B * Ho * Wo * C, 2.0 * C * B * (H * W + Ho * Wo) + 36.0 * C)
    return y, (B, Ho, Wo)


def _chain(x, geom, specs, residual=False):
    cur, g = x, geom
    for i, u in enumerate(specs):
        last = i == len(specs) - 1
        if u.kind == "pw":
            cur = _pw(cur, u, cur.shape[0], res=x if (last and residual) else None)
        elif u.kind == "dw":
            cur, g = _dw(cur, u, g)
        else:
            raise KDError(f"bf16 path: unit kind {u.kind!r} is not supported")
    return cur, g


@torch.no_grad()
def forward_bf16(model, images: torch.Tensor, points: torch.Tensor, return_intermediates: bool = False):
    """Eval-mode forward of a CompleteSegmentationModel with bf16 activations; returns fp32 logits [B, C, h, w].
    With `return_intermediates` also the two feature maps the KD objective matches (fusion_module.py:260-262's
    `camera_feat` and `lidar_feat`), as fp32 NCHW-shaped views: (logits, {"camera_feat", "lidar_feat", "logits"})."""
    from src.models.fusion_module import ConcatenationFusion, MinimalFusion, SameResolutionSegmentationHead, WeightedFusion
    ops.require_gpu_tensor(images, "forward_bf16")
    if model.training:
        raise KDError("forward_bf16 is an inference path: call model.eval() first")
    enc = model.camera_encoder
    if not getattr(enc, "return_multiscale", False) or model.camera_fpn is None:
        raise KDError("bf16 path: needs the multiscale camera encoder + FPN (the training entry points' configuration)")
    if not isinstance(model.head, SameResolutionSegmentationHead):
        raise KDError("bf16 path: only the same-resolution head is supported")
    img = images.contiguous()
    B, Cin, H, W = img.shape
    dev = img.device
    # ---- camera encoder (camera_encoder.py:63-115) ------------------------------------------------------------------------
    stem = units.UnitSpec("stem", enc.stem[0], enc.stem[1], ACT_RELU6)
    bnc = _coef(stem)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = torch.empty(B * Ho * Wo, 32, device=dev, dtype=torch.bfloat16)
    e0 = ops._prof_begin()
    lib.call("kd_bf16_stem", P(img), P(enc.stem[0].weight), P(bnc.scale), P(bnc.shift), ACT_RELU6, P(x), B, Cin, H, W, 32, stream())
    ops._prof_end(e0, "bf16_stem", 2.0 * B * Ho * Wo * 32 * Cin * 9, 4.0 * B * Cin * H * W + 2.0 * B * Ho * Wo * 32)
    geom = (B, Ho, Wo)
    feats = {}
    for name in ("stage1", "stage2", "stage3", "stage4", "stage5"):
        blk = getattr(enc, name)
        x, geom = _chain(x, geom, blk._units(), residual=blk.use_residual)
        feats[name] = (x, geom)
    # ---- FPN (fusion_module.py:51-64) -------------------------------------------------------------------------------------------
    fpn = model.camera_fpn
    lats = []
    for s in fpn.stages_to_use:
        fx, fg = feats[s]
        lats.append((_pw(fx, fpn.laterals[s].unit(), fx.shape[0]), fg))
    if len(lats) > 3:
        raise KDError("bf16 path: at most three FPN stages")
    Hm, Wm = fpn.target_size if fpn.target_size is not None else max(((g[1], g[2]) for _, g in lats), key=lambda hw: hw[0] * hw[1])
    Ct = lats[0][0].shape[1]
    fused = torch.empty(B * Hm * Wm, Ct, device=dev, dtype=torch.bfloat16)
    a = [(P(t), g[1], g[2]) for t, g in lats] + [(None, 0, 0)] * (3 - len(lats))
    e0 = ops._prof_begin()
    lib.call("kd_bf16_bilinear_sum", a[0][0], a[0][1], a[0][2], a[1][0], a[1][1], a[1][2], a[2][0], a[2][1], a[2][2], P(fused), B, Hm, Wm, Ct,
             stream())
    ops._prof_end(e0, "bf16_resize_sum", 0.0, 2.0 * Ct * (sum(t.shape[0] for t, _ in lats) + B * Hm * Wm))
    cam, cgeom = _chain(fused, (B, Hm, Wm), fpn.post.units())
    # ---- LiDAR encoder (lidar_encoder.py:57-99): points sorted by cell, layer 0 recomputed inside the layer-1 GEMM, layer 2
    # scatter-maxes straight into the fp32 BEV grid (the [points, 128] layer outputs exist only as bf16 / not at all) ----------
    lenc = model.lidar_encoder.encoder
    Hg, Wg = lenc.grid_size
    pts = points.contiguous().view(-1, 4)
    Bp, Np = points.shape[0], points.shape[1]
    r = lenc.point_cloud_range
    rng = (float(r[0]), float(r[3]), float(r[1]), float(r[4]))
    if Hg * Wg + 1 > units.SORT_MAX_BINS:
        raise KDError("bf16 path: BEV grids above 192 x 192 cells are not supported")
    spts, cell, seg_start = units.sort_points(pts, Bp, Np, Hg, Wg, rng)
    counter = seg_start[Bp * Hg * Wg:]
    u0, u1, u2 = lenc._units()
    c0, c1, c2 = _coef(u0), _coef(u1), _coef(u2)
    Mp = Bp * Np
    K0, N1, N2 = u0.conv.weight.shape[0], u1.conv.weight.shape[0], u2.conv.weight.shape[0]
    all_relu = u0.act == ACT_RELU and u1.act == ACT_RELU and u2.act == ACT_RELU
    if _LIDAR_ONE_KERNEL and all_relu and lib.kd_bf16_lidar_mlp_scatter_supported(K0, N1, N2):
        # the whole point MLP + scatter-max in one kernel: no [points, 128] tensor in HBM at all (csrc/kd_lidar_infer.hip, NP = 1)
        grid = torch.empty(Bp * Hg * Wg, N2, device=dev, dtype=torch.float32)          # zeroed by the call
        e0 = ops._prof_begin()
        lib.call("kd_bf16_lidar_mlp_scatter", P(spts), P(cell), P(counter), P(u0.conv.weight), P(u0.conv.bias), P(c0.scale), P(c0.shift),
                 P(u1.conv.weight), P(u1.conv.bias), P(c1.scale), P(c1.shift), P(u2.conv.weight), P(u2.conv.bias), P(c2.scale), P(c2.shift),
                 P(grid), Bp * Hg * Wg, Mp, K0, N1, N2, stream())
        ops._prof_end(e0, "bf16_lidar", 2.0 * Mp * (K0 * N1 + N1 * N2), 20.0 * Mp + 4.0 * (K0 * N1 + N1 * N2), counter, Mp)
    else:
        y1 = torch.empty(Mp, N1, device=dev, dtype=torch.bfloat16)
        e0 = ops._prof_begin()
        lib.call("kd_bf16_pwconv", P(spts), 4, 3, P(u1.conv.weight), P(u1.conv.bias), P(c1.scale), P(c1.shift), u1.act, P(y1), N1, None, 0, 0,
                 Mp, K0, N1, P(counter), P(u0.conv.weight), P(u0.conv.bias), P(c0.scale), P(c0.shift), u0.act, None, None, 0, stream())
        ops._prof_end(e0, "bf16_pw", 2.0 * Mp * N1 * K0, 16.0 * Mp + 2.0 * Mp * N1 + 4.0 * N1 * K0, counter, Mp)
        grid = torch.zeros(Bp * Hg * Wg, N2, device=dev, dtype=torch.float32)
        e0 = ops._prof_begin()
        lib.call("kd_bf16_pwconv", P(y1), N1, 0, P(u2.conv.weight), P(u2.conv.bias), P(c2.scale), P(c2.shift), u2.act, None, 0, None, 0, 4,
                 Mp, N1, N2, P(counter), None, None, None, None, 0, P(cell), P(grid), N2, stream())
        ops._prof_end(e0, "bf16_pw", 2.0 * Mp * N2 * N1, 2.0 * Mp * N1 + 4.0 * Mp + 4.0 * N2 * N1, counter, Mp)
    lidar_map = grid
    if (Hg, Wg) != (cgeom[1], cgeom[2]):        # fusion_module.py:239-240; the BEV grid is fp32, so the fp32 resize serves
        lidar_map = torch.empty(Bp * cgeom[1] * cgeom[2], N2, device=dev, dtype=torch.float32)
        lib.call("kd_bilinear_accum_fwd", P(grid), None, None, ops.ACT_NONE, P(lidar_map), 0, Bp, Hg, Wg, cgeom[1], cgeom[2], N2, stream())
    # ---- fusion (fusion_module.py:242-255) ---------------------------------------------------------------------------------------
    M = cam.shape[0]
    fus = model.fusion
    if isinstance(fus, ConcatenationFusion):
        uc, ul = fus.camera_proj.unit(), fus.lidar_proj.unit()
        Cc, Cl = uc.conv.weight.shape[0], ul.conv.weight.shape[0]
        cat = torch.empty(M, Cc + Cl, device=dev, dtype=torch.bfloat16)
        _pw(cam, uc, M, out=cat[:, :Cc])
        _pw(lidar_map, ul, M, out=cat[:, Cc:], a_kind=1)
        from src.models.fusion_module import _dw_unit, _pw_unit
        fz, fgeom = _chain(cat, cgeom, [_dw_unit(fus.fuse, 0), _pw_unit(fus.fuse, 3)])
    elif isinstance(fus, MinimalFusion):
        t = _pw(cam, fus.cam_proj.unit(), M)
        fz, fgeom = _pw(lidar_map, fus.lidar_proj.unit(), M, res=t, a_kind=1), cgeom
    elif isinstance(fus, WeightedFusion):
        # fusion_module.py:115-120: both projections into one [M, 2C] buffer, attention.0 + ReLU as a plain bf16 GEMM
        # (identity "BatchNorm" coefficients), then conv 2C->2 + softmax + the weighted sum in one streaming pass
        uc, ul = fus.cam_proj.unit(), fus.lidar_proj.unit()
        C = uc.conv.weight.shape[0]
        if ul.conv.weight.shape[0] != C:
            raise KDError("bf16 path: weighted fusion needs equal projection widths")
        cat = torch.empty(M, 2 * C, device=dev, dtype=torch.bfloat16)
        _pw(cam, uc, M, out=cat[:, :C])
        _pw(lidar_map, ul, M, out=cat[:, C:], a_kind=1)
        a0, a2 = fus.attention[0], fus.attention[2]
        one, zero = _identity(C, dev)
        h = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
        lib.call("kd_bf16_pwconv", P(cat), 2 * C, 0, P(a0.weight), P(a0.bias), P(one), P(zero), ACT_RELU, P(h), C, None, 0, 0, M, 2 * C, C,
                 None, None, None, None, None, 0, None, None, 0, stream())
        fz = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
        e0 = ops._prof_begin()
        lib.call("kd_bf16_weighted_tail", P(h), P(cat), P(a2.weight), P(a2.bias), P(fz), M, C, stream())
        ops._prof_end(e0, "bf16_tail", 6.0 * M * C, 8.0 * M * C)
        fgeom = cgeom
    else:
        raise KDError(f"bf16 path: fusion block {type(fus).__name__} is not supported")
    # ---- head (fusion_module.py:162-173) -------------------------------------------------------------------------------------------
    hz, hgeom = _chain(fz, fgeom, model.head.block[0].units() + model.head.block[1].units())
    cls = model.head.cls
    NC, Cin_c = cls.weight.shape[0], cls.weight.shape[1]
    logits = torch.empty(B, NC, hgeom[1], hgeom[2], device=dev, dtype=torch.float32)
    e0 = ops._prof_begin()
    lib.call("kd_bf16_cls_conv", P(hz), P(cls.weight), P(cls.bias), P(logits), hz.shape[0], hgeom[1] * hgeom[2], Cin_c, NC, stream())
    ops._prof_end(e0, "bf16_cls", 2.0 * hz.shape[0] * Cin_c * NC, hz.shape[0] * (2.0 * Cin_c + 4.0 * NC))
    if return_intermediates:       # the camera map is widened once (layout plumbing); the BEV grid is fp32 already
        return logits, {"camera_feat": ops.nchw_from_matrix(cam.float(), cgeom), "lidar_feat": ops.nchw_from_matrix(lidar_map, (Bp, cgeom[1], cgeom[2])),
                        "logits": logits}
    return logits
