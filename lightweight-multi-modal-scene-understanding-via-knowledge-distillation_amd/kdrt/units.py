"""Conv+BN(+act) "units" and the autograd Functions that chain them.

A unit is one convolution followed by BatchNorm (and an activation id).  Its forward writes the
RAW conv output once and leaves BN+activation *deferred*: the next kernel applies them while
loading (ops.Operand).  Its backward consumes the gradient of its activated output -- either
unmasked ("D") or already multiplied by act' with the BN-backward sums attached ("G") -- and
returns parameter gradients plus the gradient for its own input in the same protocol.

Which reference module each Function stands in for is noted on the Function.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch

from . import gradsink, ops
from .lib import KDError, lib
from .ops import ACT_NONE, ACT_RELU, BNC, Operand, P, ld, stream


class UnitSpec:
    """kind: 'pw' (1x1 conv / Conv1d k=1), 'dw' (depthwise 3x3), 'stem' (3x3 s2 dense), 'l0' (LiDAR 4->C),
    'ct' (ConvTranspose2d k=4 s=2 p=1, bias-free: GEMM to [M_in, Cout*16] columns + col2im)."""

    def __init__(self, kind: str, conv, bn, act: int):
        self.kind, self.conv, self.bn, self.act = kind, conv, bn, act
        self.stride = conv.stride[0] if kind in ("dw", "stem") else 1
        self.has_bias = conv.bias is not None

    def params(self) -> List[torch.Tensor]:
        p = [self.conv.weight]
        if self.has_bias:
            p.append(self.conv.bias)
        return p + [self.bn.weight, self.bn.bias]


class _Rec:
    __slots__ = ("spec", "inp", "y", "bnc", "training", "out_geom", "w", "b", "gamma", "image")


# KD_EVAL_COEFF_CACHE=0 recomputes the eval coefficients on every use (triage of a suspected stale cache).  Contract for code
# that writes parameters or BatchNorm buffers behind torch's back -- an in-place op on the tensor itself under no_grad
# (`p.copy_()`, what load_state_dict does) bumps `_version` and is seen; ANY write through `p.data` (`p.data.copy_()`,
# `p.data.mul_()`: on this torch the version counter of `p` stays where it was), a raw pointer write, a kernel-side update or
# a collective on `.data` is not: call `kdrt.ops.bump_global_epoch()` afterwards (FusedAdamW, kd_bn_finalize_train,
# GraphedKDStep and the ddp broadcast do; a user-side EMA over `.data` must too).  load_state_dict also bumps it (post-hook
# installed by CompleteSegmentationModel / the encoders) so that a state dict whose tensors alias the old ones is never stale.
def stale_cache_guard(module):
    """load_state_dict post-hook of the model classes: whatever the loaded tensors alias, the content-keyed caches start over."""
    module.register_load_state_dict_post_hook(lambda m, incompatible: ops.bump_global_epoch())


_EVAL_COEFF_CACHE = os.environ.get("KD_EVAL_COEFF_CACHE", "1") != "0"
_DW_BWD_ADD = os.environ.get("KD_DW_BWD_ADD", "1") != "0"        # 0: residual gradient of a depthwise-first block added by a separate pass


def _coeffs(spec: UnitSpec, partial, rows, C, count, training, bnc=None, device=None):
    if training:
        bnc = bnc if bnc is not None else BNC(C, device)
        ops.bn_finalize_train(partial, rows, C, count, spec.bn, bnc)
        return bnc
    # eval mode: the coefficients only depend on the BN's own tensors -- cache them until one changes (a frozen teacher
    # otherwise recomputes 31 identical coefficient vectors every step).  The product path rewrites those tensors
    # through raw device pointers (fused AdamW, kd_bn_finalize_train, graph replays, broadcasts), which torch's
    # `_version` never sees, so the key carries the writers' own epochs as well: ops.bn_epoch (this module's running
    # statistics), the epoch of the optimiser that owns gamma / beta, and the process-wide ops.GLOBAL_EPOCH.
    bn = spec.bn
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
           ops.bn_epoch(bn), ops.owner_epoch(bn.weight), ops.owner_epoch(bn.bias), ops.GLOBAL_EPOCH[0])
    if bnc is None and _EVAL_COEFF_CACHE:
        hit = getattr(bn, "_kd_eval_cache", None)
        if hit is not None and hit[0] == key:
            return hit[1]
    fresh = bnc if bnc is not None else BNC(C, device)
    ops.bn_eval_coeffs(bn, fresh)
    if bnc is None:
        bn._kd_eval_cache = (key, fresh)
    return fresh


def unit_forward(spec: UnitSpec, inp, training: bool, y: Optional[torch.Tensor] = None, bnc: Optional[BNC] = None,
                 m_dev: Optional[torch.Tensor] = None, virtual: bool = False):
    """inp: Operand (or, for 'stem', the NCHW image tensor; for 'l0', the [P,4] points).
    Returns (output Operand, record for backward)."""
    rec = _Rec()
    rec.spec, rec.inp, rec.training = spec, inp, training
    w, b = spec.conv.weight, spec.conv.bias
    rec.w, rec.b, rec.gamma = w, b, spec.bn.weight
    rec.image = None
    kind = spec.kind
    if kind == "pw":
        N, K = w.shape[0], w.shape[1]
        M = inp.M
        if inp.C != K:
            raise KDError(f"pointwise conv expects {K} input channels, got {inp.C}")
        dev = w.device
        if y is None:
            y = torch.empty(M, N, device=dev, dtype=torch.float32)
        partial, rows = None, 0
        if training:
            pro = 3 if inp.virt is not None else (1 if inp.bnc is not None else 0)
            rows = lib.kd_pwconv_stat_rows_for(M, K, N, pro, 1, 0)         # streaming kernels: one row per wave; tiled: per 128 rows
            partial = torch.empty(rows * 2 * N, device=dev, dtype=torch.float32)
        if inp.virt is not None:
            ops.l1_fwd(inp, w, y, bias=b, epi=1 if training else 0, partial=partial, partial_rows=rows, m_dev=m_dev)
        else:
            ops.pw_gemm(inp.raw, w, y, M=M, K=K, N=N, pro=1 if inp.bnc is not None else 0, pro_act=inp.act,
                        p=(inp.sc, inp.sh, None, None, None), bias=b, epi=1 if training else 0, partial=partial, partial_rows=rows,
                        m_dev=m_dev)
        rec.out_geom = inp.geom
        rec.bnc = _coeffs(spec, partial, rows, N, M, training, bnc, dev)
    elif kind == "dw":
        B, H, W = inp.geom
        C = inp.C
        if ld(inp.raw) != C:
            raise KDError("depthwise conv needs a dense NHWC input")
        s = spec.stride
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        dev = inp.raw.device
        y = torch.empty(B * Ho * Wo, C, device=dev, dtype=torch.float32)
        partial, rows = None, 0
        if training:
            rows = lib.kd_dwconv_stat_rows(B * Ho * Wo, C)
            partial = torch.empty(rows * 2 * C, device=dev, dtype=torch.float32)
        lib.call("kd_dwconv3x3_fwd", P(inp.raw), P(inp.sc), P(inp.sh), inp.act, P(w), P(y), P(partial), B, H, W, C, s,
                 stream())
        rec.out_geom = (B, Ho, Wo)
        rec.bnc = _coeffs(spec, partial, rows, C, B * Ho * Wo, training, bnc, dev)
    elif kind == "stem":
        img = inp if inp.is_contiguous() else inp.contiguous()
        ops.require_gpu_tensor(img, "stem conv")
        B, Cin, H, W = img.shape
        Cout = w.shape[0]
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        dev = img.device
        y = torch.empty(B * Ho * Wo, Cout, device=dev, dtype=torch.float32)
        partial, rows = None, 0
        if training:
            rows = lib.kd_stem_stat_rows(B * Ho * Wo)
            partial = torch.empty(rows * 2 * Cout, device=dev, dtype=torch.float32)
        lib.call("kd_stem_conv_fwd", P(img), P(w), P(y), P(partial), B, Cin, H, W, Cout, stream())
        rec.image = img
        rec.out_geom = (B, Ho, Wo)
        rec.bnc = _coeffs(spec, partial, rows, Cout, B * Ho * Wo, training, bnc, dev)
    elif kind == "l0":
        pts = inp                                    # [P, 4] contiguous
        Pn, C = pts.shape[0], w.shape[0]
        dev = pts.device
        # virtual: the [P, C] output is never written -- only the BatchNorm statistics pass runs (training), and the
        # next layer recomputes layer 0 from the points on load
        y = None if virtual else torch.empty(Pn, C, device=dev, dtype=torch.float32)
        partial, rows = None, 0
        if training:
            rows = lib.kd_rowwise_stat_rows(Pn, C)
            partial = torch.empty(rows * 2 * C, device=dev, dtype=torch.float32)
        if training or not virtual:
            lib.call("kd_lidar_l0_fwd", P(pts), P(w), P(b), P(y), P(partial), Pn, C, P(m_dev), stream())
        rec.out_geom = (Pn, 1, 1)
        rec.bnc = _coeffs(spec, partial, rows, C, Pn, training, bnc, dev)
        if virtual:
            rec.y = None
            return Operand(None, rec.out_geom, rec.bnc, spec.act, virt=(pts, w, b)), rec
    elif kind == "ct":
        B, H, W = inp.geom
        Cin, Cout = w.shape[0], w.shape[1]
        if tuple(w.shape[2:]) != (4, 4) or inp.C != Cin:
            raise KDError(f"transposed conv expects a [{inp.C}, Cout, 4, 4] weight, got {tuple(w.shape)}")
        dev = inp.raw.device
        Wf = ops.transpose(w.view(Cin, Cout * 16))                     # [N = Cout*16, K = Cin]
        col = torch.empty(inp.M, Cout * 16, device=dev, dtype=torch.float32)
        ops.pw_gemm(inp.raw, Wf, col, M=inp.M, K=Cin, N=Cout * 16, pro=1 if inp.bnc is not None else 0, pro_act=inp.act,
                    p=(inp.sc, inp.sh, None, None, None), epi=0)
        Mo = B * 4 * H * W
        y = torch.empty(Mo, Cout, device=dev, dtype=torch.float32)
        partial, rows = None, 0
        if training:
            rows = lib.kd_deconv_stat_rows(Mo, Cout)
            partial = torch.empty(rows * 2 * Cout, device=dev, dtype=torch.float32)
        lib.call("kd_deconv4x4s2_col2im_fwd", P(col), P(y), P(partial), B, H, W, Cout, stream())
        rec.out_geom = (B, 2 * H, 2 * W)
        rec.bnc = _coeffs(spec, partial, rows, Cout, Mo, training, bnc, dev)
    else:
        raise KDError(f"unknown unit kind {kind}")
    rec.y = y
    return Operand(y, rec.out_geom, rec.bnc, spec.act), rec


def run_mode(training: bool) -> int:
    """0 eval with autograd, 1 training, 2 inference (eval AND grad mode off).  Evaluated by the run_* wrappers: inside
    autograd.Function.forward grad mode is always off, so the Functions cannot tell 0 from 2 themselves."""
    return 1 if training else (0 if torch.is_grad_enabled() else 2)


def inference_tail_ok(units, infer: bool) -> bool:
    """True when a chain may end in the fused inference epilogue: eval-mode BatchNorm statistics are known before the
    GEMM runs, and without autograd nobody needs the raw conv output afterwards."""
    return infer and len(units) > 0 and units[-1].kind == "pw"


def pw_forward_final(spec: UnitSpec, inp: Operand, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Inference tail of a chain: 1x1 conv + eval BatchNorm + activation (+ residual) in ONE GEMM (epilogue 5 of
    kd_pwconv_gemm) -- the same values, bit for bit, as the raw GEMM followed by kd_bn_act_apply, minus one pass."""
    w, b = spec.conv.weight, spec.conv.bias
    N, K = w.shape[0], w.shape[1]
    if inp.C != K:
        raise KDError(f"pointwise conv expects {K} input channels, got {inp.C}")
    dev = w.device
    bnc = _coeffs(spec, None, 0, N, inp.M, False, None, dev)
    y = torch.empty(inp.M, N, device=dev, dtype=torch.float32)
    ops.pw_gemm(inp.raw, w, y, M=inp.M, K=K, N=N, pro=1 if inp.bnc is not None else 0, pro_act=inp.act,
                p=(inp.sc, inp.sh, None, None, None), bias=b, addend=res, epi=5, esc=bnc.scale, esh=bnc.shift,
                epi_act=spec.act)
    return y


# Inference tails that end in (depthwise 3x3, 1x1): one kernel for both units (csrc/kd_block.hip) -- the depthwise output, the
# widest tensor of an InvertedResidual / DWSeparableConv, is never written.  Measured inside the KD step at 256 frames
# (profiles/r02_dw_pw_fusion.txt): the one-kernel form wins where the hidden tensor is wide and the output narrow (stage 2:
# 192 -> 64 at stride 2, 1065 -> 916 us; stage 3: 384 -> 64, 1211 -> 1085 us) and loses elsewhere (its 128-pixel tiles expose one
# HBM latency per 32-channel chunk), so mode 1 (default) uses it for those shapes only.  KD_DW_PW_FUSED: 0 never, 1 by shape, 2 always.
DW_PW_FUSED = [int(os.environ.get("KD_DW_PW_FUSED", "1"))]


def dw_pw_tail_ok(units, cur: Operand) -> bool:
    mode = DW_PW_FUSED[0]
    if mode == 0 or len(units) < 2 or units[-2].kind != "dw" or units[-1].kind != "pw":
        return False
    if ops.get_gemm_arithmetic() != "split" or not isinstance(cur, Operand):
        return False
    if ld(cur.raw) != cur.C:                                      # the depthwise input must be a dense NHWC matrix
        return False
    dwu, pwu = units[-2], units[-1]
    Ch, Cout = dwu.conv.weight.shape[0], pwu.conv.weight.shape[0]
    if not lib.kd_dw_pw_infer_supported(Ch, Cout, dwu.stride):
        return False
    return mode >= 2 or (Cout == 64 and Ch >= 192)


def dw_pw_forward_final(dwu: UnitSpec, pwu: UnitSpec, inp: Operand, res: Optional[torch.Tensor] = None):
    """(depthwise 3x3 + BN + act) -> (1x1 + BN + act) [+ residual] of an inference tail in one launch -> (matrix, geom);
    the same bits as unit_forward(dw) followed by pw_forward_final (tests/test_gpu_units.py)."""
    B, H, W = inp.geom
    Ch, Cout = inp.C, pwu.conv.weight.shape[0]
    if dwu.conv.weight.shape[0] != Ch or pwu.conv.weight.shape[1] != Ch:
        raise KDError(f"depthwise / pointwise pair expects {dwu.conv.weight.shape[0]} channels, got {Ch}")
    s = dwu.stride
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    dev = inp.raw.device
    bd = _coeffs(dwu, None, 0, Ch, B * Ho * Wo, False, None, dev)
    bp = _coeffs(pwu, None, 0, Cout, B * Ho * Wo, False, None, dev)
    out = torch.empty(B * Ho * Wo, Cout, device=dev, dtype=torch.float32)
    lib.call("kd_dw_pw_infer", P(inp.raw), P(inp.sc), P(inp.sh), inp.act, P(dwu.conv.weight), P(bd.scale), P(bd.shift), dwu.act,
             P(pwu.conv.weight), P(pwu.conv.bias), P(bp.scale), P(bp.shift), pwu.act, P(res), ld(res) if res is not None else 0,
             P(out), Cout, B, H, W, Ch, s, Cout, stream())
    return out, (B, Ho, Wo)


def infer_tail(units, cur, res: Optional[torch.Tensor] = None):
    """The units of an inference-mode chain whose last unit is a 1x1 conv -> (finished output matrix, geometry)."""
    n_head = len(units) - 2
    if n_head >= 0:
        head = cur
        for u in units[:n_head]:
            head, _ = unit_forward(u, head, False)
        if dw_pw_tail_ok(units, head):
            return dw_pw_forward_final(units[-2], units[-1], head, res)
        cur = head
        units = units[n_head:]
    for u in units[:-1]:
        cur, _ = unit_forward(u, cur, False)
    return pw_forward_final(units[-1], cur, res), cur.geom


def unit_backward(rec: _Rec, g, need_input_grad: bool = True, addend: Optional[torch.Tensor] = None):
    """g: ("D", dA) unmasked gradient w.r.t. the activated output, or ("G", G, partial, rows[, pstride])
    already masked with BN-backward sums.  Returns (param grads aligned with spec.params(), g_in) where
    g_in is ("G", ...) if the input was deferred, a plain [M, C] tensor if it was materialised, or None."""
    spec = rec.spec
    y, bnc = rec.y, rec.bnc
    M, C = y.shape if y is not None else (rec.inp.shape[0], rec.w.shape[0])      # y None: virtual LiDAR layer 0
    out_op = Operand(y, rec.out_geom, bnc, spec.act)
    pstride = None
    if g[0] == "D":
        t = g[1]
        partial, rows = ops.bn_bwd_reduce(t, out_op)
        msc, msh, mact = bnc.scale, bnc.shift, spec.act
        if mact == ACT_NONE:
            msc = msh = None
    else:
        # "G": t = masked gradient [M, C];  "GS": t = (row_sorted, grid, share) -- the LiDAR scatter-max gradient as
        # per-cell tables, rebuilt on load by kd_lidar_l2_dgrad / _wgrad (only a "pw" unit with a deferred input takes it)
        t, partial, rows = g[1], g[2], g[3]
        pstride = g[4] if len(g) > 4 else None
        msc, msh, mact = None, None, ACT_NONE
    tables = g[0] == "GS"
    if tables and not (spec.kind == "pw" and rec.inp.virt is None and rec.inp.bnc is not None):
        raise KDError("table-form scatter gradient needs a pointwise unit with a deferred input")
    beta_p = spec.bn.bias
    g_buf, g_dir = gradsink.out_for(rec.gamma)
    b_buf, b_dir = gradsink.out_for(beta_p)
    want_dbias = spec.has_bias and spec.kind != "l0"
    cb_buf, cb_dir = gradsink.out_for(rec.b) if want_dbias else (None, False)
    dgamma, dbeta, abg, dbias = ops.bn_bwd_finalize(partial, rows, C, M, rec.gamma, bnc, rec.training,
                                                    want_dbias=want_dbias, pstride=pstride, dgamma=g_buf, dbeta=b_buf,
                                                    dbias=cb_buf)
    al, be, ga = abg[0], abg[1], abg[2]
    dev = rec.w.device
    kind = spec.kind
    g_in = None
    if kind == "pw":
        inp = rec.inp
        N, K = C, inp.C
        dW, w_dir = gradsink.out_for(rec.w)
        fused = (tables and need_input_grad and addend is None and _LIDAR_FUSED_BWD and lib.kd_lidar_l2_bwd_supported(N, K)
                 and ld(y) == N and ld(inp.raw) == K)               # (dense operands: the one-kernel form addresses rows by 128)
        fused1 = (inp.virt is not None and need_input_grad and addend is None and _LIDAR_FUSED_BWD and _L0_MOMENTS and not tables
                  and mact == ACT_NONE and inp.act == ACT_RELU and lib.kd_lidar_l1_bwd_supported(N, K) and ld(t) == N and ld(y) == N)
        if fused or fused1:
            pass            # data gradient and weight gradient in one kernel, below
        elif tables:
            ops.l2_wgrad(t, out_op, dW, inp=inp, al=al, be=be, ga=ga)
        elif inp.virt is not None:
            ops.l1_wgrad(t, y, dW, op=inp, al=al, be=be, ga=ga, msc=msc, msh=msh, mact=mact)
        else:
            ops.pw_wgrad(t, inp.raw, dW, M=M, N=N, K=K, X=y, d_mode=2, d_act=mact, al=al, be=be, ga=ga, msc=msc, msh=msh,
                         a_mode=1 if inp.bnc is not None else 0, a_act=inp.act, asc=inp.sc, ash=inp.sh)
        if need_input_grad:
            Wt = ops.transpose(rec.w.view(N, K), owner=rec.w)
            if fused1:
                rows_in = lib.kd_lidar_l1_bwd_stat_rows(M)
                part_in = torch.empty(rows_in * 2 * K, device=dev, dtype=torch.float32)
                m1 = torch.empty(4, K, device=dev, dtype=torch.float32)
                ops.l1_bwd(t, y, Wt, dW, op=inp, al=al, be=be, ga=ga, partial=part_in, partial_rows=rows_in, moments=m1)
                g_in = ("GM", m1, part_in, rows_in)
            elif inp.virt is not None:
                if addend is not None:
                    raise KDError("addend on a virtual LiDAR layer-0 input is not supported")
                rows_in = lib.kd_lidar_l1_dgrad_stat_rows(M, N, K)
                part_in = torch.empty(rows_in * 2 * K, device=dev, dtype=torch.float32)
                if _L0_MOMENTS:
                    # layer 0's weight gradient is linear in G0: the GEMM epilogue leaves sum G0 * point and G0 is never written
                    m1 = torch.empty(4, K, device=dev, dtype=torch.float32)
                    ops.l1_dgrad(t, y, Wt, None, op=inp, al=al, be=be, ga=ga, msc=msc, msh=msh, mact=mact, partial=part_in,
                                 partial_rows=rows_in, moments=m1)
                    g_in = ("GM", m1, part_in, rows_in)
                else:
                    gin = torch.empty(M, K, device=dev, dtype=torch.float32)
                    ops.l1_dgrad(t, y, Wt, gin, op=inp, al=al, be=be, ga=ga, msc=msc, msh=msh, mact=mact, partial=part_in,
                                 partial_rows=rows_in)
                    g_in = ("G", gin, part_in, rows_in)
            elif fused:
                gin = torch.empty(M, K, device=dev, dtype=torch.float32)
                rows_in = lib.kd_lidar_l2_bwd_stat_rows(M)
                part_in = torch.empty(rows_in * 2 * K, device=dev, dtype=torch.float32)
                ops.l2_bwd(t, out_op, Wt, gin, dW, inp=inp, al=al, be=be, ga=ga, partial=part_in, partial_rows=rows_in)
                g_in = ("G", gin, part_in, rows_in)
            elif tables:
                gin = torch.empty(M, K, device=dev, dtype=torch.float32)
                rows_in = lib.kd_lidar_l2_dgrad_stat_rows(M, N, K)
                part_in = torch.empty(rows_in * 2 * K, device=dev, dtype=torch.float32)
                ops.l2_dgrad(t, out_op, Wt, gin, inp=inp, al=al, be=be, ga=ga, partial=part_in, partial_rows=rows_in)
                g_in = ("G", gin, part_in, rows_in)
            elif inp.bnc is not None:
                gin = torch.empty(M, K, device=dev, dtype=torch.float32)
                rows_in = lib.kd_pwconv_stat_rows_for(M, N, K, 2, 2, int(addend is not None))     # (reduction width N, output width K: the data gradient)
                part_in = torch.empty(rows_in * 2 * K, device=dev, dtype=torch.float32)
                ops.pw_gemm(t, Wt, gin, M=M, K=N, N=K, A2=y, pro=2, pro_act=mact, p=(al, be, ga, msc, msh),
                            addend=addend, epi=2, X=inp.raw, esc=inp.sc, esh=inp.sh, emean=inp.bnc.mean,
                            einv=inp.bnc.invstd, epi_act=inp.act, partial=part_in, partial_rows=rows_in)
                g_in = ("G", gin, part_in, rows_in)
            else:
                dx = torch.empty(M, K, device=dev, dtype=torch.float32)
                ops.pw_gemm(t, Wt, dx, M=M, K=N, N=K, A2=y, pro=2, pro_act=mact, p=(al, be, ga, msc, msh),
                            addend=addend, epi=0)
                g_in = dx
        grads = [gradsink.finish(rec.w, dW, w_dir)]
    elif kind == "dw":
        inp = rec.inp
        B, H, W = inp.geom
        s = spec.stride
        npix_out = M
        dW, w_dir = gradsink.out_for(rec.w)
        nbytes = lib.kd_dwconv_bwd_ws_bytes(npix_out, C)
        ws = ops.workspace(nbytes, dev)
        gx = part_in = None
        rows_in = 0
        deferred = inp.bnc is not None
        if need_input_grad:
            gx = torch.empty(inp.M, C, device=dev, dtype=torch.float32)
            if deferred:
                rows_in = lib.kd_dwconv_bwd_stat_rows(inp.M, C)
                part_in = torch.empty(rows_in * 2 * C, device=dev, dtype=torch.float32)
        # a residual gradient into the same input rides inside the kernel where the one-pass form allows it (round 3)
        add_in_kernel = (addend is not None and need_input_grad and _DW_BWD_ADD and ld(addend) == C
                         and lib.kd_dwconv3x3_bwd_add_supported(C, W, s))
        if add_in_kernel:
            lib.call("kd_dwconv3x3_bwd_add", P(t), P(y), P(al), P(be), P(ga), P(msc), P(msh), mact, P(inp.raw), P(inp.sc),
                     P(inp.sh), inp.act, P(inp.bnc.mean) if deferred else None, P(inp.bnc.invstd) if deferred else None,
                     P(rec.w), P(addend), P(gx), P(part_in), P(dW), B, H, W, C, s, P(ws), nbytes, stream())
        else:
            lib.call("kd_dwconv3x3_bwd", P(t), P(y), P(al), P(be), P(ga), P(msc), P(msh), mact, P(inp.raw), P(inp.sc),
                     P(inp.sh), inp.act, P(inp.bnc.mean) if deferred else None, P(inp.bnc.invstd) if deferred else None,
                     P(rec.w), P(gx), P(part_in), P(dW), B, H, W, C, s, P(ws), nbytes, stream())
        if need_input_grad:
            if deferred:
                if addend is not None and not add_in_kernel:
                    raise KDError("addend on a deferred depthwise input needs the one-pass stride-1 column-walk backward")
                g_in = ("G", gx, part_in, rows_in)
            else:
                if addend is not None and not add_in_kernel:
                    ops.bn_act_apply(gx, None, None, ACT_NONE, gx, res=addend)
                g_in = gx
        grads = [gradsink.finish(rec.w, dW, w_dir)]
    elif kind == "stem":
        img = rec.image
        B, Cin, H, W = img.shape
        Kp = 32
        col = torch.empty(M, Kp, device=dev, dtype=torch.float32)
        lib.call("kd_stem_im2col", P(img), P(col), B, Cin, H, W, Kp, stream())
        dWp = torch.empty(C, Kp, device=dev, dtype=torch.float32)
        ops.pw_wgrad(t, col, dWp, M=M, N=C, K=Kp, X=y, d_mode=2, d_act=mact, al=al, be=be, ga=ga, msc=msc, msh=msh)
        grads = [gradsink.deliver(rec.w, dWp[:, : Cin * 9])]
    elif kind == "l0":
        if g[0] not in ("G", "GM"):
            raise KDError("LiDAR layer-0 backward expects a masked gradient")
        pts = rec.inp
        nbytes = lib.kd_lidar_l0_bwd_ws_bytes(M, C)
        ws = ops.workspace(nbytes, dev)
        dwb = torch.empty(C * 5, device=dev, dtype=torch.float32)
        moments = g[0] == "GM"
        # dW0 = sum_p (al*G0 + be*y0 + ga) * pt.  "GM": t = sum_p G0 * pt ([4, C], from the layer-1 dgrad epilogue); this
        # pass then runs without G0 (D = NULL: the be*y0 + ga part, y0 recomputed from the point) and al * t is added.
        lib.call("kd_lidar_l0_bwd", None if moments else P(t), P(y), P(rec.w), P(rec.b), P(al), P(be), P(ga), P(pts), P(dwb), M, C,
                 P(ws), nbytes, stream())
        if moments:
            dwb[: C * 4].view(C, 4).addcmul_(al.view(C, 1), t.t())
            dwb[C * 4:].addcmul_(al, dbeta)                      # sum_p G0 == dbeta of this BatchNorm
        grads = gradsink.deliver_many([(rec.w, dwb[: C * 4]), (rec.b, dwb[C * 4:])])
    elif kind == "ct":
        inp = rec.inp
        B, H, W = inp.geom
        Cin, N16 = inp.C, C * 16
        dcol = torch.empty(inp.M, N16, device=dev, dtype=torch.float32)
        lib.call("kd_deconv4x4s2_im2col_bwd", P(t), P(y), P(al), P(be), P(ga), P(msc), P(msh), mact, P(dcol), B, H, W, C,
                 stream())
        dWt = torch.empty(N16, Cin, device=dev, dtype=torch.float32)     # (dW.view(Cin, Cout*16))^T
        ops.pw_wgrad(dcol, inp.raw, dWt, M=inp.M, N=N16, K=Cin, a_mode=1 if inp.bnc is not None else 0, a_act=inp.act,
                     asc=inp.sc, ash=inp.sh)
        if need_input_grad:
            Wd = rec.w.view(Cin, N16)                                     # dgrad operand [N = Cin, K = Cout*16] as stored
            if inp.bnc is not None:
                gin = torch.empty(inp.M, Cin, device=dev, dtype=torch.float32)
                rows_in = lib.kd_pwconv_stat_rows_for(inp.M, N16, Cin, 0, 2, int(addend is not None))
                part_in = torch.empty(rows_in * 2 * Cin, device=dev, dtype=torch.float32)
                ops.pw_gemm(dcol, Wd, gin, M=inp.M, K=N16, N=Cin, addend=addend, epi=2, X=inp.raw, esc=inp.sc, esh=inp.sh,
                            emean=inp.bnc.mean, einv=inp.bnc.invstd, epi_act=inp.act, partial=part_in, partial_rows=rows_in)
                g_in = ("G", gin, part_in, rows_in)
            else:
                dx = torch.empty(inp.M, Cin, device=dev, dtype=torch.float32)
                ops.pw_gemm(dcol, Wd, dx, M=inp.M, K=N16, N=Cin, addend=addend, epi=0)
                g_in = dx
        grads = [gradsink.deliver(rec.w, ops.transpose(dWt).view_as(rec.w))]
    if want_dbias:
        grads.append(gradsink.finish(rec.b, dbias, cb_dir))
    grads += [gradsink.finish(rec.gamma, dgamma, g_dir), gradsink.finish(beta_p, dbeta, b_dir)]
    return grads, g_in


def chain_backward(recs: Sequence[_Rec], g, need_input_grad=True, first_addend=None):
    """Walk a linear chain of units in reverse.  Returns (param grads in forward order, input grad)."""
    all_grads: List[torch.Tensor] = []
    for i in range(len(recs) - 1, -1, -1):
        first = i == 0
        pg, g = unit_backward(recs[i], g, need_input_grad=(need_input_grad or not first),
                              addend=first_addend if first else None)
        all_grads = pg + all_grads
    return all_grads, g


_STEM_INFER = os.environ.get("KD_STEM_INFER", "1") != "0"       # 0: stem conv and its BatchNorm + activation as two passes also in inference


def stem_infer(spec: UnitSpec, img):
    """Inference (eval, no autograd) stem: 3x3/s2 conv + BatchNorm + ReLU6 in ONE kernel (camera_encoder.py:63-67)."""
    img = img if img.is_contiguous() else img.contiguous()
    B, Cin, H, W = img.shape
    w = spec.conv.weight
    Cout = w.shape[0]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    bnc = _coeffs(spec, None, 0, Cout, B * Ho * Wo, False, None, img.device)
    y = torch.empty(B * Ho * Wo, Cout, device=img.device, dtype=torch.float32)
    lib.call("kd_stem_conv_fwd_infer", P(img), P(w), P(bnc.scale), P(bnc.shift), spec.act, P(y), B, Cin, H, W, Cout, stream())
    return y, (B, Ho, Wo)


def _params_of(units: Sequence[UnitSpec]) -> List[torch.Tensor]:
    out = []
    for u in units:
        out += u.params()
    return out


# =================================================================================================
class ChainFn(torch.autograd.Function):
    """x -> unit_1 -> ... -> unit_n -> materialise (+ x if residual).
    Stands in for: TwinLiteEncoder.stem (camera_encoder.py:63-67), InvertedResidual.forward (:46-51),
    Conv1x1.forward (fusion_module.py:16-17), DWSeparableConv.forward (:33-34)."""

    @staticmethod
    def forward(ctx, x, units, residual, training, *params):
        infer, training = training == 2, training == 1
        recs = []
        if units[0].kind == "stem":
            ops.require_gpu_tensor(x, "TwinLiteEncoder")
            cur, xm = x, None
        else:
            xm, geom = ops.nhwc_view(x)
            cur = Operand(xm, geom)
        if inference_tail_ok(units, infer):
            out, geom = infer_tail(units, cur, res=xm if residual else None)
            return ops.nchw_from_matrix(out, geom)
        if infer and _STEM_INFER and len(units) == 1 and units[0].kind == "stem" and not residual and x.shape[1] == 3:
            out, geom = stem_infer(units[0], x)          # conv + eval BatchNorm + activation in one kernel (same bits as the two passes)
            return ops.nchw_from_matrix(out, geom)
        for u in units:
            cur, rec = unit_forward(u, cur, training)
            recs.append(rec)
        out = ops.materialize(cur, res=xm if residual else None)
        ctx.recs, ctx.residual, ctx.geom = recs, residual, cur.geom
        return ops.nchw_from_matrix(out, cur.geom)

    @staticmethod
    def backward(ctx, dout):
        dm, _ = ops.nhwc_view(dout)
        need = ctx.needs_input_grad[0] and ctx.recs[0].spec.kind != "stem"
        first_addend = dm if ctx.residual else None
        dep = gradsink.collect_flagged(ctx.recs[0].inp.raw) if (need and ctx.recs[0].inp.bnc is None) else None
        if dep is not None:            # a later consumer of this chain's input left its gradient for the first data-gradient kernel
            g2, folded = dep
            if not ctx.residual:
                first_addend = g2 if folded is None else g2 - folded        # (a fold meant for a residual block that is not one)
            elif folded is None:
                first_addend = dm + g2                                      # two addends, one slot: one elementwise pass
            elif folded.data_ptr() == dm.data_ptr():
                first_addend = g2                                           # g2 already contains this block's skip-connection term dm
            else:
                first_addend = g2 + (dm - folded)                           # the output had a third consumer: correct the folded term
        grads, g_in = chain_backward(ctx.recs, ("D", dm), need_input_grad=need, first_addend=first_addend)
        dx = None
        if need:
            dx = ops.nchw_from_matrix(g_in, ctx.recs[0].inp.geom)
        return (dx, None, None, None, *grads)


def run_chain(x, units: Sequence[UnitSpec], residual: bool, training: bool):
    return ChainFn.apply(x, list(units), residual, run_mode(training), *_params_of(units))


class PairChainFn(torch.autograd.Function):
    """x -> chain A -> chain B (+ A's output: B is a residual block) with A's output never materialised: B's first unit and
    B's residual connection read A's raw last output with its BatchNorm coefficients on load; in backward the residual
    gradient enters B's first data-gradient kernel as an addend BEFORE the activation mask, so A receives a masked gradient
    with its BatchNorm-backward sums already formed (no kd_bn_act_apply, no kd_bn_bwd_reduce over that tensor).
    Stands in for: stem -> stage1 and stage2 -> stage3 of TwinLiteEncoder.forward (camera_encoder.py:98-104) in training."""

    @staticmethod
    def forward(ctx, x, units_a, units_b, training, *params):
        training = training == 1
        recs_a, recs_b = [], []
        if units_a[0].kind == "stem":
            ops.require_gpu_tensor(x, "TwinLiteEncoder")
            cur = x
        else:
            xm, geom = ops.nhwc_view(x)
            cur = Operand(xm, geom)
        for u in units_a:
            cur, rec = unit_forward(u, cur, training)
            recs_a.append(rec)
        mid = cur
        for u in units_b:
            cur, rec = unit_forward(u, cur, training)
            recs_b.append(rec)
        if cur.geom != mid.geom or cur.C != mid.C:
            raise KDError("PairChainFn: the second chain is not a residual block over the first chain's output")
        out = ops.materialize(cur, res_op=mid)
        ctx.recs_a, ctx.recs_b = recs_a, recs_b
        return ops.nchw_from_matrix(out, cur.geom)

    @staticmethod
    def backward(ctx, dout):
        dm, _ = ops.nhwc_view(dout)
        grads_b, g = chain_backward(ctx.recs_b, ("D", dm), need_input_grad=True, first_addend=dm)
        need = ctx.needs_input_grad[0] and ctx.recs_a[0].spec.kind != "stem"
        grads_a, g_in = chain_backward(ctx.recs_a, g, need_input_grad=need)
        dx = ops.nchw_from_matrix(g_in, ctx.recs_a[0].inp.geom) if need else None
        return (dx, None, None, None, *grads_a, *grads_b)


_CHAIN_PAIRS = os.environ.get("KD_CHAIN_PAIRS", "1") != "0"
_GRAD_ROUTING = os.environ.get("KD_GRAD_ROUTING", "1") != "0"     # 0: two-consumer feature gradients summed by autograd


def grad_routing(training: bool) -> bool:
    """May the encoder mark its two-consumer maps for gradient deposits (FPNFn.backward -> ChainFn.backward)?  Only inside a
    training step that checks `gradsink.pending()` after backward (a gradient sink is installed: KDStep, Trainer)."""
    return bool(_GRAD_ROUTING and training and torch.is_grad_enabled() and gradsink.active is not None and gradsink.active.step_open)


def chain_pair_ok(units_a: Sequence[UnitSpec], units_b: Sequence[UnitSpec], training: bool, out_w: int) -> bool:
    """May A -> B run as one PairChainFn?  Training mode only; B's first unit must be able to take the residual gradient
    inside its data-gradient kernel with a deferred input: any pointwise unit, a depthwise unit only in the stride-1
    column-walk form (`out_w`: width of A's output map)."""
    if not (_CHAIN_PAIRS and training and torch.is_grad_enabled()) or not units_a or not units_b:
        return False
    if units_a[-1].kind not in ("pw", "stem", "dw") or any(u.kind not in ("pw", "dw") for u in units_b):
        return False
    first = units_b[0]
    if first.kind == "dw":
        C = first.conv.weight.shape[0]
        return bool(_DW_BWD_ADD and first.stride == 1 and lib.kd_dwconv3x3_bwd_add_supported(C, out_w, 1))
    return True


def run_chain_pair(x, units_a: Sequence[UnitSpec], units_b: Sequence[UnitSpec], training: bool):
    return PairChainFn.apply(x, list(units_a), list(units_b), run_mode(training), *_params_of(units_a), *_params_of(units_b))


# =================================================================================================
class FPNFn(torch.autograd.Function):
    """CameraFPNLite.forward (fusion_module.py:51-64): laterals (Conv1x1) -> bilinear resize to the
    largest stage -> running sum -> DWSeparableConv post block."""

    @staticmethod
    def forward(ctx, n_stage, laterals, post_units, training, target_size, *tensors):
        infer, training = training == 2, training == 1
        feats = tensors[:n_stage]
        lat_ops, lat_recs, in_geoms = [], [], []
        for f, u in zip(feats, laterals):
            xm, geom = ops.nhwc_view(f)
            op, rec = unit_forward(u, Operand(xm, geom), training)
            lat_ops.append(op); lat_recs.append(rec); in_geoms.append(geom)
        B = in_geoms[0][0]
        Ho, Wo = target_size if target_size is not None else max(((g[1], g[2]) for g in in_geoms), key=lambda hw: hw[0] * hw[1])
        Ct = lat_ops[0].C
        dev = lat_ops[0].raw.device
        fused = torch.empty(B * Ho * Wo, Ct, device=dev, dtype=torch.float32)
        if len(lat_ops) <= 3:        # the whole sum in one pass (same bits as accumulating lateral by lateral)
            a = [(P(op.raw), P(op.sc), P(op.sh), op.act, op.geom[1], op.geom[2]) for op in lat_ops] + [(None, None, None, 0, 0, 0)] * (3 - len(lat_ops))
            lib.call("kd_bilinear_sum_fwd", *a[0], *a[1], *a[2], P(fused), B, Ho, Wo, Ct, stream())
        else:
            for i, op in enumerate(lat_ops):
                _, Hi, Wi = op.geom
                lib.call("kd_bilinear_accum_fwd", P(op.raw), P(op.sc), P(op.sh), op.act, P(fused), int(i > 0), B, Hi, Wi,
                         Ho, Wo, Ct, stream())
        cur = Operand(fused, (B, Ho, Wo))
        post_recs = []
        if inference_tail_ok(post_units, infer):
            out, geom = infer_tail(post_units, cur)
            return ops.nchw_from_matrix(out, geom)
        for u in post_units:
            cur, rec = unit_forward(u, cur, training)
            post_recs.append(rec)
        out = ops.materialize(cur)
        ctx.lat_ops, ctx.lat_recs, ctx.post_recs, ctx.geom, ctx.n_stage = lat_ops, lat_recs, post_recs, (B, Ho, Wo), n_stage
        # Gradient routing for maps with a second consumer (marks set by TwinLiteEncoder.forward, training only): a map marked
        # `_kd_deposit` is ALSO the input of the next encoder stage, whose backward runs after this one -- its gradient from here
        # is deposited for that stage's data-gradient kernel instead of being summed by an autograd accumulation pass; a map
        # marked `_kd_residual_of = other` is the output of a residual block over `other`, so its gradient is folded into
        # `other`'s deposit (the block's skip-connection term) by the lateral's own data-gradient kernel.
        ctx.deposit = [bool(getattr(f, "_kd_deposit", False)) and _GRAD_ROUTING and training for f in feats]
        ctx.res_src = [None] * n_stage                        # res_src[j] = i: feats[i] is a residual block's output over feats[j]
        for i, f in enumerate(feats):
            tgt = getattr(f, "_kd_residual_of", None)
            for j, h in enumerate(feats):
                if tgt is not None and h is tgt and ctx.deposit[j] and f.shape == h.shape:
                    ctx.res_src[j] = i
        return ops.nchw_from_matrix(out, cur.geom)

    @staticmethod
    def backward(ctx, dout):
        dm, _ = ops.nhwc_view(dout)
        post_grads, dfused = chain_backward(ctx.post_recs, ("D", dm), need_input_grad=True)
        B, Ho, Wo = ctx.geom
        feat_grads, lat_grads = [], []
        n = ctx.n_stage
        order = sorted(range(n), key=lambda k: 0 if k in ctx.res_src else 1)       # residual sources first: their gradient is an addend
        gmat, pgs = [None] * n, [None] * n
        for i in order:
            op, rec = ctx.lat_ops[i], ctx.lat_recs[i]
            _, Hi, Wi = op.geom
            C = op.C
            rows = lib.kd_rowwise_stat_rows(op.M, C)
            gin = torch.empty(op.M, C, device=dm.device, dtype=torch.float32)
            partial = torch.empty(rows * 2 * C, device=dm.device, dtype=torch.float32)
            lib.call("kd_bilinear_bwd", P(dfused), P(op.raw), P(op.sc), P(op.sh), op.act, P(op.bnc.mean),
                     P(op.bnc.invstd), P(gin), P(partial), B, Hi, Wi, Ho, Wo, C, stream())
            need = ctx.needs_input_grad[5 + i]
            src = ctx.res_src[i]
            fold = need and src is not None and gmat[src] is not None
            pgs[i], gmat[i] = unit_backward(rec, ("G", gin, partial, rows), need_input_grad=need, addend=gmat[src] if fold else None)
            if need and ctx.deposit[i]:
                gradsink.deposit(rec.inp.raw, gmat[i], folded_residual=gmat[src] if fold else None)
        for i in range(n):
            lat_grads += pgs[i]
            keep = ctx.needs_input_grad[5 + i] and not ctx.deposit[i]
            feat_grads.append(ops.nchw_from_matrix(gmat[i], ctx.lat_recs[i].inp.geom) if keep else None)
        return (None, None, None, None, None, *feat_grads, *lat_grads, *post_grads)


def run_fpn(feats: Sequence[torch.Tensor], laterals: Sequence[UnitSpec], post_units: Sequence[UnitSpec], training,
            target_size=None):
    return FPNFn.apply(len(feats), list(laterals), list(post_units), run_mode(training),
                       tuple(target_size) if target_size is not None else None, *feats, *_params_of(laterals),
                       *_params_of(post_units))


# =================================================================================================
class ResizeFn(torch.autograd.Function):
    """F.interpolate(x, size, mode="bilinear", align_corners=False) on the HIP path -- the LiDAR-to-camera
    grid resize of CompleteSegmentationModel.forward (fusion_module.py:239-240) and of the fusion
    blocks' own forwards (:87-88, :102-103, :123-124)."""

    @staticmethod
    def forward(ctx, x, size):
        xm, (B, Hi, Wi) = ops.nhwc_view(x)
        Ho, Wo = size
        C = xm.shape[1]
        out = torch.empty(B * Ho * Wo, C, device=xm.device, dtype=torch.float32)
        lib.call("kd_bilinear_accum_fwd", P(xm), None, None, ACT_NONE, P(out), 0, B, Hi, Wi, Ho, Wo, C, stream())
        ctx.geom = (B, Hi, Wi, Ho, Wo, C)
        return ops.nchw_from_matrix(out, (B, Ho, Wo))

    @staticmethod
    def backward(ctx, dout):
        dm, _ = ops.nhwc_view(dout)
        B, Hi, Wi, Ho, Wo, C = ctx.geom
        gin = torch.empty(B * Hi * Wi, C, device=dm.device, dtype=torch.float32)
        lib.call("kd_bilinear_bwd", P(dm), None, None, None, ACT_NONE, None, None, P(gin), None, B, Hi, Wi, Ho, Wo, C, stream())
        return ops.nchw_from_matrix(gin, (B, Hi, Wi)), None


def run_resize(x, size):
    return ResizeFn.apply(x, (int(size[0]), int(size[1])))


# =================================================================================================
def _proj_pair(cam, lid, u_cam: UnitSpec, u_lid: UnitSpec, training):
    """Run the two Conv1x1 projections into ONE [M, Cc+Cl] raw buffer (the concat is free) with one
    combined coefficient table."""
    cm, geom = ops.nhwc_view(cam)
    lm, geom_l = ops.nhwc_view(lid)
    if geom != geom_l:
        raise KDError(f"camera / LiDAR feature maps differ in size: {geom} vs {geom_l} (resize path not built)")
    Cc, Cl = u_cam.conv.weight.shape[0], u_lid.conv.weight.shape[0]
    dev = cm.device
    cat = torch.empty(cm.shape[0], Cc + Cl, device=dev, dtype=torch.float32)
    comb = BNC(Cc + Cl, dev)
    op_c, rec_c = unit_forward(u_cam, Operand(cm, geom), training, y=cat[:, :Cc], bnc=BNC(Cc, dev, comb.buf[:, :Cc]))
    op_l, rec_l = unit_forward(u_lid, Operand(lm, geom), training, y=cat[:, Cc:], bnc=BNC(Cl, dev, comb.buf[:, Cc:]))
    return cat, comb, (op_c, rec_c), (op_l, rec_l), geom


def _feature_addend(rec, need_input_grad):
    """A gradient another consumer of this projection's input map left for it (gradsink.deposit): summed into the data
    gradient by the GEMM itself.  Only a materialised input can have one."""
    if not need_input_grad or rec.inp.bnc is not None or rec.inp.raw is None:
        return None
    return gradsink.collect(rec.inp.raw)


def _proj_pair_backward(ctx, gcat, partial, rows, Cc, Cl, need_cam, need_lid):
    """Split the masked gradient of the concat buffer between the two projection units."""
    tot = Cc + Cl
    g_c = ("G", gcat[:, :Cc], partial, rows, tot)
    g_l = ("G", gcat[:, Cc:], partial[Cc:], rows, tot)
    pg_c, dcam = unit_backward(ctx.rec_c, g_c, need_input_grad=need_cam, addend=_feature_addend(ctx.rec_c, need_cam))
    pg_l, dlid = unit_backward(ctx.rec_l, g_l, need_input_grad=need_lid, addend=_feature_addend(ctx.rec_l, need_lid))
    dcam = ops.nchw_from_matrix(dcam, ctx.rec_c.inp.geom) if need_cam else None
    dlid = ops.nchw_from_matrix(dlid, ctx.rec_l.inp.geom) if need_lid else None
    return dcam, dlid, pg_c, pg_l


class ConcatFuseFn(torch.autograd.Function):
    """The concat branch of CompleteSegmentationModel.forward (fusion_module.py:242-246):
    camera_proj / lidar_proj -> cat -> fuse (dw3x3+BN+ReLU, pw+BN+ReLU).
    Returns (fused, pre_fusion); pre_fusion is a by-product for `return_intermediates` and is
    marked non-differentiable."""

    @staticmethod
    def forward(ctx, cam, lid, u_cam, u_lid, fuse_units, training, want_pre, *params):
        infer, training = training == 2, training == 1
        cat, comb, (op_c, rec_c), (op_l, rec_l), geom = _proj_pair(cam, lid, u_cam, u_lid, training)
        cur = Operand(cat, geom, comb, ACT_RELU)
        # the activated concat buffer exists only for callers that ask for `pre_fusion` (a 2 GB pass at 256 frames otherwise
        # paid by every forward: the fuse block itself reads the raw buffer with the BatchNorm coefficients on load)
        pre_t = ops.nchw_from_matrix(ops.materialize(cur), geom) if want_pre else cat.new_empty(0)
        ctx.mark_non_differentiable(pre_t)
        recs = []
        if inference_tail_ok(fuse_units, infer):
            out, ogeom = infer_tail(fuse_units, cur)
            return ops.nchw_from_matrix(out, ogeom), pre_t
        for u in fuse_units:
            cur, rec = unit_forward(u, cur, training)
            recs.append(rec)
        out = ops.materialize(cur)
        ctx.rec_c, ctx.rec_l, ctx.recs = rec_c, rec_l, recs
        ctx.Cc, ctx.Cl = op_c.C, op_l.C
        return ops.nchw_from_matrix(out, geom), pre_t

    @staticmethod
    def backward(ctx, dout, _dpre):
        dm, _ = ops.nhwc_view(dout)
        fuse_grads, g = chain_backward(ctx.recs, ("D", dm), need_input_grad=True)
        _, gcat, partial, rows = g
        dcam, dlid, pg_c, pg_l = _proj_pair_backward(ctx, gcat, partial, rows, ctx.Cc, ctx.Cl,
                                                     ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return (dcam, dlid, None, None, None, None, None, *pg_c, *pg_l, *fuse_grads)


def run_concat_fuse(cam, lid, u_cam, u_lid, fuse_units, training, want_pre=True):
    """-> (fused, pre_fusion); pre_fusion is an empty tensor when `want_pre` is false."""
    return ConcatFuseFn.apply(cam, lid, u_cam, u_lid, list(fuse_units), run_mode(training), bool(want_pre), *u_cam.params(),
                              *u_lid.params(), *_params_of(fuse_units))


class WeightedFuseFn(torch.autograd.Function):
    """The weighted branch (fusion_module.py:248-253 with WeightedFusion.attention :115-120):
    cam_proj / lidar_proj -> cat -> conv1x1(2C->C, bias)+ReLU -> conv1x1(C->2, bias) -> softmax ->
    cam_proj*w0 + lidar_proj*w1."""

    @staticmethod
    def forward(ctx, cam, lid, u_cam, u_lid, training, w1, b1, w2, b2, *params):
        cat, comb, (op_c, rec_c), (op_l, rec_l), geom = _proj_pair(cam, lid, u_cam, u_lid, training)
        M, C = cat.shape[0], op_c.C
        if op_l.C != C or w1.shape[0] != C or w1.shape[1] != 2 * C or w2.shape[0] != 2:
            raise KDError("weighted fusion expects equal projection widths and a 2-way attention head")
        dev = cat.device
        hraw = torch.empty(M, C, device=dev, dtype=torch.float32)
        ops.pw_gemm(cat, w1, hraw, M=M, K=2 * C, N=C, pro=1, pro_act=ACT_RELU, p=(comb.scale, comb.shift, None, None, None),
                    bias=b1, epi=0)
        out = torch.empty(M, C, device=dev, dtype=torch.float32)
        wts = torch.empty(M, 2, device=dev, dtype=torch.float32)
        lib.call("kd_weighted_fuse_fwd", P(cat), P(comb.scale), P(comb.shift), P(hraw), P(w2), P(b2), P(out), P(wts), M, C,
                 stream())
        ctx.rec_c, ctx.rec_l, ctx.cat, ctx.comb, ctx.hraw, ctx.wts = rec_c, rec_l, cat, comb, hraw, wts
        ctx.w1, ctx.w2, ctx.b1, ctx.b2, ctx.C, ctx.geom = w1, w2, b1, b2, C, geom
        return ops.nchw_from_matrix(out, geom)

    @staticmethod
    def backward(ctx, dout):
        dm, _ = ops.nhwc_view(dout)
        C, cat, comb = ctx.C, ctx.cat, ctx.comb
        M = cat.shape[0]
        dev = cat.device
        dcat = torch.empty(M, 2 * C, device=dev, dtype=torch.float32)
        gh = torch.empty(M, C, device=dev, dtype=torch.float32)
        dpar = torch.empty(3 * C + 4, device=dev, dtype=torch.float32)
        nbytes = lib.kd_weighted_fuse_bwd_ws_bytes(M, C)
        ws = ops.workspace(nbytes, dev)
        lib.call("kd_weighted_fuse_bwd", P(dm), P(cat), P(comb.scale), P(comb.shift), P(ctx.hraw), P(ctx.w2), P(ctx.wts),
                 P(dcat), P(gh), P(dpar), M, C, P(ws), nbytes, stream())
        dw2, db1, db2 = gradsink.deliver_many([(ctx.w2, dpar[: 2 * C]), (ctx.b1, dpar[2 * C: 3 * C]), (ctx.b2, dpar[3 * C: 3 * C + 2])])
        dw1, w1_dir = gradsink.out_for(ctx.w1)
        ops.pw_wgrad(gh, cat, dw1, M=M, N=C, K=2 * C, d_mode=0, a_mode=1, a_act=ACT_RELU, asc=comb.scale, ash=comb.shift)
        w1t = ops.transpose(ctx.w1.view(C, 2 * C), owner=ctx.w1)
        rows = lib.kd_pwconv_stat_rows_for(M, C, 2 * C, 0, 2, 1)           # (reduction width C, output width 2C)
        partial = torch.empty(rows * 2 * 2 * C, device=dev, dtype=torch.float32)
        gcat = torch.empty(M, 2 * C, device=dev, dtype=torch.float32)
        ops.pw_gemm(gh, w1t, gcat, M=M, K=C, N=2 * C, pro=0, addend=dcat, epi=2, X=cat, esc=comb.scale, esh=comb.shift,
                    emean=comb.mean, einv=comb.invstd, epi_act=ACT_RELU, partial=partial, partial_rows=rows)
        dcam, dlid, pg_c, pg_l = _proj_pair_backward(ctx, gcat, partial, rows, C, C, ctx.needs_input_grad[0],
                                                     ctx.needs_input_grad[1])
        return (dcam, dlid, None, None, None, gradsink.finish(ctx.w1, dw1, w1_dir), db1, dw2, db2, *pg_c, *pg_l)


def run_weighted_fuse(cam, lid, u_cam, u_lid, att0, att2, training):
    return WeightedFuseFn.apply(cam, lid, u_cam, u_lid, training, att0.weight, att0.bias, att2.weight, att2.bias,
                                *u_cam.params(), *u_lid.params())


class MinimalFuseFn(torch.autograd.Function):
    """The minimal branch (fusion_module.py:248-249,255): cam_proj(cam) + lidar_proj(lidar)."""

    @staticmethod
    def forward(ctx, cam, lid, u_cam, u_lid, training, *params):
        cm, geom = ops.nhwc_view(cam)
        lm, geom_l = ops.nhwc_view(lid)
        if geom != geom_l:
            raise KDError("camera / LiDAR feature maps differ in size (resize path not built)")
        op_c, rec_c = unit_forward(u_cam, Operand(cm, geom), training)
        op_l, rec_l = unit_forward(u_lid, Operand(lm, geom), training)
        t = ops.materialize(op_c)
        out = ops.materialize(op_l, res=t, out=t)
        ctx.rec_c, ctx.rec_l = rec_c, rec_l
        return ops.nchw_from_matrix(out, geom)

    @staticmethod
    def backward(ctx, dout):
        dm, _ = ops.nhwc_view(dout)
        pg_c, dcam = unit_backward(ctx.rec_c, ("D", dm), need_input_grad=ctx.needs_input_grad[0],
                                   addend=_feature_addend(ctx.rec_c, ctx.needs_input_grad[0]))
        pg_l, dlid = unit_backward(ctx.rec_l, ("D", dm), need_input_grad=ctx.needs_input_grad[1],
                                   addend=_feature_addend(ctx.rec_l, ctx.needs_input_grad[1]))
        dcam = ops.nchw_from_matrix(dcam, ctx.rec_c.inp.geom) if dcam is not None else None
        dlid = ops.nchw_from_matrix(dlid, ctx.rec_l.inp.geom) if dlid is not None else None
        return (dcam, dlid, None, None, None, *pg_c, *pg_l)


def run_minimal_fuse(cam, lid, u_cam, u_lid, training):
    return MinimalFuseFn.apply(cam, lid, u_cam, u_lid, training, *u_cam.params(), *u_lid.params())


# =================================================================================================
class SameHeadFn(torch.autograd.Function):
    """SameResolutionSegmentationHead.forward (fusion_module.py:172-173): two DWSeparableConv blocks
    then the 1x1 classifier (bias) writing NCHW logits."""

    @staticmethod
    def forward(ctx, x, units, training, wc, bc, *params):
        xm, geom = ops.nhwc_view(x)
        cur = Operand(xm, geom)
        recs = []
        for u in units:
            cur, rec = unit_forward(u, cur, training)
            recs.append(rec)
        B, H, W = cur.geom
        NC, Cin = wc.shape[0], wc.shape[1]
        logits = torch.empty(B, NC, H, W, device=xm.device, dtype=torch.float32)
        lib.call("kd_cls_conv_fwd", P(cur.raw), P(cur.sc), P(cur.sh), cur.act, P(wc), P(bc), P(logits), cur.M, H * W, Cin,
                 NC, stream())
        ctx.recs, ctx.last, ctx.wc, ctx.bc = recs, cur, wc, bc
        return logits

    @staticmethod
    def backward(ctx, dlog):
        dlog = dlog.contiguous()
        cur, wc = ctx.last, ctx.wc
        B, H, W = cur.geom
        NC, Cin = wc.shape[0], wc.shape[1]
        M = cur.M
        dev = dlog.device
        gx = torch.empty(M, Cin, device=dev, dtype=torch.float32)
        rows = lib.kd_cls_conv_bwd_stat_rows(M, Cin)
        partial = torch.empty(rows * 2 * Cin, device=dev, dtype=torch.float32)
        dwb = torch.empty(NC * Cin + 4, device=dev, dtype=torch.float32)
        nbytes = lib.kd_cls_conv_bwd_ws_bytes(M, Cin, NC)
        ws = ops.workspace(nbytes, dev)
        lib.call("kd_cls_conv_bwd", P(dlog), P(cur.raw), P(cur.sc), P(cur.sh), cur.act, P(cur.bnc.mean), P(cur.bnc.invstd),
                 P(wc), P(gx), P(partial), P(dwb), M, H * W, Cin, NC, P(ws), nbytes, stream())
        grads, g_in = chain_backward(ctx.recs, ("G", gx, partial, rows), need_input_grad=ctx.needs_input_grad[0])
        dx = ops.nchw_from_matrix(g_in, ctx.recs[0].inp.geom) if ctx.needs_input_grad[0] else None
        dwc, dbc = gradsink.deliver_many([(wc, dwb[: NC * Cin]), (ctx.bc, dwb[NC * Cin: NC * Cin + NC])])
        return (dx, None, None, dwc, dbc,
                *grads)


def run_same_head(x, units, cls_conv, training):
    return SameHeadFn.apply(x, list(units), training, cls_conv.weight, cls_conv.bias, *_params_of(units))


class X4HeadFn(torch.autograd.Function):
    """LightweightSegmentationHead.forward (fusion_module.py:158-159): two ConvTranspose2d(4, 2, 1)+BN+ReLU
    stages (each doubles H and W) then the 3x3 classifier (bias) writing NCHW logits."""

    @staticmethod
    def forward(ctx, x, units, training, wc, bc, *params):
        xm, geom = ops.nhwc_view(x)
        cur = Operand(xm, geom)
        recs = []
        for u in units:
            cur, rec = unit_forward(u, cur, training)
            recs.append(rec)
        B, H, W = cur.geom
        NC, Cin = wc.shape[0], wc.shape[1]
        logits = torch.empty(B, NC, H, W, device=xm.device, dtype=torch.float32)
        lib.call("kd_cls3x3_fwd", P(cur.raw), P(cur.sc), P(cur.sh), cur.act, P(wc), P(bc), P(logits), B, H, W, Cin, NC,
                 stream())
        ctx.recs, ctx.last, ctx.wc, ctx.bc = recs, cur, wc, bc
        return logits

    @staticmethod
    def backward(ctx, dlog):
        dlog = dlog.contiguous()
        cur, wc = ctx.last, ctx.wc
        B, H, W = cur.geom
        NC, Cin = wc.shape[0], wc.shape[1]
        M = cur.M
        dev = dlog.device
        gx = torch.empty(M, Cin, device=dev, dtype=torch.float32)
        rows = lib.kd_cls3x3_bwd_stat_rows(M, Cin)
        partial = torch.empty(rows * 2 * Cin, device=dev, dtype=torch.float32)
        nw = NC * Cin * 9
        dwb = torch.empty(nw + 4, device=dev, dtype=torch.float32)
        nbytes = lib.kd_cls3x3_bwd_ws_bytes(M, Cin, NC)
        ws = ops.workspace(nbytes, dev)
        lib.call("kd_cls3x3_bwd", P(dlog), P(cur.raw), P(cur.sc), P(cur.sh), cur.act, P(cur.bnc.mean), P(cur.bnc.invstd),
                 P(wc), P(gx), P(partial), P(dwb), B, H, W, Cin, NC, P(ws), nbytes, stream())
        grads, g_in = chain_backward(ctx.recs, ("G", gx, partial, rows), need_input_grad=ctx.needs_input_grad[0])
        dx = ops.nchw_from_matrix(g_in, ctx.recs[0].inp.geom) if ctx.needs_input_grad[0] else None
        dwc, dbc = gradsink.deliver_many([(wc, dwb[:nw]), (ctx.bc, dwb[nw: nw + NC])])
        return (dx, None, None, dwc, dbc,
                *grads)


def run_x4_head(x, units, cls_conv, training):
    return X4HeadFn.apply(x, list(units), training, cls_conv.weight, cls_conv.bias, *_params_of(units))


# =================================================================================================
# training scatter-max: "sorted" (cell-sorted segments, default) or "atomic" (kept for A/B and for widths the
# segmented kernels do not cover)
_SCATTER_MODE = os.environ.get("KD_SCATTER", "sorted")
# with sorted points: hand the scatter-max gradient to the last layer's backward GEMMs as per-cell tables (default) or
# as the materialised [points, C] tensor ("0": A/B and tests)
_SCATTER_TABLES = os.environ.get("KD_SCATTER_TABLES", "1") != "0"
# layer 0's gradient through moments of G0 accumulated in the layer-1 dgrad epilogue (default) or through the stored G0
_L0_MOMENTS = os.environ.get("KD_L0_MOMENTS", "1") != "0"
# one backward kernel per point-MLP layer (data + weight gradient from one read of the operands, csrc/kd_lidar_bwd.hip);
# "0": the separate dgrad / wgrad GEMM launches of round 2 (A/B and tests)
_LIDAR_FUSED_BWD = os.environ.get("KD_LIDAR_FUSED_BWD", "1") != "0"
# the whole eval-mode encoder (point MLP + scatter-max) in one kernel (csrc/kd_lidar_infer.hip); "0": layer by layer
_LIDAR_FUSED_INFER = os.environ.get("KD_LIDAR_FUSED_INFER", "1") != "0"


_sort_cache: dict = {}
_sort_sharing = False


def share_point_bins(on: bool):
    """kdrt.kd brackets teacher forward + student forward of ONE KD step with share_point_bins(True) ... (False): only
    inside that bracket may the second LiDAR encoder reuse the first one's sort of the same point tensor.  Outside it
    (plain module calls, the reference-style trainer) every call sorts for itself -- a tensor refilled through a raw
    pointer between two calls could not be told from an unchanged one.  Entering or leaving drops the entries, so
    nothing computed for one step ever serves the next."""
    global _sort_sharing
    _sort_sharing = bool(on)
    _sort_cache.clear()


def clear_step_caches():
    _sort_cache.clear()


def cell_sort(pts, B, N, H, W, rng):
    """Point ids binned by BEV grid row (kd_lidar_cell_sort) -> (row_of_point, seg_start, perm).  The frozen teacher and
    the student of ONE KD step see the same point tensor, so the second caller reuses the first one's bins; the entry
    keeps the tensor alive (its address cannot be recycled) and is dropped on any in-place write or at the next step."""
    key = (pts.data_ptr(), pts._version, B, N, H, W, tuple(float(r) for r in rng), stream())
    hit = _sort_cache.get("entry") if _sort_sharing else None
    if hit is not None and hit[0] == key:
        return hit[2]
    dev = pts.device
    row_of_point = torch.empty(B * N, device=dev, dtype=torch.int32)
    seg_start = torch.empty(B * H * W + 1, device=dev, dtype=torch.int32)
    perm = torch.empty(B * N, device=dev, dtype=torch.int32)
    nbytes = lib.kd_lidar_cell_sort_ws_bytes(B, N, H, W)
    ws = ops.workspace(nbytes, dev)
    lib.call("kd_lidar_cell_sort", P(pts), B, N, H, W, float(rng[0]), float(rng[1]), float(rng[2]), float(rng[3]),
             P(row_of_point), P(seg_start), P(perm), P(ws), nbytes, stream())
    out = (row_of_point, seg_start, perm)
    if _sort_sharing:
        _sort_cache["entry"] = (key, pts, out)
    return out


SORT_MAX_BINS = 36865            # kd_lidar_sort_points: H*W + 1 histogram bins in LDS (grids up to 192 x 192)


def sort_points(pts, B, N, H, W, rng):
    """Points stably sorted by (frame, cell) (kd_lidar_sort_points) -> (pts_sorted, row_sorted, seg_start).  Shared
    between the frozen teacher and the student of one KD step exactly like cell_sort."""
    key = ("points", pts.data_ptr(), pts._version, B, N, H, W, tuple(float(r) for r in rng), stream())
    hit = _sort_cache.get("points") if _sort_sharing else None
    if hit is not None and hit[0] == key:
        return hit[2]
    dev = pts.device
    spts = torch.empty(B * N, 4, device=dev, dtype=torch.float32)
    row_sorted = torch.empty(B * N, device=dev, dtype=torch.int32)
    seg_start = torch.empty(B * H * W + 1, device=dev, dtype=torch.int32)
    nbytes = lib.kd_lidar_sort_points_ws_bytes(B, N, H, W)
    ws = ops.workspace(nbytes, dev)
    lib.call("kd_lidar_sort_points", P(pts), B, N, H, W, float(rng[0]), float(rng[1]), float(rng[2]), float(rng[3]),
             P(spts), P(row_sorted), P(seg_start), None, P(ws), nbytes, stream())
    out = (spts, row_sorted, seg_start)
    if _sort_sharing:
        _sort_cache["points"] = (key, pts, out)
    return out


class LidarFn(torch.autograd.Function):
    """SpatialLiDAREncoder.forward_vectorized (lidar_encoder.py:57-99): point MLP on all B*N points
    (layer 0 on VALU, layers 1-2 as MFMA GEMMs with M = B*N), BEV binning, scatter-max."""

    @staticmethod
    def forward(ctx, points, units, grid_hw, rng, training, *params):
        infer, training = training == 2, training == 1
        ops.require_gpu_tensor(points, "LiDAREncoder")
        B, N, D = points.shape
        if D != 4:
            raise KDError(f"LiDAR points must be [B, N, 4], got {tuple(points.shape)}")
        pts = points.contiguous().view(B * N, 4)
        H, W = grid_hw
        if infer:
            # Inference fast path (frozen teacher): out-of-range points influence nothing in eval mode
            # (no batch statistics, never scattered), so compact them away before the point MLP.
            dev = pts.device
            cpts = ccell = None
            if _SCATTER_MODE != "sorted" or H * W + 1 > SORT_MAX_BINS:
                cpts = torch.empty(B * N, 4, device=dev, dtype=torch.float32)
                ccell = torch.empty(B * N, device=dev, dtype=torch.int32)
            if _SCATTER_MODE == "sorted" and H * W + 1 <= SORT_MAX_BINS:
                # the head of the cell-sorted point array (shared with the student of the same step) IS the compacted
                # list; the fused scatter epilogue merges neighbouring rows in registers before touching the grid
                cpts, ccell, seg_start = sort_points(pts, B, N, H, W, rng)
                counter = seg_start[B * H * W:]
            elif _SCATTER_MODE != "atomic":
                row_of_point, seg_start, perm = cell_sort(pts, B, N, H, W, rng)
                counter = seg_start[B * H * W:]
                lib.call("kd_lidar_gather_sorted", P(pts), P(perm), P(row_of_point), P(counter), P(cpts), P(ccell), B * N, stream())
            else:
                counter = torch.empty(1, device=dev, dtype=torch.int32)
                lib.call("kd_lidar_compact", P(pts), P(cpts), P(ccell), P(counter), B, N, H, W, float(rng[0]), float(rng[1]),
                         float(rng[2]), float(rng[3]), stream())
            # the row count stays on the device: the kernels read it themselves (no host sync, the CPU keeps
            # running a whole step ahead of the GPU)
            last = units[-1]
            C = last.conv.weight.shape[0]
            grid = torch.empty(B * H * W, C, device=dev, dtype=torch.float32)
            if (_LIDAR_FUSED_INFER and len(units) == 3 and [u.kind for u in units] == ["l0", "pw", "pw"]
                    and all(u.act == ACT_RELU and u.has_bias for u in units)
                    and lib.kd_lidar_mlp_scatter_infer_supported(*(u.conv.weight.shape[0] for u in units))):
                # the whole eval encoder in one kernel: no activation leaves the CU
                co = [_coeffs(u, None, 0, u.conv.weight.shape[0], 0, False, None, dev) for u in units]
                ops.lidar_mlp_scatter_infer(cpts, ccell, counter, units[0].conv, units[1].conv, units[2].conv, co, grid, B * H * W)
                return ops.nchw_from_matrix(grid, (B, H, W))
            cur = cpts
            for u in units[:-1]:
                cur, _ = unit_forward(u, cur, False, m_dev=counter, virtual=(u.kind == "l0"))
            if last.kind == "pw" and cur.virt is None and cur.bnc is not None:
                # last layer + BN + ReLU + scatter-max in one kernel: its [points, C] output is never written
                bnc = _coeffs(last, None, 0, C, 0, False, None, dev)
                ops.l2_fwd_scatter(cur, last.conv.weight, last.conv.bias, bnc, last.act, ccell, grid, B * H * W, counter)
            else:
                cur, _ = unit_forward(last, cur, False, m_dev=counter)
                lib.call("kd_lidar_scatter_max_idx_fwd", P(cur.raw), P(cur.sc), P(cur.sh), cur.act, P(ccell), P(grid), B * N, C,
                         B * H * W, P(counter), stream())
            return ops.nchw_from_matrix(grid, (B, H, W))
        mode = "atomic"
        if units[-1].conv.weight.shape[0] in (64, 128, 256) and _SCATTER_MODE != "atomic":
            mode = "points" if (_SCATTER_MODE == "sorted" and H * W + 1 <= SORT_MAX_BINS) else "ids"
        seg = None
        cur = pts
        if mode == "points":
            # The point MLP runs on the points SORTED by (frame, cell) (stable, so every order-dependent sum keeps a
            # fixed order): nothing downstream depends on the point order -- the grid is indexed by cell, weight
            # gradients are sums over points -- and each cell's rows become one contiguous range.
            cur, row_sorted, seg_start = sort_points(pts, B, N, H, W, rng)
            seg = (row_sorted, seg_start, None)
        elif mode == "ids":
            seg = cell_sort(pts, B, N, H, W, rng)      # bins of point ids; the rows stay where they are
        recs = []
        for u in units:
            cur, rec = unit_forward(u, cur, training, virtual=(u.kind == "l0"))
            recs.append(rec)
        C = cur.C
        grid = torch.empty(B * H * W, C, device=pts.device, dtype=torch.float32)
        ctx.seg = seg
        if seg is not None:
            # forward max and backward tie split over the segments: atomic-free
            lib.call("kd_lidar_seg_max_fwd", P(cur.raw), P(cur.sc), P(cur.sh), cur.act, P(seg[1]), P(seg[2]),
                     P(seg[0]) if seg[2] is None else None, P(grid), B * N, B * H * W, C, stream())
        else:
            lib.call("kd_lidar_scatter_max_fwd", P(pts), P(cur.raw), P(cur.sc), P(cur.sh), cur.act, P(grid), B, N, C, H, W,
                     float(rng[0]), float(rng[1]), float(rng[2]), float(rng[3]), stream())
        ctx.recs, ctx.last, ctx.pts, ctx.grid, ctx.shape, ctx.rng = recs, cur, pts, grid, (B, N, C, H, W), rng
        return ops.nchw_from_matrix(grid, (B, H, W))

    @staticmethod
    def backward(ctx, dout):
        dm, _ = ops.nhwc_view(dout)
        B, N, C, H, W = ctx.shape
        cur, pts, rng = ctx.last, ctx.pts, ctx.rng
        dev = dm.device
        Pn = B * N
        last = ctx.recs[-1]
        if (ctx.seg is not None and ctx.seg[2] is None and _SCATTER_TABLES and last.spec.kind == "pw"
                and last.inp.virt is None and last.inp.bnc is not None):
            # rows sorted by cell: the [points, C] gradient is never written -- per-cell tables instead
            row_sorted, seg_start, _ = ctx.seg
            rows = lib.kd_lidar_seg_share_stat_rows(B * H * W, Pn)
            partial = torch.empty(rows * 2 * C, device=dev, dtype=torch.float32)
            share = torch.empty(B * H * W, C, device=dev, dtype=torch.float32)
            cnt_ws = torch.empty(B * H * W, C, device=dev, dtype=torch.float32)
            lib.call("kd_lidar_seg_share_bwd", P(cur.raw), P(cur.sc), P(cur.sh), cur.act, P(ctx.grid), P(dm), P(cur.bnc.mean),
                     P(cur.bnc.invstd), P(seg_start), P(row_sorted), P(share), P(cnt_ws), P(partial), Pn, B * H * W, C, stream())
            grads, _ = chain_backward(ctx.recs, ("GS", (row_sorted, ctx.grid, share), partial, rows), need_input_grad=False)
            return (None, None, None, None, None, *grads)
        G = torch.empty(Pn, C, device=dev, dtype=torch.float32)
        if ctx.seg is not None:
            row_of_point, seg_start, perm = ctx.seg
            rows = lib.kd_lidar_seg_stat_rows(B * H * W)
            partial = torch.empty(rows * 2 * C, device=dev, dtype=torch.float32)
            lib.call("kd_lidar_seg_max_bwd", P(cur.raw), P(cur.sc), P(cur.sh), cur.act, P(ctx.grid), P(dm), P(cur.bnc.mean),
                     P(cur.bnc.invstd), P(seg_start), P(perm), P(row_of_point), P(G), P(partial), Pn, B * H * W, C, stream())
        else:
            rows = lib.kd_lidar_scatter_stat_rows(Pn, C)
            partial = torch.empty(rows * 2 * C, device=dev, dtype=torch.float32)
            nbytes = lib.kd_lidar_scatter_bwd_ws_bytes(B, H, W, C)
            ws = ops.workspace(nbytes, dev)
            lib.call("kd_lidar_scatter_max_bwd", P(pts), P(cur.raw), P(cur.sc), P(cur.sh), cur.act, P(ctx.grid), P(dm),
                     P(cur.bnc.mean), P(cur.bnc.invstd), P(G), P(partial), B, N, C, H, W, float(rng[0]), float(rng[1]),
                     float(rng[2]), float(rng[3]), P(ws), nbytes, stream())
        grads, _ = chain_backward(ctx.recs, ("G", G, partial, rows), need_input_grad=False)
        return (None, None, None, None, None, *grads)


def run_lidar(points, units, grid_hw, rng, training):
    return LidarFn.apply(points, list(units), tuple(grid_hw), tuple(rng), run_mode(training), *_params_of(units))
