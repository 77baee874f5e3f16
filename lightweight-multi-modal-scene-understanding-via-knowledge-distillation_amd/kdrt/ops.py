"""Thin tensor-level wrappers over the C ABI (include/kd_hip.h).

Everything here is pointer plumbing: take torch tensors that already live in HBM, hand raw
device pointers + sizes + the current HIP stream to libkd_hip.so.  No arithmetic happens in Python.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .lib import KDError, lib, require_gpu_tensor

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2
BN_EPS, BN_MOMENTUM = 1e-5, 0.1

_ws = {}
_ws_captured = set()      # devices whose current workspace a hipGraph capture has seen
_ws_keep = []             # retired workspaces that captured graphs still point into


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def workspace(nbytes: int, device) -> torch.Tensor:
    """One grow-only scratch buffer per device.  All kernels are stream-ordered, and every C entry
    point consumes its workspace before it returns control to the stream, so sharing is safe.
    A buffer that was handed out during a hipGraph capture has its address baked into that graph: when a larger
    request later replaces it, it is retired to `_ws_keep` instead of being freed (a replay would otherwise scribble
    scratch data over whatever tensor the allocator put there next)."""
    key = (device.type, device.index)
    buf = _ws.get(key)
    capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    if buf is None or buf.numel() < nbytes:
        if buf is not None and key in _ws_captured:
            _ws_keep.append(buf)
            _ws_captured.discard(key)
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = buf
    if capturing:
        _ws_captured.add(key)
    return buf


def P(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def ld(t: torch.Tensor) -> int:
    """row stride (floats) of a 2-D [M, C] view with unit inner stride"""
    if t.dim() != 2 or t.stride(1) != 1:
        raise KDError(f"expected a [M, C] view with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0)


class BNC:
    """Per-channel BatchNorm coefficients of one layer for one forward pass."""
    __slots__ = ("buf", "mean", "invstd", "scale", "shift")

    def __init__(self, C: int, device, buf: Optional[torch.Tensor] = None):
        self.buf = torch.empty(4, C, device=device, dtype=torch.float32) if buf is None else buf
        self.mean, self.invstd, self.scale, self.shift = self.buf[0], self.buf[1], self.buf[2], self.buf[3]


class Operand:
    """A possibly-deferred NHWC activation: value = act(raw*sc+sh), or raw itself when bnc is None.
    raw is a [M, C] view (row stride = ld); geom = (B, H, W) with M = B*H*W.
    A VIRTUAL operand (raw is None, virt = (points [M,4], w0 [C,4,1], b0 [C])) is the LiDAR layer-0 output, which is
    never written: consumers recompute it from the 16-byte point (kd_lidar_l1_*)."""
    __slots__ = ("raw", "bnc", "act", "geom", "virt")

    def __init__(self, raw: Optional[torch.Tensor], geom, bnc: Optional[BNC] = None, act: int = ACT_NONE, virt=None):
        self.raw, self.bnc, self.act, self.geom, self.virt = raw, bnc, act, geom, virt

    @property
    def M(self):
        return self.raw.shape[0] if self.raw is not None else self.virt[0].shape[0]

    @property
    def C(self):
        return self.raw.shape[1] if self.raw is not None else self.virt[1].shape[0]

    @property
    def sc(self):
        return None if self.bnc is None else self.bnc.scale

    @property
    def sh(self):
        return None if self.bnc is None else self.bnc.shift


# ---------------------------------------------------------------------------------------------
def nhwc_view(x: torch.Tensor) -> Tuple[torch.Tensor, tuple]:
    """NCHW-shaped tensor -> ([M, C] NHWC matrix view, (B, H, W)).  Channels-last tensors are used
    in place; anything else is re-laid-out once by the allocator's copy (layout plumbing)."""
    require_gpu_tensor(x, "kdrt.ops")
    if x.dtype != torch.float32:
        raise KDError(f"fp32 tensors expected, got {x.dtype}")
    B, C, H, W = x.shape
    xp = x.permute(0, 2, 3, 1)
    if not xp.is_contiguous():
        xp = xp.contiguous()
    return xp.reshape(B * H * W, C), (B, H, W)


def nchw_from_matrix(m: torch.Tensor, geom) -> torch.Tensor:
    """[M, C] NHWC matrix -> NCHW-shaped (channels-last strided) view, no copy."""
    B, H, W = geom
    return m.view(B, H, W, m.shape[1]).permute(0, 3, 1, 2)


# ---------------------------------------------------------------------------------------------
# Epochs of raw-pointer writes (invisible to torch's tensor versions) for caches keyed on parameter / buffer contents:
# per BatchNorm module (running statistics rewritten by kd_bn_finalize_train), per optimiser (FusedAdamW tags the
# parameters it owns with itself) and process-wide (hipGraph replays, broadcasts).
GLOBAL_EPOCH = [0]


def bump_global_epoch():
    GLOBAL_EPOCH[0] += 1


def bn_epoch(bn) -> int:
    return getattr(bn, "_kd_stats_epoch", 0)


def owner_epoch(p) -> int:
    o = getattr(p, "_kd_owner", None)
    return 0 if o is None else o.epoch


def bn_finalize_train(partial, rows, C, count, bn, bnc: BNC, update_running=True, pstride=None):
    if update_running:
        bn._kd_stats_epoch = bn_epoch(bn) + 1
    lib.call("kd_bn_finalize_train", P(partial), rows, C, pstride or C, count, P(bn.weight), P(bn.bias), float(bn.eps),
             bn.momentum if bn.momentum is not None else BN_MOMENTUM,
             P(bn.running_mean) if update_running else None, P(bn.running_var) if update_running else None,
             P(bn.num_batches_tracked) if update_running else None,
             P(bnc.mean), P(bnc.invstd), P(bnc.scale), P(bnc.shift), stream())


def bn_eval_coeffs(bn, bnc: BNC):
    C = bn.running_mean.numel()
    lib.call("kd_bn_eval_coeffs", P(bn.weight), P(bn.bias), P(bn.running_mean), P(bn.running_var), bn.eps, C,
             P(bnc.mean), P(bnc.invstd), P(bnc.scale), P(bnc.shift), stream())


def set_gemm_arithmetic(mode: str) -> str:
    """'fp32': exact-fp32 MFMA products (v_mfma_f32_32x32x2_f32); 'split': every fp32 operand cut exactly into three
    bf16 pieces, six leading piece products accumulated in fp32 on the bf16 matrix pipe (fp32-grade: <= 2^-23 |x||y|
    per product).  Process-wide; returns the previous mode."""
    if mode not in ("fp32", "split"):
        raise KDError(f"unknown GEMM arithmetic {mode!r} (expected 'fp32' or 'split')")
    return "split" if lib.kd_set_gemm_split(1 if mode == "split" else 0) else "fp32"


def get_gemm_arithmetic() -> str:
    prev = lib.kd_set_gemm_split(0)
    lib.kd_set_gemm_split(prev)
    return "split" if prev else "fp32"


import os as _os
import weakref as _weakref
set_gemm_arithmetic(_os.environ.get("KD_GEMM", "split"))     # default: the faster fp32-grade arithmetic; KD_GEMM=fp32 selects exact products

PROFILE = None      # bench.py sets this to a list to time GEMM launches with HIP events on the launch stream


def _prof_begin():
    if PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _prof_end(e0, kind, flops, nbytes, m_dev=None, M=None):
    """flops / nbytes are for M rows; with a device-side row count (compacted LiDAR points) the reader scales them by
    min(m_dev, M) / M after the run (prof_scaled) -- no host sync here."""
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        PROFILE.append((kind, flops, nbytes, e0, e1, m_dev, M))


def prof_scaled(rec):
    """(kind, flops, bytes, seconds, M) of one PROFILE record, row-count corrected (call after a device synchronise);
    M is the nominal row count of the launch (lets the reader tell point-MLP launches from image-grid ones)."""
    kind, flops, nbytes, e0, e1, m_dev, M = rec
    f = 1.0 if m_dev is None else min(int(m_dev.item()), M) / float(M)
    return kind, flops * f, nbytes * f, e0.elapsed_time(e1) * 1e-3, M


def pw_gemm(A, W, C_out, *, M, K, N, A2=None, pro=0, pro_act=0, p=(None,) * 5, bias=None, addend=None, epi=0,
            X=None, esc=None, esh=None, emean=None, einv=None, epi_act=0, partial=None, partial_rows=0, m_dev=None):
    """partial_rows: the row count `partial` was sized for (lib.kd_pwconv_stat_rows_for); the library refuses the launch
    when the kernel form it selects now would write a different number of rows."""
    e0 = _prof_begin()
    lib.call("kd_pwconv_gemm", P(A), ld(A), P(A2), ld(A2) if A2 is not None else 0, pro, pro_act,
             P(p[0]), P(p[1]), P(p[2]), P(p[3]), P(p[4]), P(W), P(bias), P(C_out), ld(C_out),
             P(addend), ld(addend) if addend is not None else 0, epi, P(X), ld(X) if X is not None else 0,
             P(esc), P(esh), P(emean), P(einv), epi_act, P(partial), partial_rows, M, K, N, P(m_dev), stream())
    # algorithmic bytes: every operand tensor of the launch read or written exactly once
    _prof_end(e0, "pw_gemm", 2.0 * M * N * K,
              4.0 * (M * K * (2 if pro == 2 else 1) + M * N * (1 + (epi == 2) + (addend is not None)) + N * K), m_dev, M)


def l1_fwd(op: Operand, W, C_out, *, bias, epi, partial, partial_rows=0, m_dev=None):
    """LiDAR layer 1 forward over a virtual layer-0 operand (pts -> layer 0 -> BN+act recomputed on load)."""
    pts, w0, b0 = op.virt
    M, K, N = pts.shape[0], w0.shape[0], W.shape[0]
    e0 = _prof_begin()
    lib.call("kd_lidar_l1_fwd", P(pts), P(w0), P(b0), P(op.sc), P(op.sh), op.act, P(W), P(bias), P(C_out), ld(C_out), epi,
             P(partial), partial_rows, M, K, N, P(m_dev), stream())
    _prof_end(e0, "pw_gemm", 2.0 * M * N * K, 4.0 * (M * 4 + M * N + N * K), m_dev, M)


def l2_fwd_scatter(op: Operand, W, bias, bnc2: BNC, act2, cell_idx, grid, ncells, m_dev):
    """Eval: last point-MLP layer + BN + ReLU + BEV scatter-max in one kernel (the layer output is never written)."""
    M, K, N = op.M, op.C, W.shape[0]
    e0 = _prof_begin()
    lib.call("kd_lidar_l2_fwd_scatter", P(op.raw), ld(op.raw), P(op.sc), P(op.sh), op.act, P(W), P(bias), P(bnc2.scale),
             P(bnc2.shift), act2, P(cell_idx), P(grid), ncells, M, K, N, P(m_dev), stream())
    _prof_end(e0, "pw_gemm", 2.0 * M * N * K, 4.0 * (M * K + M * 1 + N * K), m_dev, M)     # reads A + cell index; no output tensor


def lidar_mlp_scatter_infer(pts, cell_idx, m_dev, l0, l1, l2, coeffs, grid, ncells):
    """Eval: the whole point MLP + BEV scatter-max in one kernel (csrc/kd_lidar_infer.hip).  l0 / l1 / l2: the conv modules,
    coeffs: their eval BNC triples."""
    b0, b1, b2 = coeffs
    Pn, C0, C1, C2 = pts.shape[0], l0.weight.shape[0], l1.weight.shape[0], l2.weight.shape[0]
    e0 = _prof_begin()
    lib.call("kd_lidar_mlp_scatter_infer", P(pts), P(cell_idx), P(m_dev), P(l0.weight), P(l0.bias), P(b0.scale), P(b0.shift),
             P(l1.weight), P(l1.bias), P(b1.scale), P(b1.shift), P(l2.weight), P(l2.bias), P(b2.scale), P(b2.shift), P(grid), ncells,
             Pn, C0, C1, C2, stream())
    # both GEMMs of the launch; algorithmic traffic: the 16-byte point + its cell index in, nothing out but the scatter
    _prof_end(e0, "lidar_infer", 2.0 * Pn * (C0 * C1 + C1 * C2), 4.0 * (Pn * 5 + C0 * C1 + C1 * C2), m_dev, Pn)


def l1_dgrad(t, y, Wt, gin, *, op: Operand, al, be, ga, msc, msh, mact, partial, partial_rows, moments=None):
    """moments ([4, K0] tensor, optional): receives sum_m G0 * point; with it `gin` may be None (G0 is never written)."""
    pts, w0, b0 = op.virt
    M, N1, K0 = pts.shape[0], y.shape[1], w0.shape[0]
    nbytes, ws = 0, None
    if moments is not None:
        nbytes = lib.kd_lidar_l1_dgrad_ws_bytes(M, K0)
        ws = workspace(nbytes, y.device)
    e0 = _prof_begin()
    lib.call("kd_lidar_l1_dgrad", P(t), ld(t), P(y), ld(y), P(al), P(be), P(ga), P(msc), P(msh), mact, P(Wt), P(gin),
             ld(gin) if gin is not None else K0, P(pts), P(w0), P(b0), P(op.sc), P(op.sh), P(op.bnc.mean), P(op.bnc.invstd),
             op.act, P(partial), partial_rows, P(moments), P(ws), nbytes, M, N1, K0, stream())
    _prof_end(e0, "pw_gemm", 2.0 * M * N1 * K0, 4.0 * (2 * M * N1 + (M * K0 if gin is not None else 0) + M * 4 + N1 * K0), None, M)


def l1_bwd(t, y, Wt, dW, *, op: Operand, al, be, ga, partial, partial_rows, moments):
    """Point-MLP layer 1, training backward in ONE kernel (csrc/kd_lidar_bwd.hip): weight gradient + BatchNorm-0 backward sums
    + the G0 * point moments from one read and one split of (G1, Y1); layer 0 recomputed from the points."""
    pts, w0, b0 = op.virt
    M, N1, K0 = pts.shape[0], y.shape[1], w0.shape[0]
    nbytes = lib.kd_lidar_l1_bwd_ws_bytes(M, N1, K0)
    ws = workspace(nbytes, y.device)
    e0 = _prof_begin()
    lib.call("kd_lidar_l1_bwd", P(t), ld(t), P(y), ld(y), P(al), P(be), P(ga), P(Wt), P(pts), P(w0), P(b0), P(op.sc), P(op.sh),
             P(op.bnc.mean), P(op.bnc.invstd), op.act, P(partial), partial_rows, P(moments), P(dW), M, N1, K0, P(ws), nbytes, stream())
    _prof_end(e0, "lidar_bwd", 2.0 * 2.0 * M * N1 * K0, 4.0 * (2 * M * N1 + M * 4 + 2 * N1 * K0), None, M)


def l2_dgrad(tables, out_op: Operand, Wt, gin, *, inp: Operand, al, be, ga, partial, partial_rows):
    """Data gradient of the last point-MLP layer with the scatter-max gradient rebuilt from (row_sorted, grid, share)."""
    rows_t, grid, share = tables
    y = out_op.raw
    M, N = y.shape
    K = inp.C
    t0 = _prof_begin()
    lib.call("kd_lidar_l2_dgrad", P(y), ld(y), P(rows_t), P(grid), P(share), P(al), P(be), P(ga), P(out_op.sc), P(out_op.sh),
             out_op.act, P(Wt), P(gin), ld(gin), P(inp.raw), ld(inp.raw), P(inp.sc), P(inp.sh), P(inp.bnc.mean), P(inp.bnc.invstd),
             inp.act, P(partial), partial_rows, M, N, K, stream())
    # algorithmic traffic: Y2 in, Y1 in (epilogue mask), G1 out; the tables are cache-resident (one row per ~12 points)
    _prof_end(t0, "pw_gemm", 2.0 * M * N * K, 4.0 * (M * N + 2 * M * K + M + N * K), None, M)


def l2_bwd(tables, out_op: Operand, Wt, gin, dW, *, inp: Operand, al, be, ga, partial, partial_rows):
    """Last point-MLP layer, training backward in ONE kernel (csrc/kd_lidar_bwd.hip): data gradient + BatchNorm-backward sums
    + weight gradient from one read and one split of Y2 and Y1."""
    rows_t, grid, share = tables
    y = out_op.raw
    M, N = y.shape
    K = inp.C
    nbytes = lib.kd_lidar_l2_bwd_ws_bytes(M, N, K)
    ws = workspace(nbytes, y.device)
    t0 = _prof_begin()
    lib.call("kd_lidar_l2_bwd", P(y), ld(y), P(rows_t), P(grid), P(share), P(al), P(be), P(ga), P(out_op.sc), P(out_op.sh), out_op.act,
             P(Wt), P(gin), ld(gin), P(inp.raw), ld(inp.raw), P(inp.sc), P(inp.sh), P(inp.bnc.mean), P(inp.bnc.invstd), inp.act,
             P(partial), partial_rows, P(dW), M, N, K, P(ws), nbytes, stream())
    # algorithmic traffic of THIS launch: Y2 in, Y1 in, G1 out (each once); both GEMMs' FLOPs.  Its own record kind: the launch is
    # bound by the matrix pipe + conversion VALU, not by HBM, and must not blur the roofline of the 1x1-conv GEMM family
    _prof_end(t0, "lidar_bwd", 2.0 * 2.0 * M * N * K, 4.0 * (M * N + 2 * M * K + M + 2 * N * K), None, M)


def l2_wgrad(tables, out_op: Operand, dW, *, inp: Operand, al, be, ga):
    rows_t, grid, share = tables
    y = out_op.raw
    M, N = y.shape
    K = inp.C
    nbytes = lib.kd_pwconv_wgrad_ws_bytes(M, N, K)
    ws = workspace(nbytes, y.device)
    t0 = _prof_begin()
    lib.call("kd_lidar_l2_wgrad", P(y), ld(y), P(rows_t), P(grid), P(share), P(al), P(be), P(ga), P(out_op.sc), P(out_op.sh),
             out_op.act, P(inp.raw), ld(inp.raw), P(inp.sc), P(inp.sh), inp.act, P(dW), M, N, K, P(ws), nbytes, stream())
    _prof_end(t0, "pw_wgrad", 2.0 * M * N * K, 4.0 * (M * N + M * K + M + N * K), None, M)


def l1_wgrad(t, y, dW, *, op: Operand, al, be, ga, msc, msh, mact):
    pts, w0, b0 = op.virt
    M, N, K = pts.shape[0], y.shape[1], w0.shape[0]
    nbytes = lib.kd_pwconv_wgrad_ws_bytes(M, N, K)
    ws = workspace(nbytes, t.device)
    e0 = _prof_begin()
    lib.call("kd_lidar_l1_wgrad", P(t), ld(t), P(y), ld(y), mact, P(al), P(be), P(ga), P(msc), P(msh), P(pts), P(w0), P(b0),
             P(op.sc), P(op.sh), op.act, P(dW), M, N, K, P(ws), nbytes, stream())
    _prof_end(e0, "pw_wgrad", 2.0 * M * N * K, 4.0 * (2 * M * N + M * 4 + N * K), None, M)


def pw_wgrad(D, A, dW, *, M, N, K, X=None, d_mode=0, d_act=0, al=None, be=None, ga=None, msc=None, msh=None,
             a_mode=0, a_act=0, asc=None, ash=None):
    nbytes = lib.kd_pwconv_wgrad_ws_bytes(M, N, K)
    ws = workspace(nbytes, D.device)
    e0 = _prof_begin()
    lib.call("kd_pwconv_wgrad", P(D), ld(D), P(X), ld(X) if X is not None else 0, d_mode, d_act, P(al), P(be), P(ga),
             P(msc), P(msh), P(A), ld(A), a_mode, a_act, P(asc), P(ash), P(dW), M, N, K, P(ws), nbytes, stream())
    _prof_end(e0, "pw_wgrad", 2.0 * M * N * K, 4.0 * (M * N * (2 if d_mode == 2 else 1) + M * K + N * K), None, M)


class TransposeCache:
    """W^T of every weight the data-gradient GEMMs need, refreshed by ONE launch per step instead of one per weight.

    `transpose(w)` registers a weight the first time it sees it (and transposes it on its own that time).  `refresh()` -- called
    by the gradient sink at the start of a step, when the parameters are final for that step -- re-transposes all registered
    weights with kd_transpose_batch and stamps them with the current parameter epochs; `transpose(w)` then hands out the
    cached W^T as long as the stamp still matches (any optimiser update, graph replay or broadcast moves an epoch, after
    which the per-weight launch is used again until the next refresh).  The W^T buffers are persistent, so a captured
    hipGraph that contains the refresh launch and their readers replays correctly."""

    def __init__(self):
        self.entries = {}          # (data_ptr, R, C) -> [weight view, W^T buffer, stamp, weakref to the owning parameter]
        self.table = None
        self.nblocks = 0
        self.enabled = _os.environ.get("KD_TRANSPOSE_CACHE", "1") != "0"

    @staticmethod
    def _stamp(e):
        o = e[3]()
        return None if o is None else (GLOBAL_EPOCH[0], owner_epoch(o), o._version)

    def get(self, w2d, owner):
        if not self.enabled or owner is None:
            return None
        key = (w2d.data_ptr(), w2d.shape[0], w2d.shape[1])
        e = self.entries.get(key)
        if e is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            R, Cc = w2d.shape
            self.entries[key] = [w2d.detach(), torch.empty(Cc, R, device=w2d.device, dtype=torch.float32), None, _weakref.ref(owner)]
            self.table = None
            return None
        return e[1] if (e[2] is not None and e[2] == self._stamp(e)) else None

    def refresh(self):
        if not self.enabled or not self.entries:
            return
        dead = [k for k, e in self.entries.items() if e[3]() is None]          # models that are gone: stop refreshing their weights
        if dead and not torch.cuda.is_current_stream_capturing():
            for k in dead:
                del self.entries[k]
            self.table = None
            if not self.entries:
                return
        ents = list(self.entries.values())
        if self.table is None:
            rows, blk = [], 0
            for w, wt, _, _ in ents:
                rows.append([w.data_ptr(), wt.data_ptr(), w.shape[0], w.shape[1], blk])
                blk += (w.numel() + 255) // 256
            if torch.cuda.is_current_stream_capturing():
                return                                     # (a new weight showed up after warm-up: leave it to the per-weight launch)
            self.table = torch.tensor(rows, dtype=torch.int64).to(ents[0][0].device)
            self.nblocks = blk
        lib.call("kd_transpose_batch", P(self.table), len(ents), self.nblocks, stream())
        for e in ents:
            e[2] = self._stamp(e)

    def clear(self):
        self.entries.clear()
        self.table = None


TRANSPOSES = TransposeCache()


def transpose(w2d: torch.Tensor, owner: Optional[torch.Tensor] = None) -> torch.Tensor:
    """W^T as a fresh tensor, or -- when `owner` names the parameter `w2d` is a view of -- from the per-step cache."""
    cached = TRANSPOSES.get(w2d, owner)
    if cached is not None:
        return cached
    R, Cc = w2d.shape
    out = torch.empty(Cc, R, device=w2d.device, dtype=torch.float32)
    lib.call("kd_transpose", P(w2d), P(out), R, Cc, stream())
    return out


def bn_act_apply(x, sc, sh, act, out, res=None):
    M, C = x.shape
    lib.call("kd_bn_act_apply", P(x), ld(x), P(sc), P(sh), act, P(res), ld(res) if res is not None else 0, P(out),
             ld(out), M, C, stream())
    return out


def materialize(op: Operand, res: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
                res_op: Optional[Operand] = None):
    """act(bn(raw)) (+ res) as a plain [M, C] tensor; `res_op`: the residual as a deferred operand (applied on load)."""
    if out is None:
        out = torch.empty(op.M, op.C, device=op.raw.device, dtype=torch.float32)
    if res_op is not None:
        if res is not None:
            raise KDError("materialize: give the residual as a tensor or as an operand, not both")
        if res_op.bnc is None:
            return bn_act_apply(op.raw, op.sc, op.sh, op.act, out, res_op.raw)
        lib.call("kd_bn_act_apply_res", P(op.raw), ld(op.raw), P(op.sc), P(op.sh), op.act, P(res_op.raw), ld(res_op.raw),
                 P(res_op.sc), P(res_op.sh), res_op.act, P(out), ld(out), op.M, op.C, stream())
        return out
    return bn_act_apply(op.raw, op.sc, op.sh, op.act, out, res)


def bn_bwd_reduce(D, op: Operand):
    """(sum G, sum G*xhat) partial slab for G = D * act'(.) over the deferred operand `op`."""
    M, C = op.M, op.C
    rows = lib.kd_rowwise_stat_rows(M, C)
    partial = torch.empty(rows * 2 * C, device=D.device, dtype=torch.float32)
    lib.call("kd_bn_bwd_reduce", P(D), ld(D), P(op.raw), ld(op.raw), P(op.sc), P(op.sh), op.act,
             P(op.bnc.mean), P(op.bnc.invstd), P(partial), M, C, stream())
    return partial, rows


def bn_bwd_finalize(partial, rows, C, count, gamma, bnc: BNC, training: bool, want_dbias=False, pstride=None,
                    dgamma=None, dbeta=None, dbias=None):
    """-> (dgamma, dbeta, abg[3,C], dbias|None); dgamma/dbeta/dbias may be caller-provided destinations."""
    dev = partial.device
    out = torch.empty(6, C, device=dev, dtype=torch.float32)
    dgamma = out[0] if dgamma is None else dgamma
    dbeta = out[1] if dbeta is None else dbeta
    if want_dbias and dbias is None:
        dbias = out[5]
    lib.call("kd_bn_bwd_finalize", P(partial), rows, C, pstride or C, count, P(gamma), P(bnc.mean), P(bnc.invstd), int(training),
             P(dgamma), P(dbeta), P(out[2]), P(out[3]), P(out[4]), P(dbias) if want_dbias else None, stream())
    return dgamma, dbeta, out[2:5], (dbias if want_dbias else None)
