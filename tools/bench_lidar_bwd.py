#!/usr/bin/env python3
"""Dev tool: the last point-MLP layer's training backward at the benchmarked size (256 frames x 80 000 points): the two
round-2 launches (kd_lidar_l2_dgrad + kd_lidar_l2_wgrad) against the one-kernel form (kd_lidar_l2_bwd), HIP-event timed.
Also the target of the rocprofv3 --pmc passes (python3 directly after `--`).  usage: bench_lidar_bwd.py [frames] [reps] [which]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"))
import torch
from kdrt import ops
from kdrt.ops import lib, P, stream

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
which = sys.argv[3] if len(sys.argv) > 3 else "both"
N = 80000
M, C, cells = frames * N, 128, frames * 4096
g = torch.Generator(device="cuda").manual_seed(1)
Y2 = torch.randn(M, C, device="cuda", generator=g)
Y1 = torch.randn(M, C, device="cuda", generator=g)
# sorted rows: ~62 % of the points valid, ~12 per cell, the rest (-1) at the tail -- like kd_lidar_sort_points leaves them
nvalid = int(0.62 * M)
rows = torch.full((M,), -1, dtype=torch.int32, device="cuda")
rows[:nvalid] = torch.sort(torch.randint(0, cells, (nvalid,), device="cuda", generator=g, dtype=torch.int32)).values
v = lambda: torch.rand(C, device="cuda", generator=g) + 0.5
sc2, sh2, al, be, ga, sc1, sh1, mean1, inv1 = v(), v() - 1.0, v(), v() * 0.1, v() * 0.1, v(), v() - 1.0, v(), v()
grid = torch.rand(cells, C, device="cuda", generator=g)
share = torch.randn(cells, C, device="cuda", generator=g)
W = torch.randn(C, C, device="cuda", generator=g) / C ** 0.5
Wt = ops.transpose(W)
G1 = torch.empty(M, C, device="cuda")
dW = torch.empty(C, C, device="cuda")
rd = lib.kd_lidar_l2_dgrad_stat_rows(M, C, C)
pd = torch.empty(rd * 2 * C, device="cuda")
nbw = lib.kd_pwconv_wgrad_ws_bytes(M, C, C)
rf = lib.kd_lidar_l2_bwd_stat_rows(M)
pf = torch.empty(rf * 2 * C, device="cuda")
nbf = lib.kd_lidar_l2_bwd_ws_bytes(M, C, C)
ws = torch.empty(max(nbw, nbf), dtype=torch.uint8, device="cuda")


def separate():
    lib.call("kd_lidar_l2_dgrad", P(Y2), C, P(rows), P(grid), P(share), P(al), P(be), P(ga), P(sc2), P(sh2), 1, P(Wt), P(G1), C,
             P(Y1), C, P(sc1), P(sh1), P(mean1), P(inv1), 1, P(pd), rd, M, C, C, stream())
    lib.call("kd_lidar_l2_wgrad", P(Y2), C, P(rows), P(grid), P(share), P(al), P(be), P(ga), P(sc2), P(sh2), 1, P(Y1), C, P(sc1),
             P(sh1), 1, P(dW), M, C, C, P(ws), nbw, stream())


def fused():
    lib.call("kd_lidar_l2_bwd", P(Y2), C, P(rows), P(grid), P(share), P(al), P(be), P(ga), P(sc2), P(sh2), 1, P(Wt), P(G1), C, P(Y1), C,
             P(sc1), P(sh1), P(mean1), P(inv1), 1, P(pf), rf, P(dW), M, C, C, P(ws), nbf, stream())


def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


gb = 4.0 * (3 * M * C) / 1e9
if which in ("both", "separate"):
    t = timeit(separate)
    print(f"separate kernels (dgrad + wgrad): {t:7.3f} ms  ({4.0 * 5 * M * C / 1e9 / t:.0f} GB/s of their 5 tensor passes)", flush=True)
if which in ("both", "fused"):
    t = timeit(fused)
    print(f"one kernel                      : {t:7.3f} ms  ({gb / t * 1e3:.0f} GB/s of its 3 tensor passes, {2 * 2.0 * M * C * C / t / 1e9:.0f} TFLOP/s fp32-equivalent)", flush=True)
if hasattr(lib._dll, "kd_lb_dbg_read"):          # instrumented build (KD_HIP_LIB=tools/dbg/lb/libkd_hip.so)
    import ctypes
    buf = (ctypes.c_ulonglong * 16)()
    lib._dll.kd_lb_dbg_read(buf, 1)
    fused(); torch.cuda.synchronize()
    lib._dll.kd_lb_dbg_read(buf, 1)
    its = max(buf[8], 1)                        # iterations summed over role-A waves
    names = ["V: rows fetch + epilogue", "V: convert + LDS stores", "V: issue loads", "V: barrier", "M: both k-loops", "M: stage stores", "-", "M: barrier"]
    print("per-iteration cycles (s_memtime, averaged over the waves of the role):")
    for i, n in enumerate(names):
        print(f"  {n:28s} {buf[i] / its:8.0f}")
    print(f"  V total {sum(buf[0:4]) / its:.0f}   M total {sum(buf[4:8]) / its:.0f}")
