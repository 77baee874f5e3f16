#!/usr/bin/env python3
"""Dev tool: kd_stem_conv_fwd alone at the benchmarked size (256 frames, 3 x 256 x 256 -> 128 x 128 x 32), HIP-event timed, with and without the
BatchNorm-statistics slab; KD_HIP_LIB selects a probe build (tools/dbg/stem/libkd_p<bits>.so, -DKD_STEM_PROBE).  usage: bench_stem.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"))
import torch
from kdrt.ops import lib, P, stream
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.rand(B, 3, 256, 256, device="cuda")
w = torch.randn(32, 3, 3, 3, device="cuda")
y = torch.empty(B * 128 * 128, 32, device="cuda")
rows = lib.kd_stem_stat_rows(B * 128 * 128)
part = torch.empty(rows * 2 * 32, device="cuda")
def t(p):
    for _ in range(2):
        lib.call("kd_stem_conv_fwd", P(x), P(w), P(y), p, B, 3, 256, 256, 32, stream())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.call("kd_stem_conv_fwd", P(x), P(w), P(y), p, B, 3, 256, 256, 32, stream())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3
print(os.environ.get("KD_HIP_LIB", "default lib"), f"eval {t(None):7.1f} us   train(+stats) {t(P(part)):7.1f} us")
