#!/usr/bin/env python3
"""Profiling target: the bf16-storage eval forward (BASELINE.json configs[1]) alone -- 2 warm-up + 3 forward passes of the concat model at
the bench batch (256 frames, 80 000 points, 256 x 256 images, 64 x 64 BEV grid), nothing else on the GPU.
usage: rocprofv3 ... -- python3 tools/bf16_forward_only.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from kdrt.bf16 import forward_bf16

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
B, N, HW, G = 256, 80000, 256, 64
_, model = bench.build_models(G, "concat", "concat")
model = model.to(dev).eval()
images, pts, _ = bench.synth_batch(B, N, HW, G, 1234, dev)
for _ in range(5):
    z = forward_bf16(model, images, pts)
torch.cuda.synchronize()
print("bf16 forward passes: 5 (2 warm-up + 3), logits", tuple(z.shape), "finite", bool(torch.isfinite(z).all()))
