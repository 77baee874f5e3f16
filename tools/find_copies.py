#!/usr/bin/env python3
"""Dev tool: which Python call sites issue ATen ops (copies, adds, fills) inside one KD step."""
import collections, os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from kdrt.kd import KDStep
from kdrt.optim import FusedAdamW
from torch.utils._python_dispatch import TorchDispatchMode

dev = torch.device("cuda:0")
teacher, student = bench.build_models(64, "concat", "weighted")
teacher, student = teacher.to(dev).eval(), student.to(dev).train()
opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)
step = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5], device=dev))
images, pts, labels = bench.synth_batch(4, 8000, 256, 64, 1, dev, 40.0)
for _ in range(2):
    step(images, pts, labels)
torch.cuda.synchronize()
sites = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy", "add", "clone", "fill", "zero", "mul", "contiguous", "empty")):
            fr = [f for f in traceback.extract_stack() if ("kdrt" in f.filename or "/src/" in f.filename) and "find_copies" not in f.filename]
            where = f"{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}" if fr else "autograd engine / other"
            sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    step(images, pts, labels)
torch.cuda.synchronize()
for (name, where), n in sites.most_common(45):
    print(f"{n:4d}  {name:40s} {where}")
