"""Dev diagnostic: gradient agreement between a batch and two copies of it, at several sizes / arithmetics."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import conftest  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from kdrt import ops
from kdrt.kd import KDStep
from kdrt.optim import FusedAdamW
dev = torch.device("cuda")
for arith in ("split", "fp32"):
    ops.set_gemm_arithmetic(arith)
    for (B, N) in ((2, 512), (4, 5000), (16, 80000), (64, 80000)):
        images, pts, labels = bench.synth_batch(B, N, 256, 64, 99, dev)
        res = []
        for reps in (1, 2):
            teacher, student = bench.build_models(64)
            teacher, student = teacher.to(dev).eval(), student.to(dev).train()
            opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)
            step = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5], device=dev))
            parts = step(images.repeat(reps, 1, 1, 1), pts.repeat(reps, 1, 1), labels.repeat(reps, 1, 1))
            res.append((float(parts["total"]), {n: p.grad.detach().clone() for n, p in student.named_parameters()}))
            del teacher, student, opt, step
        (l1, g1), (l2, g2) = res
        gmax = max(g.abs().max().item() for g in g1.values())
        rels = sorted((((g1[n] - g2[n]).norm() / g1[n].norm()).item(), n) for n in g1 if g1[n].norm().item() > 1e-5 * gmax * g1[n].numel() ** 0.5)
        print(arith, B, N, "loss", l1, l2, "median rel", rels[len(rels) // 2][0], "max", rels[-1], flush=True)
