"""Dev diagnostic: duplicated-batch gradient differences of the full KD step, absolute scale."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import conftest  # noqa
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
from kdrt.kd import KDStep
from kdrt.optim import FusedAdamW
dev = torch.device("cuda")
B, N = 2, 512
images, pts, labels = bench.synth_batch(B, N, 256, 64, 99, dev)
res = []
for reps in (1, 2):
    teacher, student = bench.build_models(64)
    teacher, student = teacher.to(dev).eval(), student.to(dev).train()
    opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)
    step = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5], device=dev))
    parts = step(images.repeat(reps, 1, 1, 1), pts.repeat(reps, 1, 1), labels.repeat(reps, 1, 1))
    res.append(({k: float(v) for k, v in parts.items() if v.numel() == 1}, {n: p.grad.detach().clone() for n, p in student.named_parameters()},
                parts["logits"][:B].clone()))
(l1, g1, z1), (l2, g2, z2) = res
print(l1); print(l2); print("logit diff", (z1 - z2).abs().max().item())
for n in list(g1)[:6] + list(g1)[-12:]:
    a, b = g1[n].flatten(), g2[n].flatten()
    print(f"{n:44s} |g| {a.norm().item():.3e} relL2 {((a-b).norm()/a.norm()).item():.2e} max|d|/max|g| {((a-b).abs().max()/a.abs().max()).item():.2e}")
