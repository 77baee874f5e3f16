"""Small-batch throughput: eager KDStep vs the hipGraph-captured step (tools; not the headline bench)."""
import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"))
import bench
from kdrt.kd import GraphedKDStep, KDStep
from kdrt.optim import FusedAdamW
for B in (4, 8, 32):
    res = {}
    for mode in ("eager", "graph"):
        teacher, student = bench.build_models(64)
        dev = torch.device("cuda", 0); teacher, student = teacher.to(dev).eval(), student.to(dev).train()
        opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)
        step = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5], device=dev))
        images, pts, labels = bench.synth_batch(B, 80000, 256, 64, 1, dev)
        run = GraphedKDStep(step, images, pts, labels) if mode == "graph" else (lambda *a, s=step: s(images, pts, labels))
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 20
        for _ in range(n): run()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        res[mode] = dt
    print(f"B={B}: eager {res['eager']*1e3:.2f} ms/step ({B/res['eager']:.0f} frames/s)   graph {res['graph']*1e3:.2f} ms/step ({B/res['graph']:.0f} frames/s)", flush=True)
