"""One process, B = 4 KD steps launched eagerly (for rocprofv3 --kernel-trace --stats: kernel time per family at the reference's batch)."""
import os, sys, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"))
import bench
from kdrt.kd import KDStep
from kdrt.optim import FusedAdamW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
teacher, student = bench.build_models(64)
dev = torch.device("cuda", 0); teacher, student = teacher.to(dev).eval(), student.to(dev).train()
opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)
step = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5], device=dev))
images, pts, labels = bench.synth_batch(B, 80000, 256, 64, 1, dev)
for _ in range(12): step(images, pts, labels)
torch.cuda.synchronize()
