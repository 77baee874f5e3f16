"""(rows, C, pstride, producer) of every BatchNorm finalize call of one KD step at the bench shape: which statistics slabs are tall.
usage: python tools/diag_finalize_rows.py [B]"""
import os, sys, torch
ROOT = "/root/repo" if os.path.isdir("/root/repo/oracle") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"))
import bench
from kdrt import lib as L
from kdrt.kd import KDStep
from kdrt.optim import FusedAdamW
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
teacher, student = bench.build_models(64)
dev = torch.device("cuda", 0); teacher, student = teacher.to(dev).eval(), student.to(dev).train()
opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)
step = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5], device=dev))
images, pts, labels = bench.synth_batch(B, 80000, 256, 64, 1, dev)
for _ in range(2): step(images, pts, labels)
log, prev = [], [None]
LIB = L if hasattr(L, "_dll") else L.lib
orig = type(LIB).call
def spy(self, name, *a):
    if name in ("kd_bn_finalize_train", "kd_bn_bwd_finalize"):
        log.append((name, a[1], a[2], a[3], prev[0]))
    prev[0] = name
    return orig(self, name, *a)
type(LIB).call = spy
step(images, pts, labels); torch.cuda.synchronize()
for e in log: print("%-22s rows %6d C %4d pstride %4d after %s" % e)
