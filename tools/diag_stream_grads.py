"""Dev diagnostic: KD-step gradients and BatchNorm buffers with the streaming GEMM kernels on vs off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import kd_oracle as O
from _gpu_util import build_product, load_random_state
from kdrt.lib import lib
from kdrt.losses import kd_objective

B, HW, N, G = 2, 64, 512, 16
images, pts, labels = (t.cuda() for t in O.make_inputs(B, HW, N, G, 4, pad_tail=40))
cw = torch.tensor([0.4, 3.5]).cuda()


from kdrt import ops
_rec = []
_orig = ops.bn_finalize_train
def _spy(partial, rows, C, count, bn, bnc, update_running=True, pstride=None):
    _orig(partial, rows, C, count, bn, bnc, update_running, pstride)
    ps = pstride or C
    p = partial[: rows * 2 * ps].view(rows, 2, ps)[:, :, :C].double().sum(0)
    _rec[-1].append((rows, C, count, bnc.mean.clone(), bnc.invstd.clone(), p))
ops.bn_finalize_train = _spy
import kdrt.units as U


def run(mode):
    _rec.append([])
    lib.kd_set_gemm_stream(mode)
    teacher = build_product("concat", G); load_random_state(teacher, "concat", 11); teacher.eval()
    student = build_product("weighted", G); load_random_state(student, "weighted", 12); student.train()
    with torch.no_grad():
        zt, mt = teacher(images, pts, return_intermediates=True)
    zs, ms = student(images, pts, return_intermediates=True)
    total, _ = kd_objective(zs, ms, zt, mt, labels, cw, 4.0, 1.0, 1.0, -1)
    total.backward()
    torch.cuda.synchronize()
    return ({n: p.grad.clone() for n, p in student.named_parameters()}, {k: v.clone() for k, v in student.state_dict().items() if "running" in k},
            zs.detach().clone(), {k: v.detach().clone() for k, v in ms.items()})


g0, b0, z0, m0 = run(0)
g1, b1, z1, m1 = run(2)
print("logits max diff", (z0 - z1).abs().max().item())
for k in m0:
    print("mid", k, (m0[k] - m1[k]).abs().max().item())
for k in b0:
    d = (b0[k] - b1[k]).abs().max().item()
    if d > 1e-7 * max(1.0, b0[k].abs().max().item()):
        print("buffer", k, d, b0[k].abs().max().item())
rows = []
for n in g0:
    rel = ((g0[n] - g1[n]).norm() / g0[n].norm().clamp_min(1e-20)).item()
    rows.append((rel, n))
for rel, n in sorted(rows, reverse=True)[:25]:
    print(f"grad {n:55s} rel {rel:.3e}")

for i, (a, b) in enumerate(zip(_rec[0], _rec[1])):
    dm = (a[3] - b[3]).abs().max().item(); di = ((a[4] - b[4]).abs() / a[4].abs()).max().item()
    ds = ((a[5] - b[5]).abs() / a[5].abs().clamp_min(1e-30)).max().item()
    print(f"bn call {i:2d} rows {a[0]:5d}->{b[0]:5d} C {a[1]:4d} count {a[2]:7d} | mean diff {dm:.2e} invstd rel {di:.2e} partial-sum rel {ds:.2e}")
for n in ("camera_encoder.stage5.conv.7.bias", "camera_fpn.post.net.1.bias", "camera_fpn.laterals.stage5.conv.1.bias", "fusion.cam_proj.conv.1.bias", "camera_fpn.post.net.4.bias"):
    d = (g0[n] - g1[n]).abs()
    top = torch.topk(d, 4)
    print(n, "total abs diff", d.sum().item(), "top channels", [(int(i), float(v)) for v, i in zip(top.values, top.indices)], "grad abs max", g0[n].abs().max().item())
