"""Dev diagnostic: accuracy of the KD-step gradients against an fp64 evaluation of the oracle -- GPU (both arithmetics)
next to the fp32 CPU oracle.  Shows how much of the model-level gradient mismatch is conditioning (BatchNorm-backward
cancellation), i.e. present in ANY fp32 evaluation."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import conftest  # noqa
import torch
import kd_oracle as O
from _gpu_util import build_product, load_random_state
from _util import state_template
from kdrt import ops
from kdrt.losses import kd_objective
B, HW, N, G = (2, 64, 512, 16) if len(sys.argv) < 2 else (2, 256, 80000, 64)
images, pts, labels = O.make_inputs(B, HW, N, G, 4, pad_tail=40)
cw = torch.tensor([0.4, 3.5])
def oracle(dtype):
    t_st = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in O.randomize_state(state_template("concat"), 11).items()}
    s_st = O.clone_state({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in O.randomize_state(state_template("weighted"), 12).items()}, requires_grad=True)
    im, pt = images.to(dtype), pts.to(dtype)
    with torch.no_grad():
        zt, mt = O.complete_model(im, pt, t_st, fusion_type="concat", grid=(G, G), training=False)
    zs, ms = O.complete_model(im, pt, s_st, fusion_type="weighted", grid=(G, G), training=True)
    total, _ = O.kd_loss(zs, ms, zt, mt, labels, cw.to(dtype), 4.0, 1.0, 1.0)
    total.backward()
    return total.item(), {k: v.grad.double() for k, v in s_st.items() if v.grad is not None}
l64, g64 = oracle(torch.float64)
l32, g32 = oracle(torch.float32)
def summary(name, g):
    gmax = max(v.abs().max().item() for v in g64.values())
    rels = sorted(((g[k].double().cpu() - g64[k]).norm() / g64[k].norm()).item() for k in g64 if g64[k].norm().item() > 1e-5 * gmax * g64[k].numel() ** 0.5)
    print(f"{name:22s} relL2 vs fp64: median {rels[len(rels)//2]:.2e}  90% {rels[int(len(rels)*0.9)]:.2e}  max {rels[-1]:.2e}  ({len(rels)} tensors)", flush=True)
print("loss fp64", l64, "fp32", l32)
summary("CPU oracle fp32", g32)
for arith in ("split", "fp32"):
    ops.set_gemm_arithmetic(arith)
    teacher = build_product("concat", G); load_random_state(teacher, "concat", 11); teacher.eval()
    student = build_product("weighted", G); load_random_state(student, "weighted", 12); student.train()
    with torch.no_grad():
        zt, mt = teacher(images.cuda(), pts.cuda(), return_intermediates=True)
    zs, ms = student(images.cuda(), pts.cuda(), return_intermediates=True)
    total, _ = kd_objective(zs, ms, zt, mt, labels.cuda(), cw.cuda(), 4.0, 1.0, 1.0, -1)
    total.backward()
    summary("GPU " + arith, {n: p.grad for n, p in student.named_parameters()})
