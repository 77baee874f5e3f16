"""Dev diagnostic: is one KD backward bitwise reproducible (same process, fresh models each time)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import kd_oracle as O
from _gpu_util import build_product, load_random_state
from kdrt import gradsink
from kdrt.kd import KDStep
from kdrt.losses import kd_objective
from kdrt.optim import FusedAdamW

B, HW, N, G = 2, 64, 512, 16
cw = torch.tensor([0.4, 3.5]).cuda()


def models():
    teacher = build_product("concat", G); load_random_state(teacher, "concat", 11); teacher.eval()
    student = build_product("weighted", G); load_random_state(student, "weighted", 12); student.train()
    return teacher, student


def grads(seed, use_sink, via_step=False):
    t2, s2 = models()
    o2 = FusedAdamW(s2.parameters(), lr=1e-3, weight_decay=1e-3)
    names = [n for n, _ in s2.named_parameters()]
    batch = tuple(t.cuda() for t in O.make_inputs(B, HW, N, G, seed, pad_tail=40))
    if via_step:
        KDStep(s2, t2, o2, cw)(*batch)
    else:
        if use_sink:
            sink = gradsink.install(o2.flat)
            sink.begin_step()
        else:
            gradsink.uninstall()
        o2.zero_grad()
        with torch.no_grad():
            zt, mt = t2(*batch[:2], return_intermediates=True)
        zs, ms = s2(*batch[:2], return_intermediates=True)
        total, _ = kd_objective(zs, ms, zt, mt, batch[2], cw, 4.0, 1.0, 1.0, -1)
        total.backward()
    torch.cuda.synchronize()
    return o2.flat.grad.clone(), names, o2.flat.offsets


for seed in (100, 101):
    a, names, offs = grads(seed, True)
    b, _, _ = grads(seed, True)
    c, _, _ = grads(seed, False)
    d, _, _ = grads(seed, True, via_step=True)
    sc = a.abs().max().item()
    print(f"seed {seed}: scale {sc:.3e}  sink-vs-sink {(a-b).abs().max().item()/sc:.3e}  sink-vs-autograd {(a-c).abs().max().item()/sc:.3e}  sink-vs-KDStep {(a-d).abs().max().item()/sc:.3e}")
    for x, tag in ((b, "sink2"), (c, "autograd"), (d, "kdstep")):
        bad = [(names[i], ((a - x)[offs[i]:offs[i + 1]]).abs().max().item() / sc) for i in range(len(names))]
        bad = [t for t in bad if t[1] > 1e-6]
        print("  ", tag, "differing tensors:", len(bad), bad[:6])
