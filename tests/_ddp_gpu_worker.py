"""Worker of tests/test_gpu_ddp_one_gpu.py: one rank of a 2-rank data-parallel KD job whose ranks all sit on cuda:0 and
talk over gloo (a 1-GPU box has no second device for RCCL; the code path -- HIP backward kernels writing into the flat
gradient buffer, GradSink.done -> BucketedAllReduce.notify, async all-reduce per bucket, 1/world inside fused AdamW --
is the product's).  Each rank then replays BOTH replicas alone, without any communication, and compares."""
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd"), os.path.join(ROOT, "oracle"), HERE):
    sys.path.insert(0, p)

import kd_oracle as O  # noqa: E402
from _gpu_util import build_product, load_random_state  # noqa: E402
from kdrt import gradsink  # noqa: E402
from kdrt.ddp import BucketedAllReduce, broadcast_module  # noqa: E402
from kdrt.kd import KDStep  # noqa: E402
from kdrt.optim import FusedAdamW  # noqa: E402

B, HW, N, G = 2, 64, 512, 16


def models():
    teacher = build_product("concat", G); load_random_state(teacher, "concat", 11); teacher.eval()
    student = build_product("weighted", G); load_random_state(student, "weighted", 12); student.train()
    return teacher, student


def batch(rank):
    return tuple(t.cuda() for t in O.make_inputs(B, HW, N, G, 100 + rank, pad_tail=40))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    cw = torch.tensor([0.4, 3.5]).cuda()
    teacher, student = models()
    if rank != 0:                                   # prove the broadcast: other ranks start from garbage
        with torch.no_grad():
            for p in student.parameters():
                p.add_(1.0)
    broadcast_module(student)
    opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)
    names = [n for n, p in student.named_parameters() if p.requires_grad]
    red = BucketedAllReduce(opt.flat, names, n_buckets=3)
    orders, launch = [], red._launch

    def logged(b):
        orders[-1].append(b)
        if os.environ.get("KD_DDP_SYNC") == "1":   # debugging aid: blocking reduce after a device sync
            torch.cuda.synchronize()
            dist.all_reduce(red.views[b])
            torch.cuda.synchronize()
            return
        launch(b)
    red._launch = logged
    notes = {}
    notify = red.notify

    def counted(p):
        i = red.index_of[id(p)]
        notes.setdefault(i, []).append(sum(len(o) for o in orders))      # buckets launched so far when this gradient landed
        notify(p)
    red.notify = counted
    step = KDStep(student, teacher, opt, cw, reducer=red)
    orders.append([])
    parts = step(*batch(rank))
    torch.cuda.synchronize()
    after1 = opt.flat.data.clone()
    summed = opt.flat.grad.clone()                 # the all-reduced (summed) gradients of step 1
    bn1 = {k: v.clone() for k, v in student.state_dict().items() if k.endswith("running_mean")}
    orders.append([])
    step(*batch(rank))                             # a second step: buckets re-arm, nothing deadlocks
    torch.cuda.synchronize()

    # ---- the same job without communication: each replica alone, gradients summed by hand -------------------------
    grads, bns = [], []
    for r in range(world):
        t2, s2 = models()
        o2 = FusedAdamW(s2.parameters(), lr=1e-3, weight_decay=1e-3)
        k2 = KDStep(s2, t2, o2, cw)
        sink = k2.sink
        gradsink.active = sink
        sink.begin_step(); o2.zero_grad()
        from kdrt.losses import kd_objective
        with torch.no_grad():
            zt, mt = t2(*batch(r)[:2], return_intermediates=True)
        zs, ms = s2(*batch(r)[:2], return_intermediates=True)
        total, _ = kd_objective(zs, ms, zt, mt, batch(r)[2], cw, 4.0, 1.0, 1.0, -1)
        total.backward()
        torch.cuda.synchronize()
        grads.append(o2.flat.grad.clone())
        bns.append({k: v.clone() for k, v in s2.state_dict().items() if k.endswith("running_mean")})
    want_sum = grads[0] + grads[1]
    t3, s3 = models()
    o3 = FusedAdamW(s3.parameters(), lr=1e-3, weight_decay=1e-3)
    o3.flat.grad.copy_(want_sum)
    o3.grad_scale = 1.0 / world
    o3.step()
    torch.cuda.synchronize()
    scale = max(want_sum.abs().max().item(), 1e-12)
    res = {
        "rank": rank, "orders": orders, "total": float(parts["total"]),
        "grad_err": (summed - want_sum).abs().max().item() / scale,
        "param_err": (after1 - o3.flat.data).abs().max().item(),
        "err_vs_own_only": (summed - grads[rank]).abs().max().item() / scale,
        "err_vs_other_only": (summed - grads[1 - rank]).abs().max().item() / scale,
        "bucket_err": [((summed - want_sum)[opt.flat.offsets[a]:opt.flat.offsets[e]]).abs().max().item() / scale for a, e in red.spans],
        "bn_own": max((bn1[k] - bns[rank][k]).abs().max().item() for k in bn1),
        "bn_other": max((bn1[k] - bns[1 - rank][k]).abs().max().item() for k in bn1),
        "grad_scale": opt.grad_scale,
    }
    if os.environ.get("KD_DDP_DEBUG") == "1":
        other = summed.clone()
        dist.broadcast(other, src=0)
        res["summed_same_on_both_ranks"] = bool(torch.equal(other, summed))
        rows = []
        for i, nm in enumerate(names):
            a, e = opt.flat.offsets[i], opt.flat.offsets[i] + opt.flat.params[i].numel()
            rows.append((nm, red.bucket_of[i], notes.get(i), (summed[a:e] - want_sum[a:e]).abs().max().item(),
                         (summed[a:e] - grads[rank][a:e]).abs().max().item(), (summed[a:e] - grads[1 - rank][a:e]).abs().max().item(),
                         want_sum[a:e].abs().max().item()))
        res["rows"] = rows
    # every rank holds the same parameters after the steps
    mine = opt.flat.data.clone()
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    res["ranks_agree"] = bool(torch.equal(mine, ref))
    with open(os.path.join(os.environ["KD_DDP_OUT"], f"rank{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
