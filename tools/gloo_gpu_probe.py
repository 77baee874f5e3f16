import os, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank()
torch.cuda.set_device(0)
x = torch.full((500000,), float(r + 1), device="cuda")
views = [x[:100000], x[100000:300000], x[300000:]]
for it in range(3):
    hs = [dist.all_reduce(v, async_op=True) for v in views]
    for h in hs: h.wait()
    torch.cuda.synchronize()
    print(f"rank {r} iter {it} ok sum={x[0].item()} {x[-1].item()}", flush=True)
t = torch.tensor([1.0 + r], device="cuda", dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX); print("max", t.item(), flush=True)
dist.barrier(); dist.destroy_process_group()
