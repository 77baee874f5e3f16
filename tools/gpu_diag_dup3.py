"""Dev diagnostic: InvertedResidual stride 2 -- GPU gradients vs the CPU oracle at batch 2 and at the duplicated batch 4."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import conftest  # noqa
import torch
import kd_oracle as O
from src.models.camera_encoder import InvertedResidual
dev = torch.device("cuda")
g = torch.Generator().manual_seed(1)
x0 = torch.randn(2, 32, 16, 16, generator=g)
for stride, cout in ((2, 64), (1, 32)):
  for reps in (1, 2, 4):
    torch.manual_seed(0)
    m = InvertedResidual(32, cout, stride, 6)
    st = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = x0.repeat(reps, 1, 1, 1)
    mg = m.to(dev).train()
    out = mg(x.to(dev)); (out * out).mean().backward()
    so = O.clone_state({("." + k): v for k, v in st.items()}, requires_grad=True)
    oo = O.inverted_residual(x, so, "", 32, cout, stride, 6, True); (oo * oo).mean().backward()
    worst = max((((p.grad.cpu() - so["." + n].grad).norm() / so["." + n].grad.norm()).item(), n) for n, p in mg.named_parameters())
    print("stride", stride, "batch", 2 * reps, "fwd err", (out.cpu() - oo).abs().max().item(), "worst grad rel vs oracle", worst, flush=True)
for stride, cout, reps in ((2, 64, 1), (2, 64, 2)):
    print("---- detail, stride", stride, "batch", 2 * reps)
    torch.manual_seed(0)
    m = InvertedResidual(32, cout, stride, 6)
    st = {k: v.detach().clone() for k, v in m.state_dict().items()}
    x = x0.repeat(reps, 1, 1, 1)
    mg = m.to(dev).train()
    out = mg(x.to(dev)); (out * out).mean().backward()
    so = O.clone_state({("." + k): v for k, v in st.items()}, requires_grad=True)
    oo = O.inverted_residual(x, so, "", 32, cout, stride, 6, True); (oo * oo).mean().backward()
    for n, p in mg.named_parameters():
        a, b = p.grad.cpu().flatten(), so["." + n].grad.flatten()
        d = (a - b).abs()
        print(f"{n:16s} |g| {b.norm().item():.3e} relL2 {((a-b).norm()/b.norm()).item():.2e}  max|d| {d.max().item():.3e}  n_bad {(d > 1e-4 * b.abs().max()).sum().item()}/{a.numel()}")
