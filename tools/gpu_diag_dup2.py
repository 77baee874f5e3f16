"""Dev diagnostic: where does the duplicated-batch gradient mismatch come from?  Module by module, loss = mean(out^2)."""
import os, sys
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import conftest  # noqa
import torch
from src.models.camera_encoder import TwinLiteEncoder, InvertedResidual
from src.models.lidar_encoder import LiDAREncoder
from src.models.fusion_module import Conv1x1, DWSeparableConv, WeightedFusion, SameResolutionSegmentationHead
from kdrt.losses import seg_loss, feature_mse
dev = torch.device("cuda")
def grads(make, x, reps, lossfn=None):
    torch.manual_seed(0)
    m = make().to(dev).train()
    xs = [t.repeat(reps, *([1] * (t.dim() - 1))) for t in x]
    out = m(*xs)
    outs = list(out.values()) if isinstance(out, dict) else [out]
    loss = sum((o * o).mean() for o in outs)
    loss.backward()
    return loss.item(), {n: p.grad.clone() for n, p in m.named_parameters()}
def report(name, make, x):
    (l1, g1), (l2, g2) = grads(make, x, 1), grads(make, x, 2)
    gmax = max(g.abs().max().item() for g in g1.values())
    rels = sorted((((g1[n] - g2[n]).norm() / g1[n].norm()).item(), n) for n in g1 if g1[n].norm().item() > 1e-6 * gmax * g1[n].numel() ** 0.5)
    print(f"{name:28s} loss {l1:.7f} {l2:.7f}  median rel {rels[len(rels)//2][0]:.2e}  max {rels[-1][0]:.2e} {rels[-1][1]}", flush=True)
g = torch.Generator().manual_seed(1)
report("Conv1x1 64->128", lambda: Conv1x1(64, 128), [torch.randn(2, 64, 12, 12, generator=g).to(dev)])
report("DWSeparableConv 64->128", lambda: DWSeparableConv(64, 128), [torch.randn(2, 64, 12, 12, generator=g).to(dev)])
report("InvertedResidual s1", lambda: InvertedResidual(32, 32, 1, 6), [torch.randn(2, 32, 16, 16, generator=g).to(dev)])
report("InvertedResidual s2", lambda: InvertedResidual(32, 64, 2, 6), [torch.randn(2, 32, 16, 16, generator=g).to(dev)])
report("TwinLite multiscale", lambda: TwinLiteEncoder(return_multiscale=True), [torch.rand(2, 3, 64, 64, generator=g).to(dev)])
pts = torch.randn(2, 512, 4, generator=g); pts[..., :2] *= 40
report("LiDAREncoder", lambda: LiDAREncoder(encoder_type="spatial", grid_size=(16, 16)), [pts.to(dev)])
report("WeightedFusion", lambda: WeightedFusion(128, 128, 128), [torch.randn(2, 128, 10, 10, generator=g).to(dev), torch.randn(2, 128, 10, 10, generator=g).clamp_min(0).to(dev)])
report("SameResHead", lambda: SameResolutionSegmentationHead(128, 2), [torch.randn(2, 128, 10, 10, generator=g).to(dev)])
print("---- real sizes")
report("SameResHead 64x64", lambda: SameResolutionSegmentationHead(128, 2), [torch.randn(2, 128, 64, 64, generator=g).to(dev)])
report("DWSeparableConv 64x64", lambda: DWSeparableConv(128, 64), [torch.randn(2, 128, 64, 64, generator=g).to(dev)])
report("Conv1x1 64x64", lambda: Conv1x1(128, 64), [torch.randn(2, 128, 64, 64, generator=g).to(dev)])
report("InvertedResidual s1 64x64", lambda: InvertedResidual(64, 64, 1, 6), [torch.randn(2, 64, 64, 64, generator=g).to(dev)])
report("InvertedResidual s2 128x128", lambda: InvertedResidual(32, 64, 2, 6), [torch.randn(2, 32, 128, 128, generator=g).to(dev)])
