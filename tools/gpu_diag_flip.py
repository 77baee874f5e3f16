"""Dev diagnostic: test_fusion_and_head_small[weighted] input-gradient error per data seed and GEMM arithmetic."""
import sys, os
sys.path[:0] = [os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")]
import conftest  # noqa: F401  (sys.path)
import torch
import kd_oracle as O
from _gpu_util import max_err
from kdrt import ops
from test_gpu_units import _rand_state
from src.models.fusion_module import SameResolutionSegmentationHead, WeightedFusion
import torch.nn.functional as F
for seed in range(6, 14):
    for mode in ("split", "fp32"):
        ops.set_gemm_arithmetic(mode)
        torch.manual_seed(0)
        fus, head = WeightedFusion(128, 128, 128), SameResolutionSegmentationHead(128, 2)
        stf, sth = _rand_state(fus, 41), _rand_state(head, 42)
        fus, head = fus.cuda().train(), head.cuda().train()
        g = torch.Generator().manual_seed(seed)
        cam, lid = torch.randn(2, 128, 10, 10, generator=g), torch.randn(2, 128, 10, 10, generator=g).clamp_min(0)
        cg, lg = cam.clone().cuda().requires_grad_(True), lid.clone().cuda().requires_grad_(True)
        cc, lc = cam.clone().requires_grad_(True), lid.clone().requires_grad_(True)
        z = head(fus(cg, lg))
        so = O.clone_state({**{"fusion." + k: v for k, v in stf.items()}, **{"head." + k: v for k, v in sth.items()}}, True)
        cp = O.conv1x1_block(cc, so, "fusion.cam_proj", True); lp = O.conv1x1_block(lc, so, "fusion.lidar_proj", True)
        a = F.conv2d(torch.cat([cp, lp], 1), so["fusion.attention.0.weight"], so["fusion.attention.0.bias"])
        a = F.conv2d(a.clamp_min(0), so["fusion.attention.2.weight"], so["fusion.attention.2.bias"])
        w = torch.softmax(a, 1)
        zo = O.seg_head_same(cp * w[:, 0:1] + lp * w[:, 1:2], so, "head", True)
        up = torch.randn(zo.shape, generator=g)
        (z * up.cuda()).sum().backward(); (zo * up).sum().backward()
        d = (cg.grad.cpu() - cc.grad).abs()
        bad = (d > 1e-4 * cc.grad.abs().max()).sum().item()
        print(seed, mode, "fwd %.2e" % max_err(z, zo)[1], "dcam %.2e" % max_err(cg.grad, cc.grad)[1], "dlid %.2e" % max_err(lg.grad, lc.grad)[1], "bad elems", bad, "of", d.numel())
