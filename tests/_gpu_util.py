"""Helpers shared by the GPU parity tests and tools/gpu_diag.py."""
import torch

import kd_oracle as O
from _util import state_template

FUSIONS = {"concat": 256, "minimal": 128, "weighted": 128}


def build_product(fusion, grid, num_classes=2, device="cuda", output_mode="same"):
    from src.models.camera_encoder import TwinLiteEncoder
    from src.models.fusion_module import CompleteSegmentationModel
    from src.models.lidar_encoder import LiDAREncoder
    cam = TwinLiteEncoder(return_multiscale=True)
    lid = LiDAREncoder(encoder_type="spatial", grid_size=(grid, grid), use_vectorized=True)
    m = CompleteSegmentationModel(cam, lid, num_classes=num_classes, fusion_type=fusion,
                                  fusion_out_channels=FUSIONS[fusion], camera_fpn_stages=["stage3", "stage4", "stage5"],
                                  camera_fpn_channels=128, output_mode=output_mode)
    return m.to(device)


def load_random_state(model, fusion, seed):
    """Name-keyed deterministic weights (same recipe the golden generator fed the reference)."""
    st = O.randomize_state(state_template(fusion), seed)
    sd = model.state_dict()
    for k in sd:
        if k.endswith("grid_tensor"):
            st[k] = sd[k].cpu()
    model.load_state_dict(st)
    return st


def oracle_run(st, fusion, images, pts, grid, training, labels=None, cw=None):
    """Oracle forward (+ CE backward when labels are given) on CPU.  Returns dict of tensors."""
    s = O.clone_state(st, requires_grad=labels is not None)
    logits, mids = O.complete_model(images, pts, s, fusion_type=fusion, grid=(grid, grid), training=training)
    out = {"logits": logits.detach(), **{k: v.detach() for k, v in mids.items()}, "state": s}
    if labels is not None:
        loss = O.weighted_ce(logits, labels, cw)
        loss.backward()
        out["loss"] = loss.detach()
        out["grads"] = {k: s[k].grad for k in O.trainable_keys(s)}
    return out


def max_err(a: torch.Tensor, b: torch.Tensor):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    d = (a - b).abs().max().item()
    return d, d / max(b.abs().max().item(), 1e-12)


def ftol(ref: torch.Tensor, base=1e-4, rel=5e-6) -> float:
    """Forward tolerance: abs 1e-4 (north_star) for O(1..10) tensors; for the eval-mode fixtures
    whose randomised running statistics blow activations up to 1e2..1e3, 1e-4 is below fp32
    resolution of the values themselves, so allow 5e-6 of the tensor's max magnitude."""
    return max(base, rel * ref.detach().abs().max().item())


def grads_match(got: torch.Tensor, want: torch.Tensor, l2_tol=1e-2, max_tol=2e-2, outlier_frac=0.04, gross_tol=8e-2):
    """Model-level gradient comparison.  ReLU / ReLU6 / scatter-max are discontinuous: when a
    pre-activation sits within fp32 rounding of a kink, the HIP forward (different summation order)
    and the CPU forward legitimately land on different sides.  Measured on this path: one element
    with |z| = 2.7e-6 flipped; in the layer where it happens ONE output channel moves by exactly that
    element's gradient while every other channel agrees to ~1e-8, and the flip then diffuses as a
    ~1e-3-relative perturbation into every upstream layer.  So: drop the worst few output channels
    (max(2, 4%)), require the rest to agree in relative L2 (1e-2) and elementwise (2e-2 of the
    tensor's max), and bound the whole tensor (outliers included) by a gross 8e-2 relative L2 so a
    wiring error can never hide.  Exactness to rounding (1e-5..1e-4) is asserted by the unit-level
    tests in test_gpu_units.py, whose problems are too small to land on a kink."""
    a = got.detach().float().cpu()
    b = want.detach().float().cpu()
    scale = max(b.abs().max().item(), 1e-3)
    floor = 1e-3 * b.numel() ** 0.5
    ra = a.reshape(a.shape[0], -1) if a.dim() > 1 else a.reshape(-1, 1)
    rb = b.reshape(ra.shape)
    err = (ra - rb).abs().max(dim=1).values
    k = max(2, int(outlier_frac * ra.shape[0]))
    keep = torch.ones(ra.shape[0], dtype=torch.bool)
    if ra.shape[0] > k:
        keep[torch.topk(err, k).indices] = False
    l2_all = ((a - b).norm() / max(b.norm().item(), floor)).item()
    l2 = ((ra[keep] - rb[keep]).norm() / max(rb[keep].norm().item(), floor)).item()
    mx = (err[keep].max().item() if keep.any() else 0.0) / scale
    if ra.shape[0] <= k:            # tiny tensors (e.g. the 2-element attention bias, a strongly cancelling sum):
        l2_tol, max_tol = 5e-2, 5e-2   # no channel can be dropped, a single flip shows at full weight
    ok = l2 <= l2_tol and mx <= max_tol and l2_all <= gross_tol
    return ok, f"relL2(inliers)={l2:.2e} max(inliers)/scale={mx:.2e} relL2(all)={l2_all:.2e} scale={scale:.2e}"
