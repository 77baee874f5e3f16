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
    """Forward tolerance for the eval-mode fixtures with ARBITRARY running statistics (`randomize_state`): they blow
    activations up to 1e2..1e3, where abs 1e-4 is below fp32 resolution of the values themselves, so allow 5e-6 of the
    tensor's max magnitude there.  The north-star tolerance as written -- abs 1e-4, no scaling -- is asserted on well-scaled
    activations by tests/test_gpu_headline.py (eval on calibrated statistics, train mode, the benchmarked frame shape).
    One comparison needs rel = 8e-6 and passes it explicitly (weighted-fusion `pre_fusion`, max 319: the fp32 CPU oracle is
    1.5e-6 of the maximum from its own float64 evaluation there, the one-kernel eval LiDAR encoder lands at 5.9e-6 through
    the same ill-conditioned BatchNorm while its own output is 4-5e-7 from float64)."""
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


# ---------------------------------------------------------------------------------------------------------------------
# Gradient accuracy against the float64 oracle (no flip-tolerant comparison): shared by tests/test_gpu_parity.py and the
# seed scan tools/diag_fp64_seeds.py.
FP64_B, FP64_HW, FP64_N, FP64_G = 2, 64, 512, 16


def fp64_oracle_grads(student_fusion: str, objective: str, seed: int, dtype):
    """Gradients of one training step of the CPU oracle evaluated in `dtype`.  objective: "kd" = concat teacher (eval,
    state seed 11) -> student (state seed 12), CE + T^2 KL + feature MSE; "ce" = the reference's plain weighted-CE step
    (trainer.py:86-90) on the same student."""
    images, pts, labels = O.make_inputs(FP64_B, FP64_HW, FP64_N, FP64_G, seed, pad_tail=40)
    cw = torch.tensor([0.4, 3.5])
    G = FP64_G
    cast = lambda st: {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in st.items()}
    s_st = O.clone_state(cast(O.randomize_state(state_template(student_fusion), 12)), requires_grad=True)
    zs, ms = O.complete_model(images.to(dtype), pts.to(dtype), s_st, fusion_type=student_fusion, grid=(G, G), training=True)
    if objective == "kd":
        t_st = cast(O.randomize_state(state_template("concat"), 11))
        with torch.no_grad():
            zt, mt = O.complete_model(images.to(dtype), pts.to(dtype), t_st, fusion_type="concat", grid=(G, G), training=False)
        total, _ = O.kd_loss(zs, ms, zt, mt, labels, cw.to(dtype), 4.0, 1.0, 1.0)
    else:
        total = O.weighted_ce(zs, labels, cw.to(dtype))
    total.backward()
    return {k: v.grad.double() for k, v in s_st.items() if v.grad is not None}


def fp64_gpu_grads(student_fusion: str, objective: str, seed: int):
    """The product's DEFAULT step paths, so that the gradient sink and the feature-gradient routing are in a defined state
    whatever ran before: "kd" = kdrt.kd.KDStep with its default flags (fused objective, sink installed by the step), "ce" =
    the sequence of src/training/trainer.py:Trainer._step.  lr = 0: AdamW leaves the parameters alone, the gradients stay in
    the flat buffer."""
    from kdrt import gradsink
    from kdrt.kd import KDStep
    from kdrt.losses import seg_loss
    from kdrt.optim import FusedAdamW
    images, pts, labels = (t.cuda() for t in O.make_inputs(FP64_B, FP64_HW, FP64_N, FP64_G, seed, pad_tail=40))
    cw = torch.tensor([0.4, 3.5]).cuda()
    student = build_product(student_fusion, FP64_G); load_random_state(student, student_fusion, 12); student.train()
    opt = FusedAdamW(student.parameters(), lr=0.0, weight_decay=0.0)
    try:
        if objective == "kd":
            teacher = build_product("concat", FP64_G); load_random_state(teacher, "concat", 11); teacher.eval()
            KDStep(student, teacher, opt, cw, T=4.0, alpha=1.0, beta=1.0)(images, pts, labels)
        else:
            sink = gradsink.install(opt.flat, None)
            gradsink.drop_pending()
            sink.begin_step()
            opt.zero_grad()
            total, _ = seg_loss(student(images, pts), labels, cw)
            total.backward()
            assert not gradsink.pending()
        return {n: p.grad.detach().double().cpu() for n, p in student.named_parameters()}
    finally:
        gradsink.uninstall()
        gradsink.drop_pending()


def fp64_rel_errors(g64, g):
    """Sorted per-tensor relative L2 distances to the float64 gradients, over the tensors whose gradient is not noise."""
    gmax = max(v.abs().max().item() for v in g64.values())
    keys = [k for k in g64 if g64[k].norm().item() > 1e-5 * gmax * g64[k].numel() ** 0.5]
    return sorted(((g[k].double().cpu() - g64[k]).norm() / g64[k].norm()).item() for k in keys)
