"""The KD step's fused objective (losses.kd_objective_backward: value + gradient of every loss term from one kernel pass
each, feature-MSE gradients added inside the fusion block's data-gradient GEMMs) against the autograd formulation
kd_objective(...).backward() it replaces: the same loss bits and the same gradient bits, for every student fusion block."""
import pytest
import torch

import kd_oracle as O
from _gpu_util import build_product, load_random_state

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_arith")]


def _one_step(student_fusion, fused, beta, shape, seed=12):
    from kdrt.kd import KDStep
    from kdrt.optim import FusedAdamW
    B, HW, N, G = shape
    images, pts, _ = O.make_inputs(B, HW, N, G, 4, pad_tail=40)
    labels = O.make_inputs(B, HW, N, HW // 4, 4, pad_tail=40)[2]             # labels live on the logits' grid (the camera map)
    teacher = build_product("concat", G)
    load_random_state(teacher, "concat", 11)
    student = build_product(student_fusion, G)
    load_random_state(student, student_fusion, seed)
    student.train()
    opt = FusedAdamW(student.parameters(), lr=0.0, weight_decay=0.0)         # lr 0: the flat gradient survives the step
    step = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5]).cuda(), T=4.0, alpha=1.0, beta=beta, fused_objective=fused)
    parts = step(images.cuda(), pts.cuda(), labels.cuda())
    torch.cuda.synchronize()
    return {k: parts[k].clone() for k in ("ce", "kl", "mse_cam", "mse_lidar", "total")}, opt.flat.grad.clone()


@pytest.mark.parametrize("student_fusion", ("weighted", "concat", "minimal"))
@pytest.mark.parametrize("beta", (1.0, 0.37))
def test_fused_objective_is_bit_identical_to_the_autograd_objective(student_fusion, beta):
    shape = (2, 64, 700, 16)
    (p0, g0), (p1, g1) = _one_step(student_fusion, False, beta, shape), _one_step(student_fusion, True, beta, shape)
    for k in p0:
        assert torch.equal(p0[k].view(torch.int32), p1[k].view(torch.int32)), (k, p0[k].item(), p1[k].item())
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    assert torch.equal(g0.view(torch.int32), g1.view(torch.int32)), (g0 - g1).abs().max().item()


def test_fused_objective_with_a_resized_lidar_map():
    """8 x 8 BEV grid under a 16 x 16 camera map: the deposited LiDAR-feature gradient belongs to the RESIZED map."""
    shape = (2, 64, 700, 8)
    (p0, g0), (p1, g1) = _one_step("weighted", False, 1.0, shape), _one_step("weighted", True, 1.0, shape)
    assert torch.equal(p0["total"].view(torch.int32), p1["total"].view(torch.int32))
    assert torch.equal(g0.view(torch.int32), g1.view(torch.int32)), (g0 - g1).abs().max().item()


def test_uncollected_feature_gradient_fails_loudly():
    from kdrt import KDError
    from kdrt.losses import kd_objective_backward
    B, HW, N, G = 2, 64, 700, 16
    images, pts, labels = O.make_inputs(B, HW, N, G, 4, pad_tail=40)
    teacher = build_product("concat", G).eval()
    load_random_state(teacher, "concat", 11)
    student = build_product("weighted", G)
    load_random_state(student, "weighted", 12)
    student.train()
    with torch.no_grad():
        zt, mt = teacher(images.cuda(), pts.cuda(), return_intermediates=True)
    zs, ms = student(images.cuda(), pts.cuda(), return_intermediates=True)
    ms = dict(ms)
    ms["camera_feat"] = ms["camera_feat"] * 1.0          # a copy no fusion block ever consumed: nobody will collect its gradient
    with pytest.raises(KDError):
        kd_objective_backward(zs, ms, zt, mt, labels.cuda(), torch.tensor([0.4, 3.5]).cuda())


def test_per_step_transpose_cache_matches_per_weight_transposes():
    """Three AdamW steps with the once-per-step batched W^T refresh against three steps with a kd_transpose launch per
    weight: identical parameters (a stale W^T after an optimiser update would show from step 2 on)."""
    from kdrt import ops
    from kdrt.kd import KDStep
    from kdrt.optim import FusedAdamW
    B, HW, N, G = 2, 64, 700, 16
    images, pts, labels = O.make_inputs(B, HW, N, G, 4, pad_tail=40)
    images, pts, labels = images.cuda(), pts.cuda(), labels.cuda()
    out = {}
    for cache in (False, True):
        ops.TRANSPOSES.clear()
        ops.TRANSPOSES.enabled = cache
        try:
            teacher = build_product("concat", G)
            load_random_state(teacher, "concat", 11)
            student = build_product("weighted", G)
            load_random_state(student, "weighted", 12)
            student.train()
            opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)
            step = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5]).cuda())
            for _ in range(3):
                step(images, pts, labels)
            torch.cuda.synchronize()
            out[cache] = opt.flat.data.clone() if hasattr(opt.flat, "data") else torch.cat([p.detach().flatten() for p in student.parameters()])
            if cache:
                assert len(ops.TRANSPOSES.entries) >= 10 and ops.TRANSPOSES.table is not None
        finally:
            ops.TRANSPOSES.enabled = True
            ops.TRANSPOSES.clear()
    assert torch.equal(out[False].view(torch.int32), out[True].view(torch.int32))


def test_label_grid_that_does_not_match_the_logits_is_refused():
    """The loss kernels index the target as the logits' [B, H, W]; a smaller label grid must raise, not read out of bounds."""
    from kdrt import KDError
    from kdrt.losses import seg_loss
    z = torch.randn(2, 2, 16, 16, device="cuda", requires_grad=True)
    with pytest.raises(KDError):
        seg_loss(z, torch.zeros(2, 8, 8, dtype=torch.int64, device="cuda"))
    with pytest.raises(KDError):
        seg_loss(z, torch.zeros(2, 16, 16, dtype=torch.int64))


def test_return_intermediates_by_name():
    """`return_intermediates` also takes a collection of names (the KD step's three maps): the same tensors, bit for bit, as the
    reference's bool form, and the concat block then skips materialising `pre_fusion`."""
    B, HW, N, G = 2, 64, 700, 16
    images, pts, _ = O.make_inputs(B, HW, N, G, 4, pad_tail=40)
    m = build_product("concat", G)
    load_random_state(m, "concat", 11)
    m.eval()
    with torch.no_grad():
        z_all, mids_all = m(images.cuda(), pts.cuda(), return_intermediates=True)
        z_kd, mids_kd = m(images.cuda(), pts.cuda(), return_intermediates=("camera_feat", "lidar_feat", "logits"))
    assert set(mids_all) == {"camera_feat", "lidar_feat", "pre_fusion", "post_fusion", "logits"} and mids_all["pre_fusion"].numel() > 0
    assert set(mids_kd) == {"camera_feat", "lidar_feat", "logits"}
    assert torch.equal(z_all, z_kd)
    for k in mids_kd:
        assert torch.equal(mids_all[k], mids_kd[k]), k
    with pytest.raises(ValueError):
        m(images.cuda(), pts.cuda(), return_intermediates=("camera_feat", "no_such_map"))


@pytest.mark.parametrize("student_fusion", ("weighted", "concat"))
def test_feature_gradient_routing_matches_autograd_accumulation(student_fusion, monkeypatch):
    """Stage-3 / stage-4 maps feed the next encoder stage AND the FPN.  With routing, the FPN's backward deposits their gradients
    for the next stage's data-gradient kernel (and folds d(stage-5 output) into the stage-4 deposit: stage 5 is a residual block
    over stage 4's map) instead of autograd summing two gradient tensors per map.  Same losses; gradients equal up to the
    association of one three-term sum (a + d5) + f4  vs  a + (f4 + d5)."""
    from kdrt import gradsink, units
    shape = (2, 64, 700, 16)
    res = {}
    for routing in (False, True):
        monkeypatch.setattr(units, "_GRAD_ROUTING", routing)
        seen = []
        real = gradsink.deposit
        monkeypatch.setattr(gradsink, "deposit", lambda *a, **k: (seen.append(k.get("folded_residual") is not None), real(*a, **k))[1])
        res[routing] = _one_step(student_fusion, True, 1.0, shape)
        # the two feature-MSE deposits always; with routing also the stage-3 and the stage-4 map (the latter with the fold)
        assert len(seen) == (4 if routing else 2) and sum(seen) == (1 if routing else 0), seen
        monkeypatch.setattr(gradsink, "deposit", real)
    (p0, g0), (p1, g1) = res[False], res[True]
    for k in p0:
        assert torch.equal(p0[k].view(torch.int32), p1[k].view(torch.int32)), k
    assert (g0 - g1).abs().max().item() <= 2e-6 * g0.abs().max().item(), (g0 - g1).abs().max().item()
    assert gradsink.pending() == 0
