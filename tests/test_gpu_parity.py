"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded
inputs, and against the golden vectors captured from the reference.  Tolerances: logits / loss
abs 1e-4 (BASELINE.json north_star), class indices bit-exact wherever the oracle's logit margin
exceeds the numeric tolerance, gradients 2e-4 relative to each tensor's max magnitude."""
import numpy as np
import pytest
import torch

import kd_oracle as O
from _gpu_util import FUSIONS, build_product, ftol, grads_match, load_random_state, max_err, oracle_run
from _util import golden, state_template

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("gemm_arith")]
B, HW, N, G = 2, 64, 512, 16
LOGIT_TOL = 1e-4
GRAD_RTOL = 2e-4


@pytest.mark.parametrize("fusion", list(FUSIONS))
def test_eval_forward_vs_oracle_and_golden(fusion):
    model = build_product(fusion, G)
    st = load_random_state(model, fusion, 0)
    model.eval()
    images, pts, _ = O.make_inputs(B, HW, N, G, 0, pad_tail=40)
    with torch.no_grad():
        logits, mids = model(images.cuda(), pts.cuda(), return_intermediates=True)
        ms = model.camera_encoder(images.cuda())
    ref = oracle_run(st, fusion, images, pts, G, training=False)
    gd = golden(f"model_{fusion}_s0.npz")
    for k in ("camera_feat", "lidar_feat", "pre_fusion", "post_fusion"):
        rel = 8e-6 if fusion == "weighted" and k in ("pre_fusion", "post_fusion") else 5e-6      # see _gpu_util.ftol
        assert max_err(mids[k], ref[k])[0] < ftol(ref[k], rel=rel), k
        assert max_err(mids[k], torch.from_numpy(gd["eval_" + k]))[0] < ftol(ref[k], rel=rel), k
    for k, v in ms.items():
        assert max_err(v, torch.from_numpy(gd["eval_" + k]))[0] < ftol(v), k
    assert max_err(logits, ref["logits"])[0] < ftol(ref["logits"])
    assert max_err(logits, torch.from_numpy(gd["eval_logits"]))[0] < ftol(ref["logits"])
    # bit-exact class indices wherever the margin is above the numeric tolerance
    zr = ref["logits"]
    safe = (zr[:, 0] - zr[:, 1]).abs() > 4 * ftol(zr)
    from kdrt.losses import confusion
    _, pred = confusion(logits, torch.zeros(B, G, G, dtype=torch.int64, device="cuda"))
    assert torch.equal(pred.cpu()[safe], zr.argmax(1)[safe])
    assert safe.float().mean() > 0.99


@pytest.mark.parametrize("fusion", list(FUSIONS))
def test_train_step_vs_oracle(fusion):
    from kdrt.losses import confusion, seg_loss
    model = build_product(fusion, G)
    st = load_random_state(model, fusion, 1)
    model.train()
    images, pts, labels = O.make_inputs(B, HW, N, G, 1, pad_tail=40)
    cw = torch.tensor([0.4, 3.5])
    logits, mids = model(images.cuda(), pts.cuda(), return_intermediates=True)
    ce, _ = seg_loss(logits, labels.cuda(), cw.cuda())
    ce.backward()
    ref = oracle_run(st, fusion, images, pts, G, training=True, labels=labels, cw=cw)
    gd = golden(f"model_{fusion}_s1.npz")
    assert max_err(logits, ref["logits"])[0] < LOGIT_TOL
    assert max_err(logits, torch.from_numpy(gd["train_logits"]))[0] < LOGIT_TOL
    assert abs(ce.item() - ref["loss"].item()) < LOGIT_TOL
    assert abs(ce.item() - float(gd["train_loss"])) < LOGIT_TOL
    bad = []
    for name, p in model.named_parameters():
        want = ref["grads"][name]
        assert p.grad is not None, name
        ok, msg = grads_match(p.grad, want)
        if not ok:
            bad.append((name, msg))
    assert not bad, bad
    # BN buffers after one training forward
    sd = model.state_dict()
    for k, v in ref["state"].items():
        if k.endswith(("running_mean", "running_var")):
            assert max_err(sd[k], v)[0] < 1e-4 * max(1.0, v.abs().max().item()), k
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v) == 1, k
    # class indices / confusion matrix: bit-exact on every pixel whose oracle margin exceeds the numeric tolerance (the
    # others are handed to the kernel as ignore_index on both sides, so a close call cannot waive the whole comparison)
    zr = ref["logits"]
    safe = (zr[:, 0] - zr[:, 1]).abs() > 4 * LOGIT_TOL
    assert safe.float().mean() > 0.95
    lab_safe = torch.where(safe, labels, torch.full_like(labels, -1))
    conf, pred = confusion(logits, lab_safe.cuda())
    assert np.array_equal(conf.cpu().numpy(), O.confusion_matrix(zr, lab_safe).numpy())
    assert torch.equal(pred.cpu()[safe], zr.argmax(1)[safe])


def test_lidar_edge_cases_bit_exact_cells():
    from kdrt.lib import lib
    from kdrt.ops import P, stream
    gd = golden("lidar_edges.npz")
    for case in ("edge", "outside", "nan"):
        pts = torch.from_numpy(gd[f"{case}_points"]).cuda()
        Bn, Nn = pts.shape[:2]
        cell = torch.empty(Bn * Nn, dtype=torch.int32, device="cuda")
        lib.call("kd_lidar_bev_index", P(pts.view(-1, 4)), P(cell), Bn * Nn, 16, 16, -50.0, 50.0, -50.0, 50.0, stream())
        cell = cell.cpu().numpy().reshape(Bn, Nn)
        valid = gd[f"{case}_valid"]
        assert np.array_equal(cell >= 0, valid)
        flat = gd[f"{case}_flat"] % 256
        assert np.array_equal(cell[valid], flat[valid])


@pytest.mark.parametrize("case", ("edge", "outside"))
def test_lidar_encoder_golden(case):
    from src.models.lidar_encoder import SpatialLiDAREncoder
    from _util import state_template
    gd = golden("lidar_edges.npz")
    full = state_template("weighted")
    pre = "lidar_encoder.encoder."
    st = O.randomize_state({k[len(pre):]: v for k, v in full.items() if k.startswith(pre)}, 3)
    pts = torch.from_numpy(gd[f"{case}_points"])
    for mode in ("eval", "train"):
        enc = SpatialLiDAREncoder(grid_size=(16, 16))
        st2 = dict(st); st2["grid_tensor"] = enc.state_dict()["grid_tensor"]
        enc.load_state_dict(st2)
        enc = enc.cuda().train(mode == "train")
        y = enc(pts.cuda())
        assert max_err(y, torch.from_numpy(gd[f"{case}_{mode}_out"]))[0] < LOGIT_TOL
        if case == "outside":
            assert float(y.abs().max()) == 0.0
        if mode == "train" and case == "edge":
            up = torch.from_numpy(gd[f"{case}_upstream"]).cuda()
            (y * up).sum().backward()
            for n_, p_ in enc.named_parameters():
                want = torch.from_numpy(gd[f"{case}_grad_{n_}"])
                if n_.endswith(".bias") and n_.split(".")[1] in ("0", "3", "6"):
                    # conv bias in front of a train-mode BN: the true gradient is 0, both sides hold rounding noise
                    assert p_.grad.abs().max().item() < 1e-4, n_
                    continue
                d, _ = max_err(p_.grad, want)
                assert d < 5e-4 * max(want.abs().max().item(), 1e-3), (n_, d)


def test_full_size_eval_golden():
    gd = golden("full_weighted_eval.npz")
    model = build_product("weighted", 64)
    load_random_state(model, "weighted", 2)
    model.eval()
    images, pts, _ = O.make_inputs(2, 256, 5000, 64, 2, pad_tail=300)
    with torch.no_grad():
        logits = model(images.cuda(), pts.cuda())
    want = torch.from_numpy(gd["logits"])
    assert max_err(logits, want)[0] < ftol(want)
    safe = (want[:, 0] - want[:, 1]).abs() > 4 * ftol(want)
    assert torch.equal(logits.argmax(1).cpu()[safe], torch.from_numpy(gd["argmax"])[safe])


def test_kd_step_vs_oracle_and_adamw():
    from kdrt.losses import kd_objective
    from kdrt.optim import FusedAdamW
    teacher = build_product("concat", G)
    t_st = load_random_state(teacher, "concat", 11)
    teacher.eval()
    student = build_product("weighted", G)
    s_st = load_random_state(student, "weighted", 12)
    student.train()
    opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)
    images, pts, labels = O.make_inputs(B, HW, N, G, 4, pad_tail=40)
    cw = torch.tensor([0.4, 3.5])
    with torch.no_grad():
        zt, mt = teacher(images.cuda(), pts.cuda(), return_intermediates=True)
    opt.zero_grad()
    zs, ms = student(images.cuda(), pts.cuda(), return_intermediates=True)
    total, parts = kd_objective(zs, ms, zt, mt, labels.cuda(), cw.cuda(), T=4.0, alpha=1.0, beta=1.0)
    total.backward()
    # oracle
    to = O.clone_state(t_st)
    so = O.clone_state(s_st, requires_grad=True)
    with torch.no_grad():
        zt_o, mt_o = O.complete_model(images, pts, to, fusion_type="concat", grid=(G, G), training=False)
    zs_o, ms_o = O.complete_model(images, pts, so, fusion_type="weighted", grid=(G, G), training=True)
    total_o, parts_o = O.kd_loss(zs_o, ms_o, zt_o, mt_o, labels, cw, T=4.0, alpha=1.0, beta=1.0)
    total_o.backward()
    gd = golden("kd_step.npz")
    assert abs(total.item() - total_o.item()) < 2e-4
    assert abs(total.item() - float(gd["total"])) < 2e-4
    for k in ("ce", "kl", "mse_cam", "mse_lidar"):
        assert abs(parts[k].item() - parts_o[k].item()) < 1e-4, k
    bad = []
    for name, p in student.named_parameters():
        want = so[name].grad
        ok, msg = grads_match(p.grad, want)
        if not ok:
            bad.append((name, msg))
    assert not bad, bad
    # one fused AdamW step vs the oracle's AdamW fed with the SAME (HIP) gradients: isolates the kernel
    keys = O.trainable_keys(so)
    named = dict(student.named_parameters())
    params = [named[k].detach().cpu().clone() for k in keys]
    grads = [named[k].grad.detach().cpu().clone() for k in keys]
    O.adamw_step(params, grads, [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params], step=1,
                 lr=1e-3, weight_decay=1e-3)
    opt.step()
    for k, want, g in zip(keys, params, grads):
        tiny = g.abs() < 1e-6           # Adam turns rounding-noise gradients into +-lr: skip those elements
        d = (named[k].detach().cpu() - want).abs()
        assert d[~tiny].max().item() < 2e-6 * max(1.0, want.abs().max().item()) if (~tiny).any() else True, k


def _fp64_table():
    """{"kd/weighted": [seeds], ...}: input seeds whose batch kept every pre-activation clear of a ReLU / ReLU6 / max kink in
    EVERY evaluation (CPU fp32 oracle, GPU in both GEMM arithmetics, streaming mode 0 / 1 / 2) when tools/diag_fp64_seeds.py
    scanned them on an MI355X (scan output in profiles/r03_fp64_seed_scan.txt, list in tests/golden/fp64_clean_seeds.json)."""
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fp64_clean_seeds.json")))["clean"]


def _fp64_check(objective, fusion, seed):
    from _gpu_util import fp64_gpu_grads, fp64_oracle_grads, fp64_rel_errors
    g64 = fp64_oracle_grads(fusion, objective, seed, torch.float64)
    cpu = fp64_rel_errors(g64, fp64_oracle_grads(fusion, objective, seed, torch.float32))
    gpu = fp64_rel_errors(g64, fp64_gpu_grads(fusion, objective, seed))
    return cpu, gpu


@pytest.mark.parametrize("case", sorted(_fp64_table()))
def test_gradients_against_fp64_oracle(case):
    """Gradient accuracy without the flip-tolerant comparison: the oracle evaluated in float64 is the ground truth, and the
    GPU gradients of the whole step (KD: concat teacher -> concat / minimal / weighted student; CE: the reference's plain
    step) must be as close to it as the fp32 CPU oracle is (measured: both ~2.5e-6 median, < 1e-5 max) -- a systematic
    error of a few 1e-4 (round 2's statistics-slab row-count bug) is 100x over the line.

    Three scanned seeds per case and ALL THREE must pass (round 3 let one of three fail; ADVICE r3: a data-dependent kernel
    bug -- a tail chunk, a tile boundary -- shows in one batch exactly like a ReLU kink would).  Whether a batch sits on a
    kink depends on the last bit of every sum before it, so a change of a summation order can move a committed batch onto
    one: the scan (tools/diag_fp64_seeds.py) is re-run on the GPU whenever kernels change and the table re-committed."""
    objective, fusion = case.split("/")
    seeds = _fp64_table()[case]
    assert len(seeds) >= 3, "tests/golden/fp64_clean_seeds.json: three scanned seeds per case"
    med = lambda v: v[len(v) // 2]
    verdicts = []
    for seed in seeds[:3]:
        cpu, gpu = _fp64_check(objective, fusion, seed)
        assert len(gpu) > 75
        verdicts.append((seed, med(gpu) <= max(3 * med(cpu), 1e-5) and gpu[-1] <= max(3 * cpu[-1], 5e-5), med(gpu), gpu[-1], med(cpu), cpu[-1]))
    bad = [v for v in verdicts if not v[1]]
    for v in bad:
        print(f"{case} seed {v[0]}: GPU median {v[2]:.2e} max {v[3]:.2e} vs CPU fp32 {v[4]:.2e} / {v[5]:.2e} -- re-scan (tools/diag_fp64_seeds.py)")
    assert not bad, verdicts


def test_fp64_gradient_check_goes_red_on_a_wrong_statistics_row_count(monkeypatch):
    """Teeth: re-create round 2's bug class -- BatchNorm-backward reductions that sum one slab row too few -- behind the
    library's back (the C ABI's own row-count guard cannot see a short REDUCTION) and require the float64 check to fail by
    a wide margin.  The flip-tolerant model-level comparison (grads_match, 1e-2) let a 3e-4 error through in round 2."""
    from kdrt import ops
    seed = _fp64_table()["kd/weighted"][0]
    real = ops.bn_bwd_finalize
    calls = []

    def short(partial, rows, *a, **k):
        calls.append(rows)
        return real(partial, rows - 1 if rows > 1 else rows, *a, **k)
    monkeypatch.setattr(ops, "bn_bwd_finalize", short)
    cpu, gpu = _fp64_check("kd", "weighted", seed)
    assert any(r > 1 for r in calls)
    assert gpu[len(gpu) // 2] > 10 * max(3 * cpu[len(cpu) // 2], 1e-5), (gpu[len(gpu) // 2], cpu[len(cpu) // 2])


@pytest.mark.parametrize("student", ("concat", "minimal", "weighted"))
def test_kd_losses_and_logits_vs_oracle_for_every_student(student):
    """configs[4] (fusion ablation under KD): loss terms, logits and intermediates of the KD step against the oracle for
    each student fusion (round 2 pinned only concat -> weighted)."""
    from kdrt.losses import kd_objective
    teacher = build_product("concat", G); t_st = load_random_state(teacher, "concat", 11); teacher.eval()
    model = build_product(student, G); s_st = load_random_state(model, student, 12); model.train()
    images, pts, labels = O.make_inputs(B, HW, N, G, 6, pad_tail=40)
    cw = torch.tensor([0.4, 3.5])
    with torch.no_grad():
        zt, mt = teacher(images.cuda(), pts.cuda(), return_intermediates=True)
    zs, ms = model(images.cuda(), pts.cuda(), return_intermediates=True)
    total, parts = kd_objective(zs, ms, zt, mt, labels.cuda(), cw.cuda(), T=4.0, alpha=1.0, beta=1.0)
    total.backward()
    so = O.clone_state(s_st, requires_grad=True)
    with torch.no_grad():
        zt_o, mt_o = O.complete_model(images, pts, O.clone_state(t_st), fusion_type="concat", grid=(G, G), training=False)
    zs_o, ms_o = O.complete_model(images, pts, so, fusion_type=student, grid=(G, G), training=True)
    total_o, parts_o = O.kd_loss(zs_o, ms_o, zt_o, mt_o, labels, cw, T=4.0, alpha=1.0, beta=1.0)
    total_o.backward()
    assert abs(total.item() - total_o.item()) < 2e-4
    for k in ("ce", "kl", "mse_cam", "mse_lidar"):
        assert abs(parts[k].item() - parts_o[k].item()) < 1e-4, k
    assert max_err(zs, zs_o)[0] < LOGIT_TOL
    for k in ("camera_feat", "lidar_feat", "pre_fusion", "post_fusion"):
        assert max_err(ms[k], ms_o[k])[0] < ftol(ms_o[k]), k
    bad = [(n, m) for n, p in model.named_parameters() for ok, m in [grads_match(p.grad, so[n].grad)] if not ok]
    assert not bad, bad


@pytest.mark.parametrize("fusion", ("concat", "minimal", "weighted"))
def test_inference_epilogue_same_bits_as_two_pass_eval(fusion, monkeypatch):
    """Under no_grad the tail of every eval-mode chain (1x1 conv + BatchNorm + activation + residual) runs inside the GEMM
    epilogue; with autograd on, the same model takes the raw-GEMM + apply path.  Same bits, logits and intermediates.
    (The one-kernel eval LiDAR encoder is switched off here: it sums each k-step's sixteen products in another slot order
    than the layer-by-layer kernels -- its own test below bounds that difference.)"""
    from _gpu_util import build_product, load_random_state
    from kdrt import units
    monkeypatch.setattr(units, "_LIDAR_FUSED_INFER", False)
    G = 16
    images, pts, _ = O.make_inputs(2, 64, 700, G, 5, pad_tail=60)
    images, pts = images.cuda(), pts.cuda()
    model = build_product(fusion, G)
    load_random_state(model, fusion, 21)
    model.eval()
    with torch.no_grad():
        z1, m1 = model(images, pts, return_intermediates=True)
    z2, m2 = model(images, pts, return_intermediates=True)          # grad mode: two-pass tails
    assert torch.equal(z1.view(torch.int32), z2.detach().view(torch.int32))
    assert m1.keys() == m2.keys()
    for k in m1:
        assert torch.equal(m1[k].view(torch.int32), m2[k].detach().view(torch.int32)), k


@pytest.mark.parametrize("shape", ((2, 700, 16), (3, 20000, 64)))
def test_one_kernel_eval_lidar_encoder_against_layer_by_layer(shape, monkeypatch):
    """kd_lidar_mlp_scatter_infer (csrc/kd_lidar_infer.hip: point MLP 4 -> 64 -> 128 -> 128 + scatter-max, no activation leaves
    the CU, layer 1 computed transposed) against the layer-by-layer inference path and the CPU oracle.  Layer 2 adds the same
    products in a different slot order inside each MFMA: agreement to fp32 rounding, not bit for bit."""
    from kdrt import ops, units
    from src.models.lidar_encoder import LiDAREncoder
    B, N, G = shape
    _, pts, _ = O.make_inputs(B, 64, N, G, 9, pad_tail=N // 10)
    enc = LiDAREncoder(encoder_type="spatial", grid_size=(G, G)).cuda()
    st = O.randomize_state({k: v.detach().cpu().clone() for k, v in enc.state_dict().items()}, 33)
    enc.load_state_dict(st)
    enc.eval()
    with torch.no_grad():
        fused = enc(pts.cuda())
        monkeypatch.setattr(units, "_LIDAR_FUSED_INFER", False)
        layered = enc(pts.cuda())
    if ops.get_gemm_arithmetic() != "split":
        assert torch.equal(fused, layered)          # (no one-kernel instance in the exact-fp32 arithmetic: the same path twice)
        return
    scale = layered.abs().max().item()
    assert (fused - layered).abs().max().item() <= 2e-6 * scale, ((fused - layered).abs().max().item(), scale)
    assert torch.equal(fused == 0, layered == 0)                     # empty cells stay exactly zero
    want = O.spatial_lidar_encoder(pts, {k: v for k, v in st.items()}, "encoder.", (G, G), False)
    assert max_err(fused, want)[0] < ftol(want)
