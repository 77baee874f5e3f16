#!/usr/bin/env python3
"""Golden-vector generator -- runs ONLY in the build container, where /root/reference exists.

Imports the real reference modules (by file path, so nothing from the reference is copied or
installed), drives them on seeded inputs and name-keyed deterministic weights
(`kd_oracle.randomize_state`), and writes small `.npz` fixtures under tests/golden/.  The
fixtures hold data only (inputs are regenerated from seeds by the tests; expected outputs and
per-tensor gradient digests are stored).  The GPU box never sees the reference.

usage:  python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
from __future__ import annotations

import argparse
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import kd_oracle as O  # noqa: E402

sys.dont_write_bytecode = True


def _load(ref, rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ref, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def digest(t: torch.Tensor) -> np.ndarray:
    t = t.detach().double().reshape(-1)
    head = torch.zeros(4, dtype=torch.float64)
    head[: min(4, t.numel())] = t[:4]
    return np.concatenate([[t.sum().item(), t.norm().item(), t.abs().max().item()], head.numpy()])


def build(ref_mods, fusion, out_ch, num_classes=2, grid=16, output_mode="same"):
    cam_m, lid_m, fus_m = ref_mods
    cam = cam_m.TwinLiteEncoder(return_multiscale=True)
    lid = lid_m.LiDAREncoder(encoder_type="spatial", grid_size=(grid, grid), use_vectorized=True)
    return fus_m.CompleteSegmentationModel(
        camera_encoder=cam, lidar_encoder=lid, num_classes=num_classes, fusion_type=fusion,
        fusion_out_channels=out_ch, camera_fpn_stages=["stage3", "stage4", "stage5"],
        camera_fpn_channels=128, output_mode=output_mode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(HERE, "..", "tests", "golden"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    cam_m = _load(a.ref, "src/models/camera_encoder.py", "ref_camera_encoder")
    lid_m = _load(a.ref, "src/models/lidar_encoder.py", "ref_lidar_encoder")
    fus_m = _load(a.ref, "src/models/fusion_module.py", "ref_fusion_module")
    trn_m = _load(a.ref, "src/training/trainer.py", "ref_trainer")
    mods = (cam_m, lid_m, fus_m)

    # ---- 1. default-init pins: registration order, param counts, seed-0 weight digests -------
    torch.manual_seed(0)
    enc = cam_m.TwinLiteEncoder()
    sd = enc.state_dict()
    pins = {"cam_keys": np.array(list(sd.keys())),
            "cam_digest": np.stack([digest(v.float()) for v in sd.values()]),
            "cam_params": np.int64(enc.count_parameters())}
    enc.eval()
    x = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        y = enc(x)
    pins["cam_cfg1_out_digest"] = digest(y)
    pins["cam_cfg1_out_slice"] = y[0, :8, :4, :4].numpy()
    for fusion, oc in (("concat", 256), ("minimal", 128), ("weighted", 128)):
        torch.manual_seed(0)
        m = build(mods, fusion, oc, grid=64)
        s = m.get_architecture_summary()
        pins[f"{fusion}_total"] = np.int64(int(s["total_params"].replace(",", "")))
        pins[f"{fusion}_fusion"] = np.int64(int(s["fusion_params"].replace(",", "")))
        msd = m.state_dict()
        pins[f"{fusion}_keys"] = np.array(list(msd.keys()))
        pins[f"{fusion}_shapes"] = np.array([str(tuple(v.shape)) for v in msd.values()])
        pins[f"{fusion}_dtypes"] = np.array([str(v.dtype) for v in msd.values()])
        pins[f"{fusion}_digest"] = np.stack([digest(v.float()) for v in msd.values()])
    np.savez_compressed(os.path.join(a.out, "pins.npz"), **pins)

    # ---- 2. small full-model cases, eval + train(CE backward) ---------------------------------
    B, HW, N, G = 2, 64, 512, 16
    for fusion, oc in (("concat", 256), ("minimal", 128), ("weighted", 128)):
        for seed in (0, 1):
            model = build(mods, fusion, oc, grid=G)
            st = O.randomize_state(model.state_dict(), seed)
            model.load_state_dict(st)
            images, pts, labels = O.make_inputs(B, HW, N, G, seed, pad_tail=40)
            out = {}
            model.eval()
            with torch.no_grad():
                logits, mids = model(images, pts, return_intermediates=True)
                ms = model.camera_encoder(images)
            out["eval_logits"] = logits.numpy()
            out["eval_argmax"] = logits.argmax(1).numpy()
            out["eval_margin"] = np.float64((logits[:, 0] - logits[:, 1]).abs().min().item())
            if seed == 0:                      # full intermediates only for seed 0 (fixture size)
                for k in ("camera_feat", "lidar_feat", "pre_fusion", "post_fusion"):
                    out["eval_" + k] = mids[k].contiguous().numpy()
                for k, v in ms.items():
                    out["eval_" + k] = v.numpy()
            # train mode: CE loss, grads, BN buffers after one forward
            model.load_state_dict(st)
            model.train()
            cw = torch.tensor([0.4, 3.5])
            logits, mids = model(images, pts, return_intermediates=True)
            loss = torch.nn.CrossEntropyLoss(ignore_index=-1, weight=cw)(logits, labels)
            model.zero_grad()
            loss.backward()
            out["train_logits"] = logits.detach().numpy()
            out["train_loss"] = np.float64(loss.item())
            if seed == 0:
                for k in ("camera_feat", "lidar_feat"):
                    out["train_" + k] = mids[k].detach().contiguous().numpy()
            names = [n for n, _ in model.named_parameters()]
            out["grad_keys"] = np.array(names)
            out["grad_digest"] = np.stack([digest(p.grad) for _, p in model.named_parameters()])
            out["grad_head_cls_w"] = model.head.cls.weight.grad.numpy()
            out["grad_stem_w"] = model.camera_encoder.stem[0].weight.grad.numpy()
            out["grad_lidar_w0"] = model.lidar_encoder.encoder.point_mlp[0].weight.grad.numpy()
            bufs = {k: v for k, v in model.state_dict().items()
                    if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
            out["buf_keys"] = np.array(list(bufs.keys()))
            out["buf_digest"] = np.stack([digest(v.float()) for v in bufs.values()])
            # confusion matrix through the reference's SegmentationMetrics (python loop)
            sm = trn_m.SegmentationMetrics(num_classes=2)
            sm.update(logits.detach(), labels)
            out["confusion"] = sm.confusion.copy()
            out["miou"] = np.float64(sm.compute()["miou"])
            # one AdamW step (trainer.py:56)
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-3)
            opt.step()
            out["adamw_digest"] = np.stack([digest(p) for _, p in model.named_parameters()])
            np.savez_compressed(os.path.join(a.out, f"model_{fusion}_s{seed}.npz"), **out)

    # ---- 3. LiDAR encoder edge cases ------------------------------------------------------------
    lid = lid_m.SpatialLiDAREncoder(grid_size=(16, 16))
    st = O.randomize_state(lid.state_dict(), 3)
    lid.load_state_dict(st)
    out = {}
    g = torch.Generator().manual_seed(77)
    cases = {}
    p = O.make_inputs(2, 8, 256, 16, 9)[1]
    p[0, :6, 0] = torch.tensor([50.0, 49.99, 0.0, -50.0, -50.01, 50.01])
    p[0, :6, 1] = torch.tensor([50.0, -50.0, 0.0, 49.99, 0.0, 0.0])
    p[1, 100:140] = p[1, 99]                      # exact duplicates -> tie gradient
    p[1, 200:] = 0.0                              # zero-padded tail
    cases["edge"] = p
    pn = p.clone()
    pn[1, 3, 0] = float("nan")                    # NaN point: invalid in eval; poisons BN stats in train
    cases["nan"] = pn
    q = torch.randn(2, 64, 4, generator=g)
    q[..., 0] = 60.0 + q[..., 0].abs()            # all outside => all zeros
    cases["outside"] = q
    for name, pts in cases.items():
        for mode in (("eval",) if name == "nan" else ("eval", "train")):
            lid.load_state_dict(st)
            lid.train(mode == "train")
            out[f"{name}_points"] = pts.numpy()
            coords, valid = lid.points_to_bev_coords(pts)
            out[f"{name}_valid"] = valid.numpy()
            y = lid(pts)
            out[f"{name}_{mode}_out"] = y.detach().contiguous().numpy()
            if mode == "train" and y.requires_grad:
                up = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
                lid.zero_grad()
                (y * up).sum().backward()
                out[f"{name}_upstream"] = up.numpy()
                for n_, p_ in lid.named_parameters():
                    out[f"{name}_grad_{n_}"] = p_.grad.numpy()
        flat, v2 = O.bev_cell_index(pts, (16, 16))
        gc = (coords * lid.grid_tensor).long()
        gc[..., 0].clamp_(0, 15); gc[..., 1].clamp_(0, 15)
        b = torch.arange(pts.shape[0]).view(-1, 1).expand(pts.shape[:2])
        out[f"{name}_flat"] = (b * 256 + gc[..., 1] * 16 + gc[..., 0]).numpy()
    np.savez_compressed(os.path.join(a.out, "lidar_edges.npz"), **out)

    # ---- 4. full-size eval (256^2, N=5000, grid 64) --------------------------------------------
    model = build(mods, "weighted", 128, grid=64)
    st = O.randomize_state(model.state_dict(), 2)
    model.load_state_dict(st)
    model.eval()
    images, pts, labels = O.make_inputs(2, 256, 5000, 64, 2, pad_tail=300)
    with torch.no_grad():
        logits, mids = model(images, pts, return_intermediates=True)
    np.savez_compressed(os.path.join(a.out, "full_weighted_eval.npz"),
                        logits=logits.numpy(), argmax=logits.argmax(1).numpy(),
                        margin=np.float64((logits[:, 0] - logits[:, 1]).abs().min().item()),
                        camera_feat_digest=digest(mids["camera_feat"]),
                        lidar_feat_digest=digest(mids["lidar_feat"]))

    # ---- 5. KD step composed from reference modules + stock torch losses -----------------------
    teacher = build(mods, "concat", 256, grid=G)
    teacher.load_state_dict(O.randomize_state(teacher.state_dict(), 11))
    teacher.eval()
    student = build(mods, "weighted", 128, grid=G)
    student.load_state_dict(O.randomize_state(student.state_dict(), 12))
    student.train()
    images, pts, labels = O.make_inputs(B, HW, N, G, 4, pad_tail=40)
    T, alpha, beta = 4.0, 1.0, 1.0
    with torch.no_grad():
        zt, mt = teacher(images, pts, return_intermediates=True)
    zs, ms_ = student(images, pts, return_intermediates=True)
    ce = torch.nn.CrossEntropyLoss(ignore_index=-1, weight=torch.tensor([0.4, 3.5]))(zs, labels)
    kl = F.kl_div(F.log_softmax(zs / T, 1), F.softmax(zt / T, 1), reduction="sum") / (B * G * G)
    mse = F.mse_loss(ms_["camera_feat"], mt["camera_feat"]) + F.mse_loss(ms_["lidar_feat"], mt["lidar_feat"])
    total = ce + alpha * T * T * kl + beta * mse
    student.zero_grad()
    total.backward()
    np.savez_compressed(
        os.path.join(a.out, "kd_step.npz"), ce=np.float64(ce.item()), kl=np.float64(kl.item()),
        mse=np.float64(mse.item()), total=np.float64(total.item()),
        teacher_logits=zt.numpy(), student_logits=zs.detach().numpy(),
        grad_keys=np.array([n for n, _ in student.named_parameters()]),
        grad_digest=np.stack([digest(p.grad) for _, p in student.named_parameters()]))

    # ---- 6. x4 head (test_lidar_encoder.py:281-293) ---------------------------------------------
    m4 = build(mods, "concat", 256, num_classes=3, grid=G, output_mode="x4")
    m4.load_state_dict(O.randomize_state(m4.state_dict(), 21))
    m4.eval()
    images, pts, _ = O.make_inputs(B, HW, N, G, 6)
    with torch.no_grad():
        z4 = m4(images, pts)
    np.savez_compressed(os.path.join(a.out, "head_x4.npz"), logits=z4.numpy())
    # ---- 7. the iterative LiDAR path (use_vectorized=False, lidar_encoder.py:101-143) --------------
    # the reference's own Python double loop on the "edge" points (boundaries, 41 exact duplicates, zero-padded tail):
    # forward values in both modes and, in train mode, the parameter gradients under BOTH flags -- where backward
    # runs at all: the iterative path updates `feature_map[b, :, y, x]` in place, and autograd raises as soon as a
    # cell has received two points ("modified by an inplace operation"), i.e. that path is forward-only upstream
    lid_v = lid_m.SpatialLiDAREncoder(grid_size=(16, 16), use_vectorized=True)
    lid_i = lid_m.SpatialLiDAREncoder(grid_size=(16, 16), use_vectorized=False)
    st = O.randomize_state(lid_v.state_dict(), 3)
    pts = cases["edge"]
    out = {"points": pts.numpy()}
    up = torch.randn(2, 128, 16, 16, generator=torch.Generator().manual_seed(5))
    for mode in ("eval", "train"):
        ys = {}
        for tag, enc_ in (("vec", lid_v), ("iter", lid_i)):
            enc_.load_state_dict(st)
            enc_.train(mode == "train")
            ys[tag] = enc_(pts)
        out[f"iter_{mode}_out"] = ys["iter"].detach().contiguous().numpy()
        out[f"iter_{mode}_same_bits_as_vectorized"] = np.bool_(torch.equal(ys["iter"], ys["vec"]))
        if mode == "train":
            try:
                (ys["iter"] * up).sum().backward()
                out["iter_backward"] = np.array("ok")
            except RuntimeError as e:          # in-place map update: autograd refuses -- a forward-only path upstream
                out["iter_backward"] = np.array("RuntimeError: " + str(e).split(":")[0])
    np.savez_compressed(os.path.join(a.out, "lidar_iterative.npz"), **out)

    print("golden fixtures written to", os.path.abspath(a.out))
    for f in sorted(os.listdir(a.out)):
        print(f"  {f}: {os.path.getsize(os.path.join(a.out, f))/1024:.1f} KiB")


if __name__ == "__main__":
    main()
