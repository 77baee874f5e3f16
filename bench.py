#!/usr/bin/env python3
"""bench.py -- PandaSet-shaped 2-class KD training throughput (frames/s) on N MI355X of one node.

One "step" = one knowledge-distillation training step over a synthetic batch that is already
resident in HBM: concat-fusion teacher forward (eval, no grad) -> weighted-fusion student forward
(train-mode BN) -> CE + alpha*T^2*KL + beta*feature-MSE -> student backward -> (N>1: bucketed RCCL
all-reduce of the gradients, overlapped with backward) -> fused AdamW.  Every arithmetic kernel is
a hand-written gfx950 kernel from libkd_hip.so; fp32 end to end (the parity contract is fp32).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 either launched through
torch.distributed.run (one rank per GPU, RANK / WORLD_SIZE in the environment) or started bare, in which case
this process -- before it touches any GPU -- starts its own N ranks as a child `python -m torch.distributed.run`
and exits with the child's status.  Every rank checks `WORLD_SIZE == --gpus`.  Rank 0 prints ONE JSON line.

`--student-fusion` / `--teacher-fusion` select the fusion variants (the reference's ablation sweep,
train_with_fusion_ablation.py:87-114: concat / minimal / weighted students); the headline workload is
concat teacher -> weighted student.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd")
sys.path.insert(0, PKG)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA
MFMA_F32_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
HBM_PEAK_GBPS = 8000.0            # MI355X HBM3E spec peak (same guide); a streaming copy reaches ~5.3-5.6 TB/s


FUSION_CH = {"concat": 256, "minimal": 128, "weighted": 128}     # train_with_fusion_ablation.py:87-91


def build_models(grid, teacher_fusion="concat", student_fusion="weighted"):
    from src.models.camera_encoder import TwinLiteEncoder
    from src.models.fusion_module import CompleteSegmentationModel
    from src.models.lidar_encoder import LiDAREncoder

    def mk(fusion, oc):
        return CompleteSegmentationModel(
            TwinLiteEncoder(return_multiscale=True), LiDAREncoder(encoder_type="spatial", grid_size=(grid, grid)),
            num_classes=2, fusion_type=fusion, fusion_out_channels=oc,
            camera_fpn_stages=["stage3", "stage4", "stage5"], camera_fpn_channels=128, output_mode="same")
    torch.manual_seed(0)
    teacher = mk(teacher_fusion, FUSION_CH[teacher_fusion])
    student = mk(student_fusion, FUSION_CH[student_fusion])
    return teacher, student


def synth_batch(B, N, HW, grid, seed, device, sigma=40.0):
    g = torch.Generator(device=device).manual_seed(seed)
    images = torch.rand(B, 3, HW, HW, generator=g, device=device)
    pts = torch.randn(B, N, 4, generator=g, device=device)          # lidar_encoder.py:227-234 recipe
    pts[..., 0] *= sigma
    pts[..., 1] *= sigma
    pts[..., 2] = pts[..., 2] * 4.0 - 1.0
    pts[..., 3] = torch.sigmoid(pts[..., 3])
    labels = torch.randint(0, 2, (B, grid, grid), generator=g, device=device)
    return images, pts, labels


def cpu_baseline(teacher, student, N, HW, grid, teacher_fusion, student_fusion, budget_s=30.0):
    """The CPU oracle (oracle/kd_oracle.py, a port of the reference path in stock PyTorch) timed on
    this box's host cores on a bounded sample of the same workload: B=4 frames (the reference's batch),
    3 warm-up + 10 timed KD steps; the timed count is cut only if the first warm-up
    step shows the sample would exceed `budget_s`, and the cut is stated in `sample`."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import kd_oracle as O
    # the GPU box exposes 256 logical CPUs but a 1-GPU job owns a 16-core share: use that many threads
    cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("KD_CPU_THREADS", 16)))
    torch.set_num_threads(cores)
    B = 4
    t_st = {k: v.detach().cpu().clone() for k, v in teacher.state_dict().items()}
    s_st = O.clone_state({k: v.detach().cpu().clone() for k, v in student.state_dict().items()}, requires_grad=True)
    images, pts, labels = synth_batch(B, N, HW, grid, 4321, "cpu")
    cw = torch.tensor([0.4, 3.5])
    keys = O.trainable_keys(s_st)
    m = [torch.zeros_like(s_st[k]) for k in keys]
    v = [torch.zeros_like(s_st[k]) for k in keys]

    def one(step):
        for k in keys:
            s_st[k].grad = None
        with torch.no_grad():
            zt, mt = O.complete_model(images, pts, t_st, fusion_type=teacher_fusion, grid=(grid, grid), training=False)
        zs, ms = O.complete_model(images, pts, s_st, fusion_type=student_fusion, grid=(grid, grid), training=True)
        total, _ = O.kd_loss(zs, ms, zt, mt, labels, cw)
        total.backward()
        with torch.no_grad():
            O.adamw_step([s_st[k] for k in keys], [s_st[k].grad for k in keys], m, v, step, lr=1e-3, weight_decay=1e-3)

    t0 = time.perf_counter()
    one(1)
    first = time.perf_counter() - t0
    warm, iters = 3, 10
    if first * (warm + iters) > budget_s:                     # slower host than planned: keep the run bounded
        warm = 1
        iters = max(2, min(10, int(budget_s / max(first, 1e-3)) - warm))
    for i in range(warm - 1):
        one(2 + i)
    t0 = time.perf_counter()
    for i in range(iters):
        one(1 + warm + i)
    dt = (time.perf_counter() - t0) / iters
    cut = "" if (warm, iters) == (3, 10) else f" (cut from 3 + 10 to fit {budget_s:.0f} s: first step took {first:.1f} s)"
    res = {"value": round(B / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port",
           "sample": f"{iters} timed KD steps of B={B} frames after {warm} warm-up steps{cut}, N={N} points, image {HW}x{HW}, "
                     f"{teacher_fusion} teacher -> {student_fusion} student ({dt*1e3:.0f} ms/step, torch CPU threads={torch.get_num_threads()})"}
    # BASELINE.md section 3 cases (a) and (b), same port, same cores, 3 warm-up + 10 timed iterations each:
    # (a) configs[0]: camera encoder alone, eval forward of a random 4x3x224x224 batch (test_camera_encoder.py:24-40)
    cam_st = {k[len("camera_encoder."):]: v for k, v in t_st.items() if k.startswith("camera_encoder.")}
    x = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(5))

    def timed(fn, warm=3, iters=10):
        for _ in range(warm):
            fn()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        return (time.perf_counter() - t0) / iters
    with torch.no_grad():
        ta = timed(lambda: O.twinlite_encoder(x, cam_st, "", False, False))
    # (b) the reference trainer's own step (trainer.py:86-90): weighted CE, backward, AdamW; B=4, N=5000 (the reference's
    # default points per frame)
    im5, pt5, lb5 = synth_batch(B, 5000, HW, grid, 4321, "cpu")
    s5 = O.clone_state({k: v.detach().cpu().clone() for k, v in student.state_dict().items()}, requires_grad=True)
    m5 = [torch.zeros_like(s5[k]) for k in keys]
    v5 = [torch.zeros_like(s5[k]) for k in keys]
    n5 = [0]

    def ce_step():
        n5[0] += 1
        for k in keys:
            s5[k].grad = None
        z, _ = O.complete_model(im5, pt5, s5, fusion_type=student_fusion, grid=(grid, grid), training=True)
        O.weighted_ce(z, lb5, cw).backward()
        with torch.no_grad():
            O.adamw_step([s5[k] for k in keys], [s5[k].grad for k in keys], m5, v5, n5[0], lr=1e-3, weight_decay=1e-3)
    tb = timed(ce_step)
    res["cases"] = {
        "a_camera_encoder_eval_forward_4x3x224x224": {"value": round(4 / ta, 2), "unit": "frames/s", "ms_per_forward": round(ta * 1e3, 2),
                                                      "sample": "10 timed forwards after 3 warm-ups"},
        "b_ce_train_step_B4_N5000": {"value": round(B / tb, 3), "unit": "frames/s", "ms_per_step": round(tb * 1e3, 1),
                                     "sample": f"10 timed steps after 3 warm-ups, {student_fusion} student, weighted CE + AdamW"},
        "c_kd_train_step_B4_N%d" % N: {"value": res["value"], "unit": "frames/s", "ms_per_step": round(dt * 1e3, 1), "sample": "the headline sample above"}}
    return res


def bf16_forward_bench(args, dev, images, pts, steps):
    """BASELINE.json configs[1]: camera + LiDAR concat-fusion FORWARD with bf16 activations (kdrt/bf16.py, csrc/kd_bf16.hip),
    timed beside the fp32 eval forward of the same model on the same batch; reports both rates, the logit error and the
    argmax agreement at this size.  A second mode: never the headline, never a substitute for the fp32 parity contract."""
    from kdrt.bf16 import forward_bf16
    _, model = build_models(args.grid, "concat", "concat")
    model = model.to(dev).eval()

    def rate(fn):
        for _ in range(2):
            z = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            z = fn()
        torch.cuda.synchronize()
        return args.batch * steps / (time.perf_counter() - t0), z

    with torch.no_grad():
        r32, z32 = rate(lambda: model(images, pts))
    r16, z16 = rate(lambda: forward_bf16(model, images, pts))
    rng = float(z32.max() - z32.min())
    # per-kernel-family HIP-event timing of one more forward (events on the launch stream): the bf16 mode's own roofline
    from kdrt import ops
    ops.PROFILE = []
    forward_bf16(model, images, pts)
    torch.cuda.synchronize()
    recs, ops.PROFILE = ops.PROFILE, None
    fam = {}
    for rec in recs:
        kind, flops, nbytes, secs, _ = ops.prof_scaled(rec)
        f = fam.setdefault(kind, [0.0, 0.0, 0.0, 0])
        f[0] += flops; f[1] += nbytes; f[2] += secs; f[3] += 1
    top = max(fam, key=lambda k: fam[k][2])
    tot_b, tot_s = sum(f[1] for f in fam.values()), sum(f[2] for f in fam.values())
    roof = {"bound": "hbm", "kernel": {"bf16_pw": "pw_gemm_bf16_v2_kernel / pw_gemm_bf16_kernel (1x1 conv / point MLP, one bf16 MFMA product per element)",
                                       "bf16_dw": "dw_bf16_kernel (depthwise 3x3)"}.get(top, top),
            "achieved": round(fam[top][1] / fam[top][2] / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(fam[top][1] / fam[top][2] / 1e9 / HBM_PEAK_GBPS, 4), "launches": fam[top][3],
            "traffic": None, "algorithmic_bytes_per_forward": int(fam[top][1]),
            "executed_bf16_tflops": round(fam[top][0] / fam[top][2] / 1e12, 1),
            "all_bf16_kernels": {"achieved": round(tot_b / tot_s / 1e9, 1), "frac": round(tot_b / tot_s / 1e9 / HBM_PEAK_GBPS, 4),
                                 "ms": round(tot_s * 1e3, 3), "launches": sum(f[3] for f in fam.values())},
            "by_family": {k: {"ms": round(f[2] * 1e3, 3), "GB/s": round(f[1] / f[2] / 1e9, 1), "launches": f[3]} for k, f in sorted(fam.items())},
            "note": "algorithmic bytes = every bf16/fp32 operand and result of a launch counted once; the fp32 point sort is outside these families"}
    return {"metric": "concat-fusion forward frames/sec, bf16 activations in HBM (fp32 accumulate)", "dtype": "bf16", "roofline": roof,
            "value": round(r16, 1), "unit": "frames/s", "fp32_forward_frames_per_s": round(r32, 1), "speedup_vs_fp32_forward": round(r16 / r32, 2),
            "steps": steps, "per_gpu_batch": args.batch, "points_per_frame": args.points,
            "max_abs_logit_error_vs_fp32": float(f"{float((z16 - z32).abs().max()):.3e}"), "logit_range": float(f"{rng:.4g}"),
            "argmax_agreement_vs_fp32": round(float((z16.argmax(1) == z32.argmax(1)).float().mean()), 5)}


def side_rate(args, dev, student_fusion, batch, steps, graph=False):
    """frames/s of a KD step of another configuration (same teacher, points, image), fresh models; outside the timed headline."""
    from kdrt import gradsink
    from kdrt.kd import GraphedKDStep, KDStep
    from kdrt.optim import FusedAdamW
    teacher, student = build_models(args.grid, args.teacher_fusion, student_fusion)
    teacher, student = teacher.to(dev).eval(), student.to(dev).train()
    opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)
    step = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5], device=dev), T=4.0, alpha=1.0, beta=1.0)
    images, pts, labels = synth_batch(batch, args.points, args.image, args.grid, 1234, dev, args.points_sigma)
    run = GraphedKDStep(step, images, pts, labels) if graph else (lambda: step(images, pts, labels))
    for _ in range(2):
        parts = run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        parts = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ok = bool(torch.isfinite(parts["total"]).item())
    del run, step, opt, teacher, student, images, pts, labels
    gradsink.uninstall()
    torch.cuda.empty_cache()
    return {"value": round(batch / dt, 1), "unit": "frames/s", "ms_per_step": round(dt * 1e3, 3), "steps": steps, "finite": ok}


def kernel_code_state():
    """Short hash of the kernel sources: stored beside a committed PMC measurement so a stale one is visible."""
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:16]


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This parent has not touched the
    GPU (torch.cuda.device_count() does not initialise it) and stays a plain parent: no exec, the ranks are children
    of a child `python -m torch.distributed.run`; its stdout (rank 0's JSON line) passes straight through."""
    rehearse = os.environ.get("KD_REHEARSE_ON_ONE_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not rehearse:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this node")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    raise SystemExit(subprocess.call(cmd, env=env))


def selfcheck(args, dev, teacher_fusion, student_fusion):
    """Correctness gate on the benchmarked size (per-GPU batch B, N points: [B*N, 128] tensors beyond 2^31 elements at
    the default B=256): a batch made of TWO COPIES of the first B/2 frames must give the same losses and logits as
    those B/2 frames alone (BatchNorm statistics, mean losses and mean gradients are invariant under duplication), and
    everything must be finite.  Same seeds as the timed run; fresh models, freed afterwards."""
    from kdrt import gradsink
    from kdrt.losses import kd_objective
    half = args.batch // 2
    if half < 1:
        return {"skipped": "per-GPU batch of 1"}
    images, pts, labels = synth_batch(half, args.points, args.image, args.grid, 1234, dev, args.points_sigma)
    cw = torch.tensor([0.4, 3.5], device=dev)
    res = []
    prev_sink, gradsink.active = gradsink.active, None
    try:
        for reps in (1, 2):
            teacher, student = build_models(args.grid, teacher_fusion, student_fusion)
            teacher, student = teacher.to(dev).eval(), student.to(dev).train()
            im, pt, lb = images.repeat(reps, 1, 1, 1), pts.repeat(reps, 1, 1), labels.repeat(reps, 1, 1)
            with torch.no_grad():
                zt, mt = teacher(im, pt, return_intermediates=True)
            zs, ms = student(im, pt, return_intermediates=True)
            total, parts = kd_objective(zs, ms, zt, mt, lb, cw, 4.0, 1.0, 1.0, -1)
            total.backward()
            gn = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in student.parameters()))
            vals = {k: float(v) for k, v in parts.items()} | {"total": float(total.detach()), "grad_norm": float(gn)}
            groups = {}                                       # per top-level module: a missing small group must show
            for nm, p in student.named_parameters():
                groups[nm.split(".")[0]] = groups.get(nm.split(".")[0], 0.0) + float((p.grad.double() ** 2).sum())
            vals.update({"grad_norm/" + k: v ** 0.5 for k, v in groups.items()})
            res.append((vals, zs[:half].detach().clone()))
            del teacher, student, zt, mt, zs, ms, total, parts, im, pt, lb
            torch.cuda.empty_cache()
    finally:
        gradsink.active = prev_sink
    (a, za), (b, zb) = res
    import math
    finite = all(math.isfinite(x) for x in list(a.values()) + list(b.values()))
    dl = max(abs(a[k] - b[k]) / max(1.0, abs(a[k])) for k in a if not k.startswith("grad_norm"))
    dz = float((za - zb).abs().max()) / max(1.0, float(za.abs().max()))
    dg = max(abs(a[k] - b[k]) / max(a[k], 1e-12) for k in a if k.startswith("grad_norm"))     # whole model and every module group
    ok = finite and dl <= 2e-5 and dz <= 1e-4 and dg <= 2e-2
    out = {"ok": bool(ok), "what": f"{2*half} frames (two copies of {half}) vs {half} frames, same seeds as the timed run",
           "losses_at_bench_batch": {k: round(v, 6) for k, v in b.items() if not k.startswith("grad_norm/")}, "max_rel_loss_diff": dl, "max_rel_logit_diff": dz,
           "rel_grad_norm_diff": dg, "tolerances": {"loss": 2e-5, "logits": 1e-4, "grad_norm": 2e-2}}
    if not ok:
        raise SystemExit("bench.py self-check FAILED at the benchmarked size: " + json.dumps(out))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    # 256 frames/GPU: the larger of the two throughput batches SURVEY.md section 8d names (the reference's B=4 is
    # launch-bound); ~56 GiB of the 288 GB HBM.  Measured on one MI355X (round 1): 4 -> 677 (907 under hipGraph
    # replay), 32 -> 1960, 64 -> 2100, 128 -> 2260, 256 -> 2307 frames/s
    ap.add_argument("--batch", type=int, default=int(os.environ.get("KD_BENCH_BATCH", 256)), help="frames per GPU per step")
    ap.add_argument("--points", type=int, default=80000)
    ap.add_argument("--image", type=int, default=256)
    ap.add_argument("--grid", type=int, default=64)
    ap.add_argument("--student-fusion", choices=sorted(FUSION_CH), default="weighted",
                    help="fusion variant of the student (the ablation sweep of train_with_fusion_ablation.py:87-114)")
    ap.add_argument("--teacher-fusion", choices=sorted(FUSION_CH), default="concat")
    ap.add_argument("--points-sigma", type=float, default=40.0,
                    help="std (m) of the synthetic x/y coordinates; 40 = the reference's recipe (the headline workload), small "
                         "values pile the points into few BEV cells like the near field of a real sweep (robustness check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-selfcheck", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write the profiled GEMM launches one per line to this file")
    ap.add_argument("--no-bf16-forward", action="store_true", help="skip the bf16-storage forward line (configs[1])")
    ap.add_argument("--no-side-benches", action="store_true",
                    help="skip the fusion-ablation (configs[4]) and small-batch (B=4 eager / hipGraph) lines of the N=1 run")
    ap.add_argument("--force-reducer", action="store_true",
                    help="--gpus 1 only: bring up RCCL with a world of one rank and run the timed steps WITH the bucketed "
                         "gradient all-reduce (a one-rank sum: same bits), then time the same steps without it and report the delta")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args)                                    # never returns
    if args.force_reducer and args.gpus != 1:
        raise SystemExit("--force-reducer is the 1-GPU rehearsal of the RCCL path; with --gpus N > 1 the reducer always runs")

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # KD_REHEARSE_ON_ONE_GPU=1: every rank uses cuda:0 and the gloo backend -- only to rehearse the N>1
    # code path (broadcast, bucketed async all-reduce, MAX-reduced timing) on a 1-GPU box; never a result.
    rehearse = os.environ.get("KD_REHEARSE_ON_ONE_GPU") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank_info = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)    # "nccl" is RCCL on ROCm
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        # every rank present and reduced over: an all-reduce of ones must count --gpus ranks; device ids gathered
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        seen = int(ones.item())
        if seen != args.gpus:
            raise SystemExit(f"bench.py: all-reduce of ones saw {seen} ranks, expected {args.gpus}")
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "name": props.name,
                "host": socket.gethostname(), "pid": os.getpid()}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        if not rehearse and len({(r["host"], r["device"]) for r in gathered}) != world:
            raise SystemExit(f"bench.py: two ranks share a GPU: {gathered}")
        rank_info = {"rccl_ranks_seen": seen, "ranks": gathered}
    elif args.force_reducer:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rank_info = {"rccl_ranks_seen": int(ones.item())}

    from kdrt import ops
    from kdrt.ddp import BucketedAllReduce, broadcast_module
    from kdrt.kd import KDStep
    from kdrt.optim import FusedAdamW

    check = None
    if not args.no_selfcheck:
        check = selfcheck(args, dev, args.teacher_fusion, args.student_fusion)     # every rank checks its own GPU
        torch.cuda.reset_peak_memory_stats(dev)
    teacher, student = build_models(args.grid, args.teacher_fusion, args.student_fusion)
    teacher, student = teacher.to(dev).eval(), student.to(dev).train()
    if world > 1 or args.force_reducer:
        broadcast_module(student)
        broadcast_module(teacher)
    names = [n for n, p in student.named_parameters() if p.requires_grad]
    opt = FusedAdamW(student.parameters(), lr=1e-3, weight_decay=1e-3)          # trainer.py:56 values
    reducer = BucketedAllReduce(opt.flat, names, n_buckets=3, force=args.force_reducer) if (world > 1 or args.force_reducer) else None
    step = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5], device=dev), T=4.0, alpha=1.0, beta=1.0,
                  reducer=reducer)
    images, pts, labels = synth_batch(args.batch, args.points, args.image, args.grid, 1234 + rank, dev, args.points_sigma)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    first = None
    for i in range(args.warmup):
        parts = step(images, pts, labels)
        if i == 0:
            first = parts
    barrier()
    if reducer is not None:
        reducer.exposed = []                                  # event pairs around the waits of finish(): see the `comm` block
    t0 = time.perf_counter()
    for _ in range(args.steps):
        parts = step(images, pts, labels)
        if first is None:
            first = parts
    barrier()
    elapsed = time.perf_counter() - t0
    comm = None
    if reducer is not None:
        # exposed all-reduce time: how long the COMPUTE stream sat in the waits of reducer.finish() (HIP events recorded on it
        # right before and right after the waits); 0 when the reductions finished under the rest of backward
        ev, reducer.exposed = reducer.exposed, None
        exposed_ms = sum(a.elapsed_time(b) for a, b in ev) / max(len(ev), 1)
        comm = {"bucket_bytes": [int(v.numel() * 4) for v in reducer.views], "buckets_per_step": len(reducer.views),
                "collectives_per_step": reducer.collectives_issued / max(args.warmup + args.steps, 1),
                "allreduce_exposed_ms_per_step": round(exposed_ms, 4), "steps_measured": len(ev),
                "how": "HIP events on the compute stream around the work.wait() calls of BucketedAllReduce.finish(), timed steps only"}
    # outputs of the timed region must be numbers: step-0 and last-step losses, every parameter and moment finite
    first_losses = {k: float(first[k]) for k in ("total", "ce", "kl", "mse_cam", "mse_lidar")}
    last_losses = {k: float(parts[k]) for k in ("total", "ce", "kl", "mse_cam", "mse_lidar")}
    finite = (all(x == x and abs(x) != float("inf") for x in list(first_losses.values()) + list(last_losses.values()))
              and bool(torch.isfinite(opt.flat.data).all()) and bool(torch.isfinite(opt.flat.grad).all())
              and bool(torch.isfinite(opt.exp_avg_sq).all()))
    if world > 1:
        f = torch.tensor([1.0 if finite else 0.0], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        finite = bool(f.item() > 0)
    if not finite:
        raise SystemExit(f"bench.py: non-finite loss / parameter after the timed steps: {first_losses} -> {last_losses}")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    frames = args.batch * world * args.steps
    out = {
        "metric": "PandaSet 2-class KD training frames/sec", "value": round(frames / elapsed, 2), "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU over gloo, not a result)" if rehearse else ""),
        "config": {
            "workload": f"KD step: {args.teacher_fusion}-fusion teacher fwd (eval) -> {args.student_fusion}-fusion student fwd/bwd "
                        "(train BN), CE + T^2*KL(T=4) + feature-MSE, fused AdamW; random-init weights",
            "teacher_fusion": args.teacher_fusion, "student_fusion": args.student_fusion,
            "image": f"3x{args.image}x{args.image}", "points_per_frame": args.points, "bev_grid": f"{args.grid}x{args.grid}",
            "num_classes": 2, "gemm_arithmetic": ("fp32 operands as 3 bf16 pieces, 6 piece products on the bf16 matrix pipe, fp32 accumulate (fp32-grade)"
                                                   if ops.get_gemm_arithmetic() == "split" else "exact fp32 MFMA products"),
            "per_gpu_batch": args.batch, "global_batch": args.batch * world,
            "parallelism": f"dp{world}" + (" (bucketed RCCL all-reduce overlapped with backward)" if world > 1 else ""),
            "peak_hbm_gib_per_gpu": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1)},
        "checks": {"finite": finite, "losses_step0": {k: round(v, 6) for k, v in first_losses.items()},
                   "losses_last": {k: round(v, 6) for k, v in last_losses.items()}, "selfcheck": check},
    }
    if rank_info is not None:
        out.update(rank_info)
    if comm is not None:
        if world > 1:
            t = torch.tensor([comm["allreduce_exposed_ms_per_step"]], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            comm["allreduce_exposed_ms_per_step_max_over_ranks"] = round(float(t.item()), 4)
        out["comm"] = comm
    if world > 1:
        out["cpu_baseline"] = None
        out["bf16_forward"] = None
        out["why_null"] = "cpu_baseline / bf16_forward / kd_step_bf16_teacher are measured by the N=1 run only (rank 0, one GPU)"
    if args.force_reducer:
        # the same steps WITHOUT the reducer (its hooks silenced), same models and optimiser state: the per-step cost of
        # the three asynchronous RCCL all-reduces + wait() ordering on this box
        reducer.enabled = False
        plain = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5], device=dev), T=4.0, alpha=1.0, beta=1.0)
        for _ in range(max(1, args.warmup)):
            plain(images, pts, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            plain(images, pts, labels)
        torch.cuda.synchronize()
        tp = (time.perf_counter() - t0) / args.steps
        reducer.enabled = True
        out["forced_reducer"] = {
            "what": "world of ONE rank over RCCL (backend nccl): 3 async all_reduce per step on slices of the flat gradient buffer",
            "ms_per_step_with_reducer": round(1e3 * elapsed / args.steps, 3), "ms_per_step_plain": round(1e3 * tp, 3),
            "delta_ms_per_step": round(1e3 * (elapsed / args.steps - tp), 3),
            "collectives_issued": reducer.collectives_issued, "bucket_bytes": [int(v.numel() * 4) for v in reducer.views]}

    if not args.no_roofline:
        # Live per-kernel timing of the dominant kernel family (the fp32-MFMA pointwise-conv GEMMs):
        # HIP events around every GEMM launch on the launch stream, over 2 extra steps.  EVERY rank runs
        # the extra steps (they contain the gradient all-reduce: a rank-0-only step would deadlock the
        # other ranks' collectives); only rank 0 records events and reports.
        if rank == 0:
            ops.PROFILE = []
        for _ in range(2):
            step(images, pts, labels)
        barrier()
    if rank == 0 and not args.no_roofline:
        recs, ops.PROFILE = ops.PROFILE, None
        agg = {}
        torch.cuda.synchronize()
        arith = ops.get_gemm_arithmetic()
        # per-launch roofline floor: max(matrix-pipe time, HBM time) with the pipe peak of the arithmetic in use
        # (split: 6 bf16 MFMA products per fp32 product -> 2500 / 6 TFLOP/s fp32-equivalent) and the 8 TB/s HBM peak
        pipe_peak = (MFMA_BF16_PEAK_TFLOPS / 6.0 if arith == "split" else MFMA_F32_PEAK_TFLOPS) * 1e12
        groups = {}
        per_launch = []
        for rec in recs:
            kind, flops, nbytes, secs, M = ops.prof_scaled(rec)      # compacted-point launches: actual row count
            per_launch.append((kind, M, flops, nbytes, secs))
            a = agg.setdefault(kind, [0.0, 0.0, 0.0, 0])
            a[0] += flops; a[1] += nbytes; a[2] += secs; a[3] += 1
            grp = "lidar_point_mlp" if M == args.batch * args.points else "camera_fpn_fusion_head"
            q = groups.setdefault(grp, [0.0, 0.0, 0.0, 0.0, 0])
            q[0] += flops; q[1] += nbytes; q[2] += secs; q[3] += max(flops / pipe_peak, nbytes / (HBM_PEAK_GBPS * 1e9)); q[4] += 1
        if args.dump_launches:           # dev aid: the two profiled steps launch by launch (nominal rows, N*K, time, rates)
            with open(args.dump_launches, "w") as f:
                f.write("kind        rows        N*K   us      GB/s   fp32-equiv TFLOP/s   floor us (max of pipe, HBM)\n")
                for kind, M, flops, nbytes, secs in per_launch:
                    floor = max(flops / pipe_peak, nbytes / (HBM_PEAK_GBPS * 1e9))
                    f.write(f"{kind:10s} {M:9d} {int(flops / (2.0 * max(M, 1))):9d} {secs*1e6:8.1f} {nbytes/secs/1e9:8.1f} {flops/secs/1e12:8.1f} {floor*1e6:10.1f}\n")
        g = agg.get("pw_gemm", [0, 0, 1e-9, 0])
        w = agg.get("pw_wgrad", [0, 0, 1e-9, 0])
        lb = agg.get("lidar_bwd", None)
        # The family is HBM-bound since its products moved to the bf16 matrix pipe (bf16x6 split arithmetic): the
        # roofline is algorithmic bytes / launch time against the HBM peak.  The matrix-pipe view is kept beside it:
        # fp32-equivalent FLOP/s (2MKN) and the bf16 FLOP/s actually executed (6 piece products per product).
        gbps = g[1] / g[2] / 1e9
        tf = g[0] / g[2] / 1e12
        # Headline fraction: the roofline north_star words its target in -- the camera-side convolution
        # GEMMs (encoder, FPN, fusion, head; forward + data gradient + weight gradient) against the per-launch roofline
        # max(FLOP / matrix-pipe peak, bytes / HBM peak) summed over their launches.  `achieved` = their fp32-equivalent
        # TFLOP/s, `peak` = the TFLOP/s the same launches would reach if each ran exactly at its own floor, frac = achieved /
        # peak.  The HBM-only view of the forward / data-gradient family (rounds 1-3's headline) stays beside it in `hbm_only`.
        cam = groups.get("camera_fpn_fusion_head", [0.0, 0.0, 1e-9, 1e-12, 0])
        cam_frac = cam[3] / cam[2]
        cam_tf = cam[0] / cam[2] / 1e12
        out["roofline"] = {
            "bound": "mixed", "achieved": round(cam_tf, 2), "peak": round(cam_tf / max(cam_frac, 1e-9), 2), "unit": "TFLOP/s",
            "frac": round(cam_frac, 4), "traffic": None,
            "what": "camera / FPN / fusion / head 1x1-convolution GEMMs (fwd + dgrad + wgrad launches of two profiled steps): sum of per-launch "
                    "floors max(2MKN*6 / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s) over their HIP-event time; fp32-equivalent TFLOP/s",
            "mfma_busy_pct": None,
            "hbm_only": {"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(gbps / HBM_PEAK_GBPS, 4),
                         "what": "forward + data-gradient GEMM family incl. the point-MLP forward layers: algorithmic bytes / launch time"},
            "kernel": "pw_stream_kernel<KB,NB,PRO,EPI> / pw_gemm_kernel<PRO,EPI> (1x1-conv fwd + dgrad incl. the point-MLP forward layers, weight-resident streaming form where the shape allows, tiled form otherwise; " +
                      ("bf16x6 split products on v_mfma_f32_32x32x16_bf16, fp32 accumulate)" if arith == "split"
                       else "v_mfma_f32_32x32x2_f32)"),
            "launches_per_step": g[3] // 2, "avg_launch_us": round(1e6 * g[2] / max(g[3], 1), 2),
            "algorithmic_mbyte_per_launch": round(g[1] / max(g[3], 1) / 1e6, 2),
            "algorithmic_gflop_per_launch": round(g[0] / max(g[3], 1) / 1e9, 3),
            "fp32_equivalent_tflops": round(tf, 2),
            "matrix_pipe": ({"executed_bf16_tflops": round(6 * tf, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "frac": round(6 * tf / MFMA_BF16_PEAK_TFLOPS, 4)}
                            if arith == "split" else {"executed_fp32_tflops": round(tf, 2), "peak": MFMA_F32_PEAK_TFLOPS, "frac": round(tf / MFMA_F32_PEAK_TFLOPS, 4)}),
            "gemm_time_share_of_step": round(g[2] / 2 / (elapsed / args.steps), 3),
            # all GEMM launches (fwd + dgrad + wgrad) by where they sit in the model; frac_of_roofline = sum of per-launch
            # max(flops / pipe peak, bytes / HBM peak) over the measured time; fp32_mfma_frac = fp32-equivalent FLOP/s / 157.3
            "by_group": {k: {"launches_per_step": v[4] // 2, "ms_per_step": round(v[2] / 2 * 1e3, 2),
                             "fp32_equivalent_tflops": round(v[0] / v[2] / 1e12, 1), "algorithmic_GBps": round(v[1] / v[2] / 1e9, 0),
                             "frac_of_roofline": round(v[3] / v[2], 3),
                             "fp32_mfma_frac": round(v[0] / v[2] / 1e12 / MFMA_F32_PEAK_TFLOPS, 3)} for k, v in groups.items()},
            "wgrad": {"achieved": round(w[1] / w[2] / 1e9, 1), "unit": "GB/s", "fp32_tflops": round(w[0] / w[2] / 1e12, 2),
                      "launches_per_step": w[3] // 2, "time_share_of_step": round(w[2] / 2 / (elapsed / args.steps), 3)},
        }
        if lb is not None:
            # the one-kernel LiDAR layer backwards (kd_lidar_l2_bwd / kd_lidar_l1_bwd: data + weight gradient of a point-MLP layer
            # from one read of its operands): bound by the matrix pipe (12 bf16 MFMA products per fp32 multiply-add pair) and the
            # conversion VALU beside it, not by HBM -- reported against the pipe, with their (already minimal) HBM rate beside it
            out["roofline"]["lidar_layer_backward"] = {
                "kernel": "lidar_l2_bwd_kernel / lidar_l1_bwd_kernel (csrc/kd_lidar_bwd.hip)", "bound": "mfma",
                "launches_per_step": lb[3] // 2, "ms_per_step": round(lb[2] / 2 * 1e3, 2),
                "executed_bf16_tflops": round(6 * lb[0] / lb[2] / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                "frac": round(6 * lb[0] / lb[2] / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                "fp32_equivalent_tflops": round(lb[0] / lb[2] / 1e12, 1), "hbm_GBps_of_its_minimal_traffic": round(lb[1] / lb[2] / 1e9, 0),
                "replaces": "kd_lidar_l2_dgrad + _l2_wgrad + _l1_dgrad + _l1_wgrad: 21.8 ms and 94 GB of operand passes per step in round 2"}
        # HBM bytes per launch come from PMC counters, which need their own rocprofv3 passes: use the committed
        # measurement of this exact workload (profiles/), null for any other configuration
        try:   # matrix-pipe busy % of the camera GEMM kernels inside the step: a committed PMC measurement, tagged with its kernel hash
            mu = json.load(open(os.path.join(ROOT, "profiles", "r04_camera_gemm_mfma_busy.json")))
            out["roofline"]["mfma_busy_pct"] = mu["camera_gemm_mfma_busy_pct"]
            out["roofline"]["mfma_busy_detail"] = {k: mu[k] for k in ("by_kernel", "how", "kernel_hash") if k in mu}
            out["roofline"]["mfma_busy_is_current"] = mu.get("kernel_hash") == kernel_code_state()
        except (OSError, KeyError, ValueError):
            pass
        for fn in ("r04_bench_pmc_traffic_B256.json", "r03_bench_pmc_traffic_B256.json", "r02_bench_pmc_traffic_B256.json", "r01_bench_pmc_traffic_B256.json"):
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
                wl = pmc["workload"]
                same = ((wl["per_gpu_batch"], wl["points_per_frame"], wl["image"], wl["bev_grid"]) == (args.batch, args.points, args.image, args.grid)
                        and wl.get("student_fusion", "weighted") == args.student_fusion
                        and wl.get("teacher_fusion", "concat") == args.teacher_fusion)
                if same:
                    out["roofline"]["traffic"] = pmc["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_unit"] = f"bytes/launch (PMC FETCH_SIZE*2 + WRITE_SIZE, profiles/{fn})"
                    # PMC counters need their own rocprofv3 passes, so this figure is a committed measurement: say which
                    # kernel sources it was taken on and whether they are the ones running now
                    out["roofline"]["traffic_code_state"] = pmc.get("code_state")
                    out["roofline"]["traffic_kernel_hash"] = pmc.get("kernel_hash")
                    out["roofline"]["traffic_is_current"] = pmc.get("kernel_hash") == kernel_code_state()
                    break
            except (OSError, KeyError, ValueError):
                continue
    if rank == 0 and world == 1 and not args.no_bf16_forward:
        # The same KD step with the frozen teacher on the bf16-storage path (KDStep(teacher_storage="bf16")): a second
        # mode reported BESIDE the fp32 headline, never instead of it.  Student forward / backward / AdamW stay fp32.
        if reducer is not None:
            reducer.enabled = False                            # (--force-reducer: this second-mode step runs without it)
        step_b = KDStep(student, teacher, opt, torch.tensor([0.4, 3.5], device=dev), T=4.0, alpha=1.0, beta=1.0,
                        teacher_storage="bf16")
        nb = max(3, min(args.steps, 10))
        for _ in range(2):
            pb = step_b(images, pts, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nb):
            pb = step_b(images, pts, labels)
        torch.cuda.synchronize()
        tb = time.perf_counter() - t0
        out["kd_step_bf16_teacher"] = {
            "value": round(args.batch * nb / tb, 2), "unit": "frames/s", "ms_per_step": round(1e3 * tb / nb, 3), "steps": nb,
            "dtype": "teacher activations bf16 (fp32 accumulate); student forward/backward/optimiser f32",
            "vs_f32_step": round((args.batch * nb / tb) / out["value"], 3),
            "finite": bool(torch.isfinite(pb["total"]).item()),
            "note": "second mode (tests/test_gpu_bf16.py states its tolerance); not the headline"}
        del step_b
    if rank == 0 and world == 1 and not args.no_bf16_forward:
        del step
        torch.cuda.empty_cache()
        out["bf16_forward"] = bf16_forward_bench(args, dev, images, pts, max(3, min(args.steps, 10)))
    if rank == 0 and world == 1 and not args.no_side_benches:
        # configs[4] (fusion ablation under the same concat teacher) and the reference's own batch size (B=4: launch-bound;
        # eager and as one replayed hipGraph).  Bounded: 5 timed steps per student at the benchmarked batch, 20 at B=4.
        torch.cuda.empty_cache()
        ns = max(3, min(args.steps, 5))

        def guarded(*a, **k):
            # a side line that fails (e.g. out of memory beside the headline models on a smaller card) must not cost the headline line
            try:
                return side_rate(*a, **k)
            except Exception as e:                                 # noqa: BLE001 -- reported in the JSON, never swallowed silently
                torch.cuda.empty_cache()
                return {"error": f"{type(e).__name__}: {e}"[:300]}
        out["ablation"] = {"what": f"KD step frames/s per student fusion, {args.teacher_fusion} teacher, B={args.batch}, same box, {ns} timed steps each (configs[4])"}
        for sf in ("concat", "minimal", "weighted"):
            out["ablation"][sf] = guarded(args, dev, sf, args.batch, ns)
        out["small_batch"] = {"what": f"the reference's batch (B=4), {args.student_fusion} student, N={args.points} points: eager launches vs the whole step as one replayed hipGraph, 20 timed steps each",
                              "B4_eager": guarded(args, dev, args.student_fusion, 4, 20),
                              "B4_hipgraph": guarded(args, dev, args.student_fusion, 4, 20, graph=True)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(teacher, student, args.points, args.image, args.grid, args.teacher_fusion, args.student_fusion)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
