#!/usr/bin/env python3
"""Dev tool: one KD step by kernel family from a rocprofv3 kernel-stats CSV (run total / steps).
usage: step_decomposition.py <p_kernel_stats.csv> <steps>"""
import collections, csv, sys

FAMILIES = [
    ("1x1 conv / point MLP: forward + data gradient (pw_stream / pw_gemm)", ("pw_stream_kernel", "pw_gemm_kernel")),
    ("1x1 conv / point MLP: weight gradient (pw_wgrad / pw_wgrad_rs + slab reduce)", ("pw_wgrad_kernel", "pw_wgrad_rs_kernel", "wgrad_reduce")),
    ("depthwise 3x3 forward / backward (dw_*)", ("dw_fwd", "dw_bwd")),
    ("fused inference tails (dw_pw_infer)", ("dw_pw_infer",)),
    ("LiDAR: point sort, layer-0 passes, scatter-max forward / backward (seg_*, lidar_*, sort)", ("seg_", "lidar_", "sort_", "scan_", "stable_", "rank_", "hist", "scatter", "gather_sorted", "bev_")),
    ("BatchNorm: statistics reductions, finalize, apply (bn_*, slab_*)", ("bn_", "slab_")),
    ("FPN resize + sum, fusion attention, classifier (bilinear_*, weighted_fuse_*, cls_*)", ("bilinear", "weighted_fuse", "cls_", "col2im", "convT")),
    ("stem conv (forward, im2col for its weight gradient)", ("stem_",)),
    ("losses, metrics, AdamW (seg_loss_*, mse_*, adamw_*, confusion)", ("seg_loss", "mse_", "adamw", "confusion", "argmax")),
    ("ATen element-wise (autograd gradient accumulation, scalar loss arithmetic)", ("at::native",)),
    ("runtime copies / fills (model upload, memsets)", ("rocclr",)),
    ("transposes of weights for the data gradient", ("transpose",)),
]
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
agg = collections.OrderedDict((f[0], [0.0, 0]) for f in FAMILIES)
other = [0.0, 0, []]
for r in rows:
    n = r["Name"]
    for label, keys in FAMILIES:
        if any(k in n for k in keys) and not (label.startswith("LiDAR") and "seg_loss" in n):
            agg[label][0] += float(r["TotalDurationNs"]); agg[label][1] += int(r["Calls"])
            break
    else:
        other[0] += float(r["TotalDurationNs"]); other[1] += int(r["Calls"]); other[2].append(n[:50])
tot = sum(v[0] for v in agg.values()) + other[0]
print(f"{'family':100s} {'ms/step':>8s} {'share':>6s} {'launches/step':>14s}")
for label, (t, c) in agg.items():
    print(f"{label:100s} {t / steps / 1e6:8.2f} {100 * t / tot:5.1f}% {c / steps:14.1f}")
print(f"{'other':100s} {other[0] / steps / 1e6:8.2f} {100 * other[0] / tot:5.1f}% {other[1] / steps:14.1f}   {sorted(set(other[2]))[:6]}")
print(f"{'total kernel time':100s} {tot / steps / 1e6:8.2f}")
