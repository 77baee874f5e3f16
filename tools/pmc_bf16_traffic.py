#!/usr/bin/env python3
"""HBM traffic of the bf16-storage forward (tools/bf16_forward_only.py: 5 identical passes) from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE) plus the kernel-stats pass of the same program: per kernel family, per forward pass.
gfx950 corrections per MI355X_MICROARCH.md: both counters in KiB; FETCH_SIZE counts 64 B per 128-B request for wide reads -> x2.
usage: pmc_bf16_traffic.py <dir FETCH> <dir WRITE> <dir stats> [passes=5]"""
import collections, csv, glob, sys


def one(d, pat):
    return (glob.glob(d + "/*/" + pat) + glob.glob(d + "/" + pat))[0]


def fam(k):
    k = k.replace("(anonymous namespace)::", "")
    for key, name in (("pw_gemm_bf16", "bf16 1x1 conv GEMM"), ("dw_bf16", "bf16 depthwise 3x3"), ("lidar_mlp_scatter_infer_kernel<1>", "bf16 LiDAR encoder (one kernel)"),
                      ("stem_bf16", "bf16 stem"), ("bilinear_sum_bf16", "bf16 FPN resize + sum"), ("cls_bf16", "bf16 classifier"), ("weighted_tail_bf16", "bf16 weighted tail"),
                      ("stable_", "fp32 point sort"), ("scan_", "fp32 point sort"), ("bn_eval_coeffs", "BatchNorm eval coefficients")):
        if key in k:
            return name
    return "other (" + k.split("(")[0][:40] + ")"


def load(d, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(one(d, "*_counter_collection.csv"))):
        if r["Counter_Name"] == name:
            a = agg[fam(r["Kernel_Name"])]
            a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg


passes = float(sys.argv[4]) if len(sys.argv) > 4 else 5.0
fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
tm = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(one(sys.argv[3], "*_kernel_stats.csv"))):
    a = tm[fam(r["Name"])]
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
print(f"{'family':40s} {'launches/fwd':>12s} {'ms/fwd':>8s} {'HBM read MB/fwd':>16s} {'HBM write MB/fwd':>17s} {'GB/s (PMC bytes / kernel time)':>30s}")
tot = [0.0, 0.0, 0.0]
for k in sorted(tm, key=lambda k: -tm[k][1]):
    f = fe[k][1] * 1024 * 2 / passes / 1e6 if k in fe else 0.0
    w = wr[k][1] * 1024 / passes / 1e6 if k in wr else 0.0
    ms = tm[k][1] / passes / 1e6
    tot[0] += f; tot[1] += w; tot[2] += ms
    print(f"{k:40s} {tm[k][0]/passes:12.1f} {ms:8.3f} {f:16.1f} {w:17.1f} {(f + w) / max(ms, 1e-9):30.1f}")
print(f"{'all kernels of one forward':40s} {'':12s} {tot[2]:8.3f} {tot[0]:16.1f} {tot[1]:17.1f} {(tot[0] + tot[1]) / tot[2]:30.1f}")
