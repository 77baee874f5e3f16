#!/usr/bin/env python3
"""Dev tool: forward / data-gradient GEMM launches of the last KD step of two rocprofv3 --kernel-trace CSVs, side by side
in launch order (same step structure, different kernel selection).  usage: trace_ab.py A_kernel_trace.csv B_kernel_trace.csv"""
import csv, re, sys


def load(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adamw_dev_kernel" in r["Kernel_Name"]]
    out = []
    for k in range(1, len(idx)):                      # average over the profiled steps (skip the first: warm-up)
        step = []
        for r in rows[idx[k - 1] + 1: idx[k] + 1]:
            n = r["Kernel_Name"]
            if "pw_gemm" in n or "pw_stream" in n:
                short = re.sub(r"\(anonymous namespace\)::|kd_stream::|void |\(GemmArgs\)", "", n)
                step.append((short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]),
                             (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        out.append(step)
    n = len(out)
    return [(s[0], s[1], s[2], sum(o[i][3] for o in out) / n) for i, s in enumerate(out[0])]


A, B = load(sys.argv[1]), load(sys.argv[2])
ta = tb = 0.0
for x, y in zip(A, B):
    flag = "" if x[0] == y[0] else ("  slower" if y[3] > x[3] * 1.03 else ("  faster" if y[3] < x[3] * 0.97 else "  same"))
    ta += x[3]; tb += y[3]
    print(f"{x[0]:40s} {x[1]:>7d}x{x[2]} {x[3]:9.1f} | {y[0]:46s} {y[1]:>5d}x{y[2]} {y[3]:9.1f}{flag}")
print(f"sum: {ta / 1e3:.2f} ms vs {tb / 1e3:.2f} ms")
