#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals and per-shape GEMM timings."""
import collections, csv, glob, sys
d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
st = glob.glob(d + "/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(st)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms over {steps:g} steps = {tot/1e6/steps:.2f} ms/step")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("kd_stream::", "").replace("void ", "")[:64]
    print(f'{float(r["TotalDurationNs"])/1e6/steps:8.3f} ms/step {100*float(r["TotalDurationNs"])/tot:5.1f}%  calls/step {float(r["Calls"])/steps:6.1f}  avg {float(r["AverageNs"])/1e3:9.1f} us  {n}')
tr = glob.glob(d + "/*/*_kernel_trace.csv")[0]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(tr)):
    n = r["Kernel_Name"]
    if "pw_gemm" in n or "pw_wgrad" in n or "pw_stream" in n:
        short = n.replace("(anonymous namespace)::", "").replace("kd_stream::", "").replace("void ", "").split("(")[0]
        k = (short, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1))
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("--- GEMM launches by (kernel, workgroups)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k[0]:50s} {k[1]:>6} WGs  calls/step {v[0]/steps:5.1f}  avg {v[1]/v[0]:9.1f} us  {v[1]/1e3/steps:7.3f} ms/step")
