#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals and per-shape GEMM timings of the TIMED
steps only.  usage: prof_summary.py <dir> [keep]
The run is cut into steps at `adamw_tick_kernel` (exactly one launch per KD step, its last kernel); only the trailing
`keep` steps are kept (default: all but the first two -- the first step allocates and uploads, the second still sizes
workspaces), so warm-up launches (model upload copies, first-use memsets, cache fills) do not reach the table.  Writes the
windowed per-kernel table next to the trace as p_kernel_stats_timed.csv (same columns tools/step_decomposition.py reads)."""
import collections, csv, glob, os, sys
d = sys.argv[1]
tr = glob.glob(d + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(tr)), key=lambda r: int(r["Start_Timestamp"]))
ticks = [int(r["End_Timestamp"]) for r in rows if "adamw_tick_kernel" in r["Kernel_Name"]]
if len(ticks) < 2:
    sys.exit(f"{len(ticks)} step boundaries (adamw_tick_kernel) in {tr}: cannot window")
keep = int(float(sys.argv[2])) if len(sys.argv) > 2 else max(1, len(ticks) - 2)
keep = min(keep, len(ticks) - 1)
lo, hi = ticks[-keep - 1], ticks[-1]
win = [r for r in rows if lo < int(r["Start_Timestamp"]) and int(r["End_Timestamp"]) <= hi]
steps = float(keep)
per = collections.defaultdict(lambda: [0, 0])
for r in win:
    p = per[r["Kernel_Name"]]
    p[0] += 1
    p[1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in per.values())
out = os.path.join(os.path.dirname(tr), "p_kernel_stats_timed.csv")
with open(out, "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
    for n, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
        w.writerow([n, v[0], v[1], v[1] / v[0], 100.0 * v[1] / tot])
idx = [i for i, r in enumerate(rows) if "adamw_tick_kernel" in r["Kernel_Name"]]
print("launches per step (cut at adamw_tick_kernel):", [b - a for a, b in zip(idx[:-1], idx[1:])],
      "-- bench.py's two HIP-event-profiled roofline steps and their first-use allocations are among the kept ones")
print(f"{len(ticks)} steps in the trace, the last {keep} kept (warm-up dropped): {len(win)} launches = {len(win)/steps:.1f} launches/step, "
      f"wall {(hi - lo)/1e6/steps:.2f} ms/step, kernel time {tot/1e6/steps:.2f} ms/step")
short = lambda n: n.replace("(anonymous namespace)::", "").replace("kd_stream::", "").replace("void ", "")
for n, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:48]:
    print(f"{v[1]/1e6/steps:8.3f} ms/step {100*v[1]/tot:5.1f}%  calls/step {v[0]/steps:6.1f}  avg {v[1]/v[0]/1e3:9.1f} us  {short(n)[:72]}")
aten = [(short(n), v) for n, v in per.items() if "at::native" in n and v[1] / v[0] >= 20e3]
print("--- at::native kernels averaging >= 20 us:", "none" if not aten else "")
for n, v in aten:
    print(f"    calls/step {v[0]/steps:5.1f}  avg {v[1]/v[0]/1e3:8.1f} us  {n[:110]}")
agg = collections.defaultdict(lambda: [0, 0.0])
for r in win:
    n = r["Kernel_Name"]
    if "pw_gemm" in n or "pw_wgrad" in n or "pw_stream" in n or "lidar_l" in n or "lidar_mlp" in n:
        k = (short(n).split("(")[0], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) * int(r.get("Grid_Size_Y", 1) or 1))
        agg[k][0] += 1
        agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
print("--- GEMM launches by (kernel, workgroups)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{k[0]:50s} {k[1]:>6} WGs  calls/step {v[0]/steps:5.1f}  avg {v[1]/v[0]:9.1f} us  {v[1]/1e3/steps:7.3f} ms/step")
