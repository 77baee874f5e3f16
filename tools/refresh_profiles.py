#!/usr/bin/env python3
"""Dev tool: turn one evidence run (rocprofv3 stats dir, the two PMC dirs, the bench JSON of the stats run) into the committed
profiles/r02_* files.  usage: refresh_profiles.py <run dir with stats/, pmc_f/, pmc_w/, stats_bench.json> "code state text" """
import csv, glob, json, os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
E, state = sys.argv[1], sys.argv[2]
for d in ("stats", "pmc_f", "pmc_w"):                      # prof_summary / pmc tools expect <dir>/<sub>/*.csv
    os.makedirs(f"{E}/{d}/x", exist_ok=True)
    for f in glob.glob(f"{E}/{d}/p_*.csv"):
        shutil.move(f, f"{E}/{d}/x/")
summ = subprocess.run([sys.executable, f"{ROOT}/tools/prof_summary.py", f"{E}/stats", "6"], capture_output=True, text=True, check=True).stdout
rows = list(csv.DictReader(open(f"{E}/stats/x/p_kernel_stats.csv")))
tot = lambda sel: (sum(int(r["Calls"]) for r in rows if sel(r["Name"])), sum(float(r["TotalDurationNs"]) for r in rows if sel(r["Name"])))
fc, ft = tot(lambda n: "pw_gemm_kernel" in n or "pw_stream_kernel" in n)
wc, wt = tot(lambda n: "pw_wgrad_kernel" in n)
_, dt = tot(lambda n: "dw_" in n and "dw_pw_infer" not in n)
_, zt = tot(lambda n: "dw_pw_infer" in n)
s = json.load(open(f"{E}/stats_bench.json"))
r = s["roofline"]
open(f"{ROOT}/profiles/r02_bench_kernel_summary_B256.txt", "w").write(f"""rocprofv3 --kernel-trace --stats of the default bench.py workload, round-2 final code (MI355X, ROCm 7.2)
command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck --no-bf16-forward
(256 frames x 80 000 points per step; 6 steps = 1 warm-up + 3 timed + 2 roofline steps; full CSV: r02_bench_kernel_stats_B256.csv)
bench.py line of the same run: {s['value']} frames/s, {s['ms_per_step']} ms/step, roofline.frac {r['frac']}, live HIP-event average of the
forward / data-gradient GEMM family {r['avg_launch_us']} us over {r['launches_per_step']} launches per step.
rocprof, same family (pw_gemm_kernel + pw_stream_kernel, all instances): {fc} calls, {ft / fc / 1e3:.1f} us average, {ft / 6e6:.2f} ms/step.
weight-gradient GEMMs (pw_wgrad_kernel): {wc} calls, {wt / 6e6:.2f} ms/step.  depthwise kernels (dw_*): {dt / 6e6:.2f} ms/step (round 1: 16.4)
+ {zt / 6e6:.2f} ms/step for the two fused inference tails (dw_pw_infer_kernel, which contain their 1x1 convolutions).

{summ}""")
shutil.copy(f"{E}/stats/x/p_kernel_stats.csv", f"{ROOT}/profiles/r02_bench_kernel_stats_B256.csv")
out = subprocess.run([sys.executable, f"{ROOT}/tools/pmc_bench_traffic.py", f"{E}/pmc_f", f"{E}/pmc_w", f"{ROOT}/profiles/r02_bench_pmc_traffic_B256.json", state],
                     capture_output=True, text=True, check=True).stdout
pj = json.load(open(f"{ROOT}/profiles/r02_bench_pmc_traffic_B256.json"))
alg = r["algorithmic_mbyte_per_launch"]
open(f"{ROOT}/profiles/r02_bench_pmc_traffic_B256.txt", "w").write(f"""HBM traffic per launch inside bench.py (B=256 frames x 80 000 points, split arithmetic), round-2 final code state
two separate passes: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (each with --kernel-trace only) over
`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-selfcheck --no-bf16-forward`, joined by tools/pmc_bench_traffic.py
(KiB -> bytes; FETCH_SIZE x2: gfx950 counts 64 B per 128-B request on wide streaming reads)
family = pw_gemm_kernel (tiled) + pw_stream_kernel (weight-resident streaming): the dispatcher picks one per shape
({r['launches_per_step']} launches per step: the teacher's stage-2 / stage-3 project convolutions run inside the fused inference tails, kd_block.hip)

{out}
algorithmic bytes of the fwd/dgrad family in the same workload (bench.py roofline.algorithmic_mbyte_per_launch): {alg} MB/launch
=> measured / algorithmic = {pj['hbm_bytes_per_launch'] / 1e6 / alg:.3f} (round 1, tiled kernels only: 1.025).  Writes are at the algorithmic figure; the extra
reads: the column-tiled streaming launches (N = 192 / 384 / 768 as 3 / 3 / 6 tiles) stream A once per tile and not all of it hits L2,
and the LiDAR layer-2 data gradient gathers its two per-cell tables (537 MB each) per point row.
""")
dec = subprocess.run([sys.executable, f"{ROOT}/tools/step_decomposition.py", f"{ROOT}/profiles/r02_bench_kernel_stats_B256.csv", "6"], capture_output=True, text=True, check=True).stdout
open(f"{ROOT}/profiles/r02_step_decomposition.txt", "w").write("One KD step (256 frames x 80 000 points, concat teacher -> weighted student) by kernel family: rocprofv3 --kernel-trace --stats of the default bench.py,\nround-2 final code (profiles/r02_bench_kernel_stats_B256.csv: run total / 6 steps; the 'runtime copies' line is mostly the one-time model upload).\ntools/step_decomposition.py profiles/r02_bench_kernel_stats_B256.csv 6\n\n" + dec)
print(f"family {fc} calls avg {ft / fc / 1e3:.1f} us (live {r['avg_launch_us']}); wgrad {wt / 6e6:.2f}; dw {dt / 6e6:.2f} + {zt / 6e6:.2f}; total {sum(float(x['TotalDurationNs']) for x in rows) / 6e6:.2f} ms/step")
print(out, "ratio", pj["hbm_bytes_per_launch"] / 1e6 / alg, "hash", pj["kernel_hash"])
