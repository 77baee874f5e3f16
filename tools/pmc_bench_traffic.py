#!/usr/bin/env python3
"""HBM traffic per launch of the GEMM kernels inside bench.py, from two separate rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE) of the same command.  gfx950 corrections per MI355X_MICROARCH.md: both counters
are in KiB; FETCH_SIZE counts 64 B per 128-B request for wide streaming reads -> x2.
The fwd/dgrad family is pw_gemm_kernel (tiled form) + pw_stream_kernel (weight-resident streaming form): the
dispatcher picks one per shape, bench.py's roofline covers both.
usage: pmc_bench_traffic.py <dir FETCH_SIZE pass> <dir WRITE_SIZE pass> [out.json "code state text"]"""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def family(k):
    if "pw_gemm_kernel" in k or "pw_stream_kernel" in k:
        return "pw_gemm+pw_stream"
    if "pw_wgrad_kernel" in k or "pw_wgrad_rs_kernel" in k:
        return "pw_wgrad_kernel"
    if "lidar_l2_bwd_kernel" in k or "lidar_l1_bwd_kernel" in k:
        return "lidar_bwd"
    if "lidar_mlp_scatter_infer" in k:
        return "lidar_infer"
    if "dw_fwd" in k or "dw_bwd" in k:
        return "depthwise"
    return None


def load(d, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open((glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0])):
        if r["Counter_Name"] != name:
            continue
        fam = family(r["Kernel_Name"])
        if fam:
            agg[fam][0] += 1
            agg[fam][1] += float(r["Counter_Value"])
    return agg


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
res = {}
for fam in ("pw_gemm+pw_stream", "pw_wgrad_kernel", "lidar_bwd", "lidar_infer", "depthwise"):
    if fam not in fe:
        continue
    n = fe[fam][0]
    assert n == wr[fam][0] and n > 0, (fam, fe[fam], wr[fam])
    f = fe[fam][1] * 1024 * 2 / n
    w = wr[fam][1] * 1024 / n
    res[fam] = (n, f, w)
    print(f"{fam:18s} launches {n:5d}  FETCH_SIZEx2 {f/1e6:9.2f} MB/launch  WRITE_SIZE {w/1e6:9.2f} MB/launch  HBM {(f+w)/1e6:9.2f} MB/launch")
if len(sys.argv) > 3:
    sys.path.insert(0, ROOT)
    import bench
    n, f, w = res["pw_gemm+pw_stream"]
    json.dump({
        "workload": {"per_gpu_batch": 256, "points_per_frame": 80000, "image": 256, "bev_grid": 64,
                     "teacher_fusion": "concat", "student_fusion": "weighted"},
        "kernel": "pw_gemm_kernel + pw_stream_kernel (1x1-conv forward and data gradient)", "launches": n,
        "fetch_size_x2_bytes_per_launch": round(f), "write_size_bytes_per_launch": round(w), "hbm_bytes_per_launch": round(f + w),
        "how": "two separate rocprofv3 passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE, each with --kernel-trace only) over "
               "`python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-selfcheck --no-bf16-forward`, joined by "
               "tools/pmc_bench_traffic.py; KiB -> bytes, FETCH_SIZE x2 (gfx950 wide streaming reads)",
        "arithmetic": "split (default)", "code_state": sys.argv[4] if len(sys.argv) > 4 else "",
        "kernel_hash": bench.kernel_code_state()}, open(sys.argv[3], "w"), indent=1)
