#!/usr/bin/env python3
"""Matrix-pipe / VALU / LDS utilisation of the GEMM-class kernels (tiled, streaming, fused LiDAR) from three rocprofv3 --pmc
passes over the same program (tools/bench_gemm, or bench.py for the kernels of a real KD step)
(pass 1: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; pass 2: SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES;
pass 3: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS).  Rows are (kernel instance, grid size) averages.
GRBM_GUI_ACTIVE is summed over the 8 XCDs; busy-cycle counters over the 1024 SIMDs / 256 CUs."""
import collections, csv, glob, sys
def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open((glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0])):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("kd_stream::", "").replace("void ", "")
        if not any(t in n for t in ("pw_gemm_kernel", "pw_wgrad_kernel", "pw_wgrad_rs_kernel", "pw_stream_kernel", "lidar_l", "lidar_mlp", "pw_gemm_bf16", "dw_")):
            continue
        k = (n.split("(")[0], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":          # once per dispatch: its duration under the counter pass
            t = agg[k]["_ns"]
            t[0] += 1; t[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg
p1, p2, p3 = (load(d) for d in sys.argv[1:4])
p4 = load(sys.argv[4]) if len(sys.argv) > 4 else {}
print("# clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration of the same pass; wait columns (pass 4, optional) are shares of SQ_WAVE_CYCLES:")
print("# a wave waiting on anything (s_waitcnt, barrier, dependency) / on an instruction issue slot / on an LDS instruction")
print(f"{'kernel':44s} {'WGs':>7s} {'us':>8s} {'GHz':>5s} {'MFMA busy %':>11s} {'VALU inst-active %':>18s} {'LDS busy %':>10s} {'bank-conflict % of LDS':>22s} {'VALU insts/wave-cycle':>21s} {'wait any %':>10s} {'wait inst %':>11s} {'wait LDS %':>10s}")
for k in sorted(p1, key=lambda k: -p1[k]["GRBM_GUI_ACTIVE"][1]):
    avg = lambda P, c: P[k][c][1] / max(P[k][c][0], 1) if k in P and c in P[k] else float("nan")
    cyc = avg(p1, "GRBM_GUI_ACTIVE") / 8.0                     # per-XCD active cycles = kernel duration in cycles
    mf = 100 * avg(p1, "SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024)
    cyc2 = cyc
    va = 100 * avg(p2, "SQ_ACTIVE_INST_VALU") / (cyc2 * 1024)  # cycles a SIMD spends issuing/executing VALU (per SIMD)
    lds = 100 * avg(p3, "SQ_LDS_IDX_ACTIVE") / (cyc * 256)
    bc = 100 * avg(p3, "SQ_LDS_BANK_CONFLICT") / max(avg(p3, "SQ_LDS_IDX_ACTIVE"), 1)
    ipc = avg(p2, "SQ_INSTS_VALU") / max(avg(p2, "SQ_WAVE_CYCLES"), 1)
    us = avg(p1, "_ns") / 1e3
    wc = max(avg(p4, "SQ_WAVE_CYCLES"), 1) if p4 else float("nan")
    w = [100 * avg(p4, c) / wc if p4 else float("nan") for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS")]
    print(f"{k[0]:44s} {k[1]:7d} {us:8.1f} {cyc / max(us, 1e-9) / 1e3:5.2f} {mf:11.1f} {va:18.1f} {lds:10.1f} {bc:22.1f} {ipc:21.3f} {w[0]:10.1f} {w[1]:11.1f} {w[2]:10.1f}")
