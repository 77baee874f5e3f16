#!/usr/bin/env python3
"""Matrix-pipe / VALU / LDS utilisation of the GEMM kernels from three rocprofv3 --pmc passes over tools/bench_gemm
(pass 1: SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES; pass 2: SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES;
pass 3: SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS).  Rows are (kernel instance, grid size) averages.
GRBM_GUI_ACTIVE is summed over the 8 XCDs; busy-cycle counters over the 1024 SIMDs / 256 CUs."""
import collections, csv, glob, sys
def load(d):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(glob.glob(d + "/*/*_counter_collection.csv")[0])):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if "pw_gemm_kernel" not in n and "pw_wgrad_kernel" not in n:
            continue
        k = (n.split("(")[0], int(r["Grid_Size"]) // int(r["Workgroup_Size"]))
        a = agg[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    return agg
p1, p2, p3 = (load(d) for d in sys.argv[1:4])
print(f"{'kernel':44s} {'WGs':>7s} {'MFMA busy %':>11s} {'VALU inst-active %':>18s} {'LDS busy %':>10s} {'bank-conflict % of LDS':>22s} {'VALU insts/wave-cycle':>21s}")
for k in sorted(p1, key=lambda k: -p1[k]["GRBM_GUI_ACTIVE"][1]):
    avg = lambda P, c: P[k][c][1] / max(P[k][c][0], 1) if k in P and c in P[k] else float("nan")
    cyc = avg(p1, "GRBM_GUI_ACTIVE") / 8.0                     # per-XCD active cycles = kernel duration in cycles
    mf = 100 * avg(p1, "SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024)
    cyc2 = cyc
    va = 100 * avg(p2, "SQ_ACTIVE_INST_VALU") / (cyc2 * 1024)  # cycles a SIMD spends issuing/executing VALU (per SIMD)
    lds = 100 * avg(p3, "SQ_LDS_IDX_ACTIVE") / (cyc * 256)
    bc = 100 * avg(p3, "SQ_LDS_BANK_CONFLICT") / max(avg(p3, "SQ_LDS_IDX_ACTIVE"), 1)
    ipc = avg(p2, "SQ_INSTS_VALU") / max(avg(p2, "SQ_WAVE_CYCLES"), 1)
    print(f"{k[0]:44s} {k[1]:7d} {mf:11.1f} {va:18.1f} {lds:10.1f} {bc:22.1f} {ipc:21.3f}")
