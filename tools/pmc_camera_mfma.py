#!/usr/bin/env python3
"""Matrix-pipe busy % of the CAMERA-side GEMM kernels inside one KD step (bench.py's `roofline.mfma_busy_pct`).
usage: pmc_camera_mfma.py <dir of the rocprofv3 pass `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES`> OUT.json "code state"
Camera-side = every dispatch of the 1x1-convolution GEMM kernels (pw_gemm / pw_stream / pw_wgrad / pw_wgrad_rs) shorter than
1.5 ms at the benchmarked batch -- the point-MLP launches of the same kernels run 2.4 ms or more.  busy % = sum of
SQ_VALU_MFMA_BUSY_CYCLES / sum of (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs), i.e. time-weighted over those dispatches."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d, out, state = sys.argv[1], sys.argv[2], sys.argv[3]
f = (glob.glob(d + "/*/*_counter_collection.csv") + glob.glob(d + "/*_counter_collection.csv"))[0]
disp = collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("kd_stream::", "").replace("void ", "").split("(")[0]
    if not any(t in n for t in ("pw_gemm_kernel", "pw_wgrad_kernel", "pw_stream_kernel", "pw_wgrad_rs_kernel", "pw_rs_kernel")):
        continue
    e = disp[r["Dispatch_Id"]]
    e["name"] = n.split("<")[0]
    e[r["Counter_Name"]] = float(r["Counter_Value"])
    e["us"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
by = collections.defaultdict(lambda: [0.0, 0.0, 0.0, 0])
for e in disp.values():
    if e["us"] >= 1500.0 or "GRBM_GUI_ACTIVE" not in e:
        continue
    b = by[e["name"]]
    b[0] += e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); b[1] += e["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0; b[2] += e["us"]; b[3] += 1
tot_b, tot_c = sum(b[0] for b in by.values()), sum(b[1] for b in by.values())
import bench
res = {"camera_gemm_mfma_busy_pct": round(100.0 * tot_b / max(tot_c, 1.0), 1),
       "by_kernel": {k: {"mfma_busy_pct": round(100.0 * b[0] / max(b[1], 1.0), 1), "dispatches": b[3], "ms": round(b[2] / 1e3, 3),
                         "sustained_GHz": round(b[1] / 1024.0 / max(b[2], 1e-9) / 1e3, 2)} for k, b in sorted(by.items())},
       "how": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES over `python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "
              "--no-selfcheck --no-bf16-forward --no-roofline --no-side-benches`; GEMM-kernel dispatches under 1.5 ms (camera / FPN / fusion / head); tools/pmc_camera_mfma.py",
       "code_state": state, "kernel_hash": bench.kernel_code_state()}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
