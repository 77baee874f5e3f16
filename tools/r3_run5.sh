#!/bin/bash
set -u
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 180 python -m pytest tests/test_gpu_gemm_shapes.py -q -x -k "fused_backward" > $O/unit.log 2>&1; echo "unit rc=$?"; tail -3 $O/unit.log
grep -q "passed" $O/unit.log || exit 1
timeout -k 10 300 python3 tools/bench_lidar_bwd.py 256 5 both > $O/time.log 2>&1 || exit 1
cat $O/time.log
for p in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"; do
  n=$(echo $p | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $p --output-format csv -d $O/pmc_$n -- python3 tools/bench_lidar_bwd.py 256 2 fused > $O/pmc_$n.log 2>&1 || { echo "pmc $n failed"; tail -5 $O/pmc_$n.log; }
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("gpurun_out/r3f/pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lidar_l2_bwd" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, s) in sorted(agg.items()):
    print(f"{k:28s} calls {n:3d} avg {s / n:.4g}")
PY
