#!/bin/bash
set -u
O=gpurun_out/r3p; mkdir -p $O
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step unit 240 python -m pytest tests/test_gpu_gemm_shapes.py -q -x -k "fused_backward"
grep -q "passed" $O/unit.log || { tail -30 $O/unit.log; exit 1; }
step lidar 600 python -m pytest tests/test_gpu_lidar_segments.py tests/test_gpu_full_size.py tests/test_gpu_units.py -q -x -k "lidar or full or twolite or model"
step bench 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-forward
tail -3 $O/unit.log $O/lidar.log; tail -1 $O/bench.log | cut -c1-300
