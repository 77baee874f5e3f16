#!/bin/bash
set -u
O=gpurun_out/r3u; mkdir -p $O
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step tests 1100 python -m pytest tests/test_gpu_gemm_shapes.py tests/test_gpu_units.py tests/test_gpu_parity.py tests/test_gpu_lidar_segments.py -q -x -k "not fp64"
step bench 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-forward
tail -n 3 $O/tests.log; tail -n 1 $O/bench.log | cut -c1-200
