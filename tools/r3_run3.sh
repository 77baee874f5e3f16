#!/bin/bash
set -u
O=gpurun_out/r3c; mkdir -p $O
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step fused_unit 180 python -m pytest tests/test_gpu_gemm_shapes.py -q -x -k "fused_backward"
grep -q "passed" $O/fused_unit.log || { tail -30 $O/fused_unit.log; exit 1; }
step lidar 600 python -m pytest tests/test_gpu_lidar_segments.py tests/test_gpu_full_size.py -q -x
step bench_sep 600 env KD_LIDAR_FUSED_BWD=0 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-bf16-forward --no-selfcheck
step bench_fused 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-bf16-forward --no-selfcheck
step rccl 600 python -m pytest tests/test_gpu_rccl_world1.py -q
step fp64 900 python -m pytest tests/test_gpu_parity.py -q -k "fp64 or every_student"
step rest 1000 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_units.py tests/test_gpu_bf16.py -q
tail -4 $O/*.log | cut -c1-600
