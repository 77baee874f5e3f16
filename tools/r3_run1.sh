#!/bin/bash
# round-3 GPU session 1: RCCL world-1 path, slab guard, seed scan, full GPU suite, bench (+ forced reducer)
set -u
O=gpurun_out/r3a; mkdir -p $O
step() { # name timeout cmd...
  local name=$1 t=$2; shift 2
  echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?
  echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi
}
step rccl 600 python -m pytest tests/test_gpu_rccl_world1.py -x -q
step guard 600 python -m pytest tests/test_gpu_gemm_shapes.py -x -q -k "refused or streaming"
step bench_forced 600 python bench.py --steps 5 --warmup 2 --force-reducer --no-cpu-baseline --no-bf16-forward --no-selfcheck
step scan 900 python tools/diag_fp64_seeds.py $O/seeds.json 1 14 2
step tests 1100 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_gradients_against_fp64_oracle --deselect tests/test_gpu_parity.py::test_fp64_gradient_check_goes_red_on_a_wrong_statistics_row_count
tail -3 $O/*.log
