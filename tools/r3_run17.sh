#!/bin/bash
set -u
O=gpurun_out/r3x; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step tests1 600 python -m pytest tests/test_gpu_kd_objective.py tests/test_gpu_bf16.py tests/test_gpu_rccl_world1.py -q -x
step tests2 900 python -m pytest tests/test_gpu_units.py tests/test_gpu_parity.py -q -x -k "not fp64"
step bench 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-forward
bash tools/r3_prof.sh r3x > $O/prof.log 2>&1
tail -n 3 $O/tests1.log; tail -n 3 $O/tests2.log; tail -n 1 $O/bench.log | cut -c1-200; head -3 $O/summary.txt; cat $O/decomp.txt
