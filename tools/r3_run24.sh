#!/bin/bash
set -u
O=gpurun_out/r4e; mkdir -p $O
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step tests1 600 python -m pytest tests/test_gpu_units.py -q
tail -n 4 $O/tests1.log
