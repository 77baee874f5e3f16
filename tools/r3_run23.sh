#!/bin/bash
set -u
O=gpurun_out/r4d; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step tests1 600 python -m pytest tests/test_gpu_units.py -q -x
step tests2 1000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_units.py
step bench 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-forward
step benchA 600 env KD_CHAIN_PAIRS=0 KD_DW_BWD_ADD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-forward --no-selfcheck --no-roofline
step benchB 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-forward --no-selfcheck --no-roofline
tail -n 4 $O/tests1.log; tail -n 6 $O/tests2.log; for f in bench benchA benchB; do tail -n 1 $O/$f.log | cut -c1-160; done
