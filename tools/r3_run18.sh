#!/bin/bash
set -u
O=gpurun_out/r3y; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step tests 1100 python -m pytest tests -m gpu -q
step bench 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-bf16-forward
step dw256 300 ./tools/bench_dw 256
step dw32 120 ./tools/bench_dw 32
bash tools/r3_prof.sh r3y > $O/prof.log 2>&1
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-bf16-forward --no-roofline"
step pmc4 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $O/pmc4 -o p -- $B
tail -n 3 $O/tests.log; tail -n 1 $O/bench.log | cut -c1-200; head -1 $O/summary.txt; cat $O/decomp.txt; cat $O/dw256.log
