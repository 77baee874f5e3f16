#!/bin/bash
# round-end validation: full GPU suite, PMC traffic passes on the final kernels, default bench line
set -u
O=gpurun_out/fin4; mkdir -p $O
export TMPDIR=/tmp
step() { local name=$1 t=$2; shift 2; echo "=== $name" | tee -a $O/steps.log
  timeout -k 10 $t "$@" > $O/$name.log 2>&1; local rc=$?; echo "rc=$rc" | tee -a $O/steps.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name: stopping" | tee -a $O/steps.log; exit 1; fi; }
step tests 1100 python -m pytest tests -m gpu -q
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-selfcheck --no-bf16-forward --no-roofline --no-side-benches"
step pmcF 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcF -o p -- $B
step pmcW 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcW -o p -- $B
step smoke 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
step bench 900 python bench.py
tail -n 5 $O/tests.log; tail -n 2 $O/smoke.log; tail -n 1 $O/bench.log | cut -c1-200
