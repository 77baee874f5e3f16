#!/bin/bash
# usage: r3_prof.sh <tag>: kernel-trace + stats of the default bench workload -> gpurun_out/<tag>/
O=gpurun_out/$1; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-selfcheck --no-bf16-forward --no-side-benches > $O/stats_bench.log 2>&1
grep '^{' $O/stats_bench.log | tail -1 > $O/stats_bench.json
ls $O/stats | head
ls $O/stats/*/p_kernel_trace.csv >/dev/null 2>&1 || { mkdir -p $O/stats/x; mv $O/stats/p_*.csv $O/stats/x/; }
python3 tools/prof_summary.py $O/stats > $O/summary.txt 2>&1
KEEP=$(sed -n 's/.*the last \([0-9]*\) kept.*/\1/p' $O/summary.txt | head -1)
python3 tools/step_decomposition.py $(ls $O/stats/*/p_kernel_stats_timed.csv | head -1) ${KEEP:-1} > $O/decomp.txt 2>&1
head -5 $O/summary.txt; cat $O/decomp.txt
