# A/B of the pipelined depthwise forward: KD_DW_FWD_PIPE bit 0 = stride 1, bit 1 = stride 2 (default 3); isolated table + in-step
mkdir -p gpurun_out/q4v
timeout -k 10 900 python -m pytest tests/test_gpu_units.py tests/test_gpu_headline.py -q -x > gpurun_out/q4v/tests.log 2>&1; tail -2 gpurun_out/q4v/tests.log
KD_DW_FWD_PIPE=3 ./tools/bench_dw 256 > gpurun_out/q4v/dw_3.txt 2>&1; KD_DW_FWD_PIPE=1 ./tools/bench_dw 256 > gpurun_out/q4v/dw_1.txt 2>&1; KD_DW_FWD_PIPE=3 ./tools/bench_dw 256 > gpurun_out/q4v/dw_3b.txt 2>&1
for f in dw_3 dw_1 dw_3b; do echo $f; cut -c1-52 gpurun_out/q4v/$f.txt | grep " s2 "; done
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-forward --no-side-benches --no-selfcheck"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/q4v/$name.json 2> gpurun_out/q4v/$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/q4v/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['frac'])"; }
run p3 KD_DW_FWD_PIPE=3 &&
run p1 KD_DW_FWD_PIPE=1 &&
run p3b KD_DW_FWD_PIPE=3 &&
run p1b KD_DW_FWD_PIPE=1
