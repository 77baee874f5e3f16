# in-step A/B of the pipelined stride-1 depthwise forward (KD_DW_FWD_PIPE=0: segment-by-segment kernel everywhere), + the unit tests that cover it
mkdir -p gpurun_out/q4v
timeout -k 10 900 python -m pytest tests/test_gpu_units.py tests/test_gpu_headline.py -q -x > gpurun_out/q4v/tests.log 2>&1; tail -2 gpurun_out/q4v/tests.log
./tools/bench_dw 256 > gpurun_out/q4v/dw_pipe.txt 2>&1; KD_DW_FWD_PIPE=0 ./tools/bench_dw 256 > gpurun_out/q4v/dw_old.txt 2>&1
cut -c1-60 gpurun_out/q4v/dw_pipe.txt; cut -c1-60 gpurun_out/q4v/dw_old.txt
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-forward --no-side-benches --no-selfcheck"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/q4v/$name.json 2> gpurun_out/q4v/$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/q4v/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['frac'])"; }
run pipe A=1 &&
run old KD_DW_FWD_PIPE=0 &&
run pipe2 A=1 &&
run old2 KD_DW_FWD_PIPE=0
