# A/B of the stride-1 depthwise backward tile kernel with / without the next item's first rows prefetched (tools/dbg/tile_old = without)
mkdir -p gpurun_out/q4x
timeout -k 10 900 python -m pytest tests/test_gpu_units.py -q -x -k "dw_" > gpurun_out/q4x/tests.log 2>&1; tail -2 gpurun_out/q4x/tests.log
./tools/bench_dw 256 > gpurun_out/q4x/dw_new.txt 2>&1; LD_PRELOAD=$PWD/tools/dbg/tile_old/libkd_hip.so ./tools/bench_dw 256 > gpurun_out/q4x/dw_old.txt 2>&1
./tools/bench_dw 256 > gpurun_out/q4x/dw_new2.txt 2>&1
for f in dw_new dw_old dw_new2; do echo $f; cut -c108-220 gpurun_out/q4x/$f.txt; done
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-forward --no-side-benches --no-selfcheck"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/q4x/$name.json 2> gpurun_out/q4x/$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/q4x/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['frac'])"; }
run new A=1 &&
run old KD_HIP_LIB=tools/dbg/tile_old/libkd_hip.so &&
run new2 A=1 &&
run old2 KD_HIP_LIB=tools/dbg/tile_old/libkd_hip.so
