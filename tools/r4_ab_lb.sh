# A/B of lidar_l2_bwd_kernel's epilogue owner: matrix waves (default) vs the round-3 stage tile + vector-wave epilogue (tools/dbg/lb_old)
mkdir -p gpurun_out/q4z
timeout -k 10 900 python -m pytest tests/test_gpu_lidar_segments.py tests/test_gpu_units.py tests/test_gpu_headline.py tests/test_gpu_gemm_shapes.py -q -x -k "lidar or headline or l2 or l1" > gpurun_out/q4z/tests.log 2>&1; tail -2 gpurun_out/q4z/tests.log
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-forward --no-side-benches --no-selfcheck"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B --dump-launches gpurun_out/q4z/launch_$name.txt > gpurun_out/q4z/$name.json 2> gpurun_out/q4z/$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/q4z/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['frac'])"; grep lidar_bwd gpurun_out/q4z/launch_$name.txt | head -2; }
run new A=1 &&
run old KD_HIP_LIB=tools/dbg/lb_old/libkd_hip.so &&
run new2 A=1 &&
run old2 KD_HIP_LIB=tools/dbg/lb_old/libkd_hip.so
