mkdir -p gpurun_out/q4j
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-forward --no-side-benches --no-selfcheck"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B --dump-launches gpurun_out/q4j/launch_$name.txt > gpurun_out/q4j/$name.json 2> gpurun_out/q4j/$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/q4j/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['frac'], r['by_group']['camera_fpn_fusion_head']['ms_per_step'], r['by_group']['lidar_point_mlp']['ms_per_step'])"; }
run tstore A=1
run dword KD_HIP_LIB=tools/dbg/dword/libkd_hip.so
run stream1 KD_GEMM_STREAM=1
run stream0 KD_GEMM_STREAM=0
run tstore2 A=1
