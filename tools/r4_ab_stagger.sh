# in-step A/B of the streaming kernels' staggered start (KD_STREAM_STAGGER builds under tools/dbg/stagN): same box, same process order
mkdir -p gpurun_out/q4n
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-forward --no-side-benches --no-selfcheck"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B --dump-launches gpurun_out/q4n/launch_$name.txt > gpurun_out/q4n/$name.json 2> gpurun_out/q4n/$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/q4n/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['frac'], r['by_group']['camera_fpn_fusion_head']['ms_per_step'], r['by_group']['lidar_point_mlp']['ms_per_step'])"; }
run base A=1 &&
run stag1 KD_HIP_LIB=tools/dbg/stag1/libkd_hip.so &&
run stag2 KD_HIP_LIB=tools/dbg/stag2/libkd_hip.so &&
run stag4 KD_HIP_LIB=tools/dbg/stag4/libkd_hip.so &&
run base2 A=1
