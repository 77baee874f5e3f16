# the bench's bf16_forward block (frames/s, per-family ms) + the bf16 tests
mkdir -p gpurun_out/q4s
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -q > gpurun_out/q4s/tests.log 2>&1; tail -2 gpurun_out/q4s/tests.log
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-benches --no-selfcheck --no-roofline"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/q4s/$name.json 2> gpurun_out/q4s/$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/q4s/$name.json')); b=d['bf16_forward']; f=b['roofline']['by_family']; print('$name', b['value'], 'pw ms', f['bf16_pw']['ms'], 'GB/s', f['bf16_pw']['GB/s'], 'dw ms', f['bf16_dw']['ms'], f['bf16_dw']['GB/s'], 'lidar', f['bf16_lidar']['ms'], 'all', b['roofline']['all_bf16_kernels']['ms'], 'err', b['max_abs_logit_error_vs_fp32'], 'kd', d['kd_step_bf16_teacher']['value'])"; }
run a A=1 &&
run b A=1
