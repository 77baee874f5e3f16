# in-step A/B of the frozen teacher's fused (depthwise + 1x1) inference tails: KD_DW_PW_FUSED=1 (by shape, the default so far) vs 0 (never) vs 2 (always)
mkdir -p gpurun_out/q4w
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-forward --no-side-benches --no-selfcheck"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/q4w/$name.json 2> gpurun_out/q4w/$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/q4w/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['frac'])"; }
run shape KD_DW_PW_FUSED=1 &&
run never KD_DW_PW_FUSED=0 &&
run always KD_DW_PW_FUSED=2 &&
run shape2 KD_DW_PW_FUSED=1 &&
run never2 KD_DW_PW_FUSED=0
