# in-step A/B of the four-channel ("tall") BatchNorm finalize: KD_BN_TALL=0 (never) vs the default (>= 1024 rows) vs 2048
mkdir -p gpurun_out/q4p
python -m pytest tests/test_gpu_units.py -q -k "bn_finalize" > gpurun_out/q4p/tests.log 2>&1; tail -2 gpurun_out/q4p/tests.log
B="python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-bf16-forward --no-side-benches --no-selfcheck"
run() { name=$1; shift; env "$@" timeout -k 10 200 $B > gpurun_out/q4p/$name.json 2> gpurun_out/q4p/$name.err; python -c "
import json,sys; d=json.load(open('gpurun_out/q4p/$name.json')); r=d['roofline']; print('$name', d['value'], d['ms_per_step'], r['frac'])"; }
run tall A=1 &&
run short KD_BN_TALL=0 &&
run tall2048 KD_BN_TALL=2048 &&
run tall512 KD_BN_TALL=512 &&
run tall_again A=1 &&
run short_again KD_BN_TALL=0
