export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p gpurun_out
timeout 60 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "xcd or another_plan or matches_scipy or fanout or spmm3" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
timeout 90 python -m pytest tests/test_gpu_shapes.py -m gpu -q -x -p no:cacheprovider -k "xcd_share or yelp_shape" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
for C in 1 0; do
  SRH_XCD_CALIBRATE=$C timeout 60 python bench.py --steps 1300 --warmup 30 --no-cpu-baseline --no-eval --no-dropin > gpurun_out/bench_cal$C.json 2> gpurun_out/bench_cal$C.err
  python -c "
import json; d=json.loads(open('gpurun_out/bench_cal$C.json').read().strip().splitlines()[-1]); r=d['roofline']
print('calibrate=$C', d['ms_per_step'], d['steady_state']['ms_per_step'], d['value'], r['launch_us_by_flavour'])"
done
timeout 100 python -m pytest tests/test_gpu_engine.py tests/test_gpu_kernels.py tests/test_gpu_dropin.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
