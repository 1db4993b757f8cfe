#!/bin/bash
# the driver's literal flags, several times on one box: how much a 20-step region placed over an epoch boundary moves from run to run
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; : > gpurun_out/driver20_repeat.txt
for i in $(seq 1 ${REPEAT:-5}); do
  SRH_BENCH_STEP_EVENTS=${STEP_EVENTS:-1} timeout 300 python bench.py --gpus 1 --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-eval --no-dropin 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('run $i: value', d['value'], 'ms_per_step', d['ms_per_step'], 'boundaries', d['config']['epoch_boundaries_in_region'], d['config'].get('epoch_boundary_host_ms'), d['config'].get('step_us'), '| steady', d['steady_state']['ms_per_step'], d['value_steady_state'])" | tee -a gpurun_out/driver20_repeat.txt
done
