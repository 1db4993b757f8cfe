#!/bin/bash
# On the GPU box: bench.py (training part only) once per library in tools/spmm_lab/alt/ named on the command line, the
# product's own library first and last.   usage: tools/spmm_lab/ab_libs.sh sr srh ...
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0; mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp selfrec_amd/lib/libselfrec_hip.so /tmp/orig.so
run() {
  timeout 300 python bench.py --steps 1300 --warmup 30 --no-cpu-baseline --no-eval --no-dropin > gpurun_out/bench_$1.json 2> gpurun_out/bench_$1.err; echo "bench $1 exit $?"
  python -c "
import json; d=json.loads(open('gpurun_out/bench_$1.json').read().strip().splitlines()[-1]); r=d['roofline']
print('$1', d['ms_per_step'], d['steady_state']['ms_per_step'], d['value'], r['launch_us_by_flavour'], d['final_losses'])"
}
run product
for N in "$@"; do cp tools/spmm_lab/alt/libselfrec_hip_$N.so selfrec_amd/lib/libselfrec_hip.so; run $N; done
cp /tmp/orig.so selfrec_amd/lib/libselfrec_hip.so
run product2
