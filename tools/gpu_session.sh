#!/bin/bash
# One gpurun call = tests + smoke + bench + profiles; everything lands in gpurun_out/.
# usage: tools/gpu_session.sh [tests] [smoke] [bench] [prof] [pmc]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT
STAGES="${*:-smoke tests bench prof}"
echo "== stages: $STAGES"; rocm-smi --showproductname 2>/dev/null | grep -m2 -i "card\|gfx" ; nproc
for s in $STAGES; do
case $s in
smoke)
  timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -5 $OUT/smoke.log;;
tests)
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests exit $?"
  grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -40;;
bench)
  timeout 900 python bench.py --steps ${BENCH_STEPS:-1300} --warmup 50 > $OUT/bench.log 2> $OUT/bench.err; echo "bench exit $?"
  tail -3 $OUT/bench.err; tail -2 $OUT/bench.log;;
benchquick)
  timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline > $OUT/benchquick.log 2> $OUT/benchquick.err; echo "benchquick exit $?"
  tail -3 $OUT/benchquick.err; tail -2 $OUT/benchquick.log;;
eager)
  timeout 600 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-eval --no-graph > $OUT/bench_eager.log 2> $OUT/bench_eager.err; echo "eager exit $?"
  tail -3 $OUT/bench_eager.err; tail -2 $OUT/bench_eager.log;;
profgraph)
  rm -rf $OUT/profgraph; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/profgraph -o trace -- python $OLDPWD/bench.py --steps 600 --warmup 50 --no-cpu-baseline --no-eval > $OLDPWD/$OUT/profgraph.log 2>&1); echo "profgraph exit $?"
  f=$(find $OUT/profgraph -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py "$f" > $OUT/profgraph_kernel_stats.txt && head -16 $OUT/profgraph_kernel_stats.txt; tail -1 $OUT/profgraph.log | cut -c1-300
  find $OUT/profgraph -name "*.db" -size +40M -delete;;
profeval)
  rm -rf $OUT/profeval; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/profeval -o trace -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$OUT/profeval.log 2>&1); echo "profeval exit $?"
  f=$(find $OUT/profeval -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py "$f" | grep -E "kernel  |gemm_nt|topk|mask_kernel|cand_|hit_flags|Memset|fill" > $OUT/profeval_kernel_stats.txt; cat $OUT/profeval_kernel_stats.txt
  find $OUT/profeval -name "*.db" -size +40M -delete;;
prof)
  rm -rf $OUT/prof; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-eval --no-graph > $OLDPWD/$OUT/prof.log 2>&1); echo "prof exit $?"
  f=$(find $OUT/prof -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py "$f" > $OUT/prof_kernel_stats.txt && head -24 $OUT/prof_kernel_stats.txt
  find $OUT/prof -name "*.db" -size +40M -delete;;
gather)
  mkdir -p $OUT; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_bw.hip -o /tmp/gather_bw && timeout 300 /tmp/gather_bw > $OUT/gather_bw.log 2>&1; echo "gather exit $?"; cat $OUT/gather_bw.log;;
big)
  timeout 1500 python tools/big_graph.py > $OUT/big_graph.log 2>&1; echo "big exit $?"; grep -v amdgpu.ids $OUT/big_graph.log | tail -8;;
evalpmc)
  rm -rf $OUT/evalpmc; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $OLDPWD/$OUT/evalpmc -o pmc -- python $OLDPWD/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OLDPWD/$OUT/evalpmc.log 2>&1); echo "evalpmc exit $?"
  f=$(find $OUT/evalpmc -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_summary.py "$f" | grep -E "gemm_nt|topk|mask_kernel|nce_tile" | tee $OUT/evalpmc_summary.txt;;
sharded1)
  SRH_FORCE_SHARDED=1 timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench_sharded1.log 2> $OUT/bench_sharded1.err; echo "sharded1 exit $?"; tail -3 $OUT/bench_sharded1.err; tail -1 $OUT/bench_sharded1.log | cut -c1-400;;
colsprof)
  # per-kernel times of one rank's share of a column-sharded step (virtual rank 0 of G, stand-in communicator)
  for G in 2 4 8; do
    rm -rf $OUT/prof_cols$G; (cd /tmp && COLS_PROBE_WORLDS=$G COLS_PROBE_MODES=eager timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof_cols$G -o trace -- python $OLDPWD/tools/cols_probe.py > $OLDPWD/$OUT/prof_cols$G.log 2>&1); echo "colsprof $G exit $?"
    f=$(find $OUT/prof_cols$G -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py "$f" | grep -v "at::native\|rocclr\|normalize_kernel\|degree_kernel" > $OUT/prof_cols${G}_kernel_stats.txt && head -16 $OUT/prof_cols${G}_kernel_stats.txt
    find $OUT/prof_cols$G -name "*.db" -size +30M -delete
  done;;
profcols1)
  rm -rf $OUT/profcols1; (cd /tmp && SRH_FORCE_SHARDED=1 SRH_SHARD_LAYOUT=cols timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/profcols1 -o trace -- python $OLDPWD/bench.py --steps 300 --warmup 30 --no-cpu-baseline > $OLDPWD/$OUT/profcols1.log 2>&1); echo "profcols1 exit $?"
  f=$(find $OUT/profcols1 -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py "$f" > $OUT/profcols1_kernel_stats.txt && head -24 $OUT/profcols1_kernel_stats.txt; tail -1 $OUT/profcols1.log | cut -c1-300
  find $OUT/profcols1 -name "*.db" -size +30M -delete;;
sharded1cols)
  SRH_FORCE_SHARDED=1 SRH_SHARD_LAYOUT=cols timeout 600 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench_sharded1cols.log 2> $OUT/bench_sharded1cols.err; echo "sharded1cols exit $?"; tail -3 $OUT/bench_sharded1cols.err; tail -1 $OUT/bench_sharded1cols.log | cut -c1-700;;
zipf)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_zipf.hip -o /tmp/gather_zipf 2>/dev/null && timeout 300 /tmp/gather_zipf > $OUT/gather_zipf.log 2>&1; echo "zipf exit $?"; cat $OUT/gather_zipf.log;;
matrix)
  timeout 900 python tools/model_matrix.py > $OUT/model_matrix.log 2>&1; echo "matrix exit $?"; cat $OUT/model_matrix.log | grep -v amdgpu.ids;;
nce)
  timeout 600 python tools/nce_ab.py > $OUT/nce_ab.log 2>&1; echo "nce exit $?"; cat $OUT/nce_ab.log;;
ncepmc)
  for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-40); rm -rf $OUT/ncepmc_$tag
    (cd /tmp && SRH_NCE_SPLITS=8 timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OLDPWD/$OUT/ncepmc_$tag -o pmc -- python $OLDPWD/tools/nce_ab.py child 2048 > $OLDPWD/$OUT/ncepmc_$tag.log 2>&1); echo "ncepmc exit $?"
    f=$(find $OUT/ncepmc_$tag -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_summary.py "$f" | grep -i "nce_" | head -8
  done;;
cols)
  timeout 600 python -m pytest tests/test_gpu_cols.py -q --tb=short -x -p no:cacheprovider > $OUT/tests_cols.log 2>&1; echo "cols tests exit $?"
  tail -30 $OUT/tests_cols.log;;
colsprobe)
  timeout 600 python tools/cols_probe.py > $OUT/cols_probe.log 2>&1; echo "colsprobe exit $?"; grep -v amdgpu.ids $OUT/cols_probe.log
  SRH_SPMM_THIN=1 COLS_PROBE_KERNELS_ONLY=1 timeout 600 python tools/cols_probe.py > $OUT/cols_probe_thin.log 2>&1; echo "colsprobe (lane-per-row A/B) exit $?"; grep spmm $OUT/cols_probe_thin.log;;
ab)
  timeout 600 python tools/spmm_ab.py > $OUT/spmm_ab.log 2>&1; echo "ab exit $?"; tail -45 $OUT/spmm_ab.log;;
pmcdense)
  # dense-flavour SpMM only: FETCH_SIZE / WRITE_SIZE (KB per dispatch) in separate passes, then the json bench.py reads
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $c | tr ' ' '_'); rm -rf $OUT/pmcdense_$tag
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OLDPWD/$OUT/pmcdense_$tag -o pmc -- python $OLDPWD/tools/spmm_pmc.py > $OLDPWD/$OUT/pmcdense_$tag.log 2>&1); echo "pmcdense $c exit $?"
    f=$(find $OUT/pmcdense_$tag -name "*.db" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" --tail 40 | grep spmm_rows > $OUT/pmcdense_$tag.summary.txt; cat $OUT/pmcdense_$tag.summary.txt
    find $OUT/pmcdense_$tag -name "*.db" -size +40M -delete
  done;;
pmccols)
  # SpMM launches of the column-sharded layout: HBM-side traffic and issue counters, one counter group per pass
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-40); rm -rf $OUT/pmccols_$tag
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OLDPWD/$OUT/pmccols_$tag -o pmc -- python $OLDPWD/tools/spmm_pmc_cols.py > $OLDPWD/$OUT/pmccols_$tag.log 2>&1); echo "pmccols $c exit $?"
    f=$(find $OUT/pmccols_$tag -name "*.db" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" --tail 30 | grep "spmm_" > $OUT/pmccols_$tag.summary.txt; cat $OUT/pmccols_$tag.summary.txt
    find $OUT/pmccols_$tag -name "*.db" -size +30M -delete
  done;;
pmc)
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $c | tr ' ' '_'); rm -rf $OUT/pmc_$tag
    (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c -d $OLDPWD/$OUT/pmc_$tag -o pmc -- python $OLDPWD/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-eval --no-graph > $OLDPWD/$OUT/pmc_$tag.log 2>&1); echo "pmc $c exit $?"
    f=$(find $OUT/pmc_$tag -name "*.db" | head -1)
    [ -n "$f" ] && python tools/pmc_summary.py "$f" > $OUT/pmc_$tag.summary.txt 2>&1 && head -12 $OUT/pmc_$tag.summary.txt
    find $OUT/pmc_$tag -name "*.db" -size +40M -delete
  done;;
esac
done
echo "== done"
