#!/bin/bash
# One gpurun call = a list of stages; everything lands in gpurun_out/.
# usage: tools/gpu_session.sh [stage ...]      stages: smoke tests testsall testsf4 testsnew rfab detab bench benchquick benchdriver benchbig benchworld2 prof profeval evalprobe evalab evalpmc lossprobe nceab nceprec pmc big cols oplevel refmodels refmodelsfuse determinism precision lab*
# (budget note from round 2: a call is charged for getting the box as well as for the run -- 20 s when a warm box is at hand,
#  3-5 min when not, whatever the command: batch stages into one call, and keep the last minutes for a final check)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT
STAGES="${*:-smoke tests bench}"
echo "== stages: $STAGES"; rocm-smi --showproductname 2>/dev/null | grep -m2 -i "card\|gfx"; nproc
stats() { f=$(find "$1" -name "*.db" | head -1); [ -n "$f" ] && python tools/rocpd_stats.py "$f"; }
for s in $STAGES; do
t0=$(date +%s)
case $s in
smoke)
  timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -3 $OUT/smoke.log;;
tests)
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests exit $?"
  grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -30;;
testsall)
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/tests.log 2>&1; echo "tests exit $?"
  grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -40;;
testsf4)
  timeout 900 python -m pytest tests/test_gpu_f4.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/tests_f4.log 2>&1; echo "testsf4 exit $?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/tests_f4.log | head -40;;
testsnew)
  timeout 1500 python -m pytest tests/test_gpu_shapes.py -m gpu -q --tb=short -p no:cacheprovider > $OUT/tests_new.log 2>&1; echo "testsnew exit $?"
  grep -E "^(FAILED|ERROR)|passed|failed|Error|assert" $OUT/tests_new.log | tail -40;;
lab)
  timeout 900 python tools/spmm_lab/run.py ${LAB_ARGS:-} > $OUT/spmm_lab.log 2>&1; echo "lab exit $?"; grep -v amdgpu.ids $OUT/spmm_lab.log | tail -${LAB_TAIL:-20};;
lab3)
  timeout 900 python tools/spmm_lab/run.py --ids first-appearance --no-colclass > $OUT/spmm_lab_nocc.log 2>&1; echo "lab3 exit $?"; grep -v amdgpu.ids $OUT/spmm_lab_nocc.log | tail -${LAB_TAIL:-20};;
lab2)
  timeout 900 python tools/spmm_lab/run.py --ids first-appearance > $OUT/spmm_lab_fa.log 2>&1; echo "lab2 exit $?"; grep -v amdgpu.ids $OUT/spmm_lab_fa.log | tail -${LAB_TAIL:-20};;
labprobe)
  timeout 300 python tools/spmm_lab/run.py --ids first-appearance --probe > $OUT/lab_probe.txt 2>&1; echo "labprobe exit $?"; grep -v amdgpu.ids $OUT/lab_probe.txt | tail -${LAB_TAIL:-60};;
labplans)
  timeout 300 python tools/spmm_lab/run.py --ids first-appearance --plans > $OUT/lab_plans.txt 2>&1; echo "labplans exit $?"; grep -v amdgpu.ids $OUT/lab_plans.txt | tail -${LAB_TAIL:-14};;
labbalance)
  timeout 300 python tools/spmm_lab/run.py --ids first-appearance --balance --balance-flavour ${BALANCE_FLAVOUR:-dense} > $OUT/lab_balance.txt 2>&1; echo "labbalance exit $?"; grep -v amdgpu.ids $OUT/lab_balance.txt | tail -10;;
labgap)
  timeout 300 python tools/spmm_lab/run.py --ids first-appearance --gap > $OUT/lab_gap.txt 2>&1; echo "labgap exit $?"; grep -v amdgpu.ids $OUT/lab_gap.txt | tail -8;;
labbits)
  timeout 400 python tools/spmm_lab/run.py --ids first-appearance --class-bits ${CLASS_BITS:-0,1,2,3,4,6,9} > $OUT/lab_classbits.txt 2>&1; echo "labbits exit $?"; grep -v amdgpu.ids $OUT/lab_classbits.txt | tail -12;;
gatherlds)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/gather_lds.hip -o /tmp/gather_lds.out && timeout 120 /tmp/gather_lds.out > $OUT/gather_lds.txt 2>&1; echo "gatherlds exit $?"; cat $OUT/gather_lds.txt;;
evalprobe)
  timeout 600 python tools/eval_probe.py > $OUT/eval_probe.log 2>&1; echo "evalprobe exit $?"; grep -v amdgpu.ids $OUT/eval_probe.log | tail -3;;
lossprobe)
  rm -rf $OUT/lossprobe; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/lossprobe -o trace -- python $OLDPWD/tools/loss_probe.py > $OLDPWD/$OUT/lossprobe.log 2>&1); echo "lossprobe exit $?"
  stats $OUT/lossprobe | grep -E "kernel  |bpr_|nce_" > $OUT/lossprobe_kernel_stats.txt; cat $OUT/lossprobe_kernel_stats.txt; tail -1 $OUT/lossprobe.log
  find $OUT/lossprobe -name "*.db" -size +40M -delete;;
lossprobef32)
  rm -rf $OUT/lossprobef32; (cd /tmp && LOSS_PROBE_PRECISION=f32 timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/lossprobef32 -o trace -- python $OLDPWD/tools/loss_probe.py > $OLDPWD/$OUT/lossprobef32.log 2>&1); echo "lossprobef32 exit $?"
  stats $OUT/lossprobef32 | grep -E "kernel  |bpr_|nce_" > $OUT/lossprobef32_kernel_stats.txt; cat $OUT/lossprobef32_kernel_stats.txt; tail -1 $OUT/lossprobef32.log
  find $OUT/lossprobef32 -name "*.db" -size +40M -delete;;
losspmc)
  # counters of the loss section's kernels (one --pmc group per pass), per-dispatch means
  : > $OUT/loss_pmc.txt
  for G in "${LOSSPMC_A:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE}" "${LOSSPMC_B:-SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE}"; do
    rm -rf $OUT/losspmc; (cd /tmp && LOSS_PROBE_ITERS=30 LOSS_PROBE_PRECISION=${LOSSPMC_PRECISION:-f32} timeout 400 rocprofv3 --kernel-trace --pmc $G -d $OLDPWD/$OUT/losspmc -o pmc -- python $OLDPWD/tools/loss_probe.py > $OLDPWD/$OUT/losspmc.log 2>&1); echo "losspmc exit $?"
    f=$(find $OUT/losspmc -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_summary.py "$f" | grep -E "nce_|bpr_" | tee -a $OUT/loss_pmc.txt
  done
  rm -rf $OUT/losspmc;;
mfmavalu)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/microbench/mfma_f32_valu.hip -o /tmp/mfma_f32_valu.out && timeout 120 /tmp/mfma_f32_valu.out > $OUT/mfma_f32_valu.txt 2>&1; echo "mfmavalu exit $?"; cat $OUT/mfma_f32_valu.txt;;
ncef32ab)
  # the f32 InfoNCE passes under other compile-time switches (ALT_SRC=losses tools/spmm_lab/build_alt.sh <name> "<flags>")
  cp selfrec_amd/lib/libselfrec_hip.so /tmp/orig.so
  : > $OUT/ncef32ab.txt
  for N in ${NCE_LIBS:-base}; do
    [ "$N" = base ] || cp tools/spmm_lab/alt/libselfrec_hip_$N.so selfrec_amd/lib/libselfrec_hip.so
    rm -rf $OUT/ncef32ab_$N; (cd /tmp && LOSS_PROBE_ITERS=100 LOSS_PROBE_PRECISION=f32 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/ncef32ab_$N -o trace -- python $OLDPWD/tools/loss_probe.py > $OLDPWD/$OUT/ncef32ab_$N.log 2>&1); echo "ncef32ab $N exit $?"
    echo "== $N" >> $OUT/ncef32ab.txt; stats $OUT/ncef32ab_$N | grep -E "nce_tile" | tee -a $OUT/ncef32ab.txt
    rm -rf $OUT/ncef32ab_$N
    cp /tmp/orig.so selfrec_amd/lib/libselfrec_hip.so
  done;;
blockedab)
  # column-blocked / relabelled propagation product at the 1 M x 500 k, d = 128 shape (tools/spmm_blocked_ab.py)
  timeout ${BLOCKEDAB_TIMEOUT:-1500} python tools/spmm_blocked_ab.py > $OUT/spmm_blocked_ab.txt 2>&1; echo "blockedab exit $?"; grep -v amdgpu.ids $OUT/spmm_blocked_ab.txt | tail -30;;
blockedpmc)
  # fabric traffic + L2 hit rate of chosen variants: BLOCKED_PMC="ids:full ids:blocks6 relabel:full"
  : > $OUT/spmm_blocked_pmc.txt
  for V in ${BLOCKED_PMC:-ids:full}; do
    for G in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum WRITE_SIZE"; do
      rm -rf $OUT/blockedpmc; (cd /tmp && SPMM_AB_ONLY=$V SPMM_AB_ITERS=3 timeout 900 rocprofv3 --kernel-trace --pmc $G -d $OLDPWD/$OUT/blockedpmc -o pmc -- python $OLDPWD/tools/spmm_blocked_ab.py > $OLDPWD/$OUT/blockedpmc.log 2>&1); echo "blockedpmc $V [$G] exit $?"
      f=$(find $OUT/blockedpmc -name "*.db" | head -1); echo "== $V" >> $OUT/spmm_blocked_pmc.txt; [ -n "$f" ] && python tools/pmc_summary.py "$f" | grep -E "spmm_rows" | tee -a $OUT/spmm_blocked_pmc.txt
    done
  done
  rm -rf $OUT/blockedpmc;;
ncestamps)
  cp selfrec_amd/lib/libselfrec_hip.so /tmp/orig.so; cp tools/spmm_lab/alt/libselfrec_hip_stamps.so selfrec_amd/lib/libselfrec_hip.so
  timeout 300 python tools/nce_stamps.py 2>&1 | grep -v amdgpu.ids > $OUT/nce_stamps.txt; echo "ncestamps exit $?"; cat $OUT/nce_stamps.txt
  cp /tmp/orig.so selfrec_amd/lib/libselfrec_hip.so;;
ncemodeab)
  timeout 600 python tools/nce_mode_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/nce_mode_ab.txt; echo "ncemodeab exit $?"; cat $OUT/nce_mode_ab.txt;;
fuseadamab)
  timeout 600 python tools/fuse_adam_ab.py 2>&1 | grep -v amdgpu.ids > $OUT/fuse_adam_ab.txt; echo "fuseadamab exit $?"; cat $OUT/fuse_adam_ab.txt;;
fuseadamlibs)
  # the same A/B per alt library (FUSE_LIBS="nopf ..."), the product's own library first and last
  cp selfrec_amd/lib/libselfrec_hip.so /tmp/orig.so
  for N in product ${FUSE_LIBS:-nopf} product2; do
    case $N in product*) cp /tmp/orig.so selfrec_amd/lib/libselfrec_hip.so;; *) cp tools/spmm_lab/alt/libselfrec_hip_$N.so selfrec_amd/lib/libselfrec_hip.so;; esac
    echo "== $N"; AB_REPS=${AB_REPS:-3} timeout 600 python tools/fuse_adam_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/fuse_adam_ab_$N.txt | tail -5
  done
  cp /tmp/orig.so selfrec_amd/lib/libselfrec_hip.so;;
boundarytrace)
  rm -rf $OUT/btrace; (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $OLDPWD/$OUT/btrace -o trace -- python $OLDPWD/tools/boundary_region.py > $OLDPWD/$OUT/boundary_region.log 2>&1); echo "boundarytrace exit $?"
  grep -v amdgpu.ids $OUT/boundary_region.log | tail -6
  f=$(find $OUT/btrace -name "*.db" | head -1); [ -n "$f" ] && python tools/boundary_gaps.py "$f" ${GAP_US:-25} ${GAP_LAST:-40000} | tail -60 | tee $OUT/boundary_gaps.txt
  find $OUT/btrace -name "*.db" -size +30M -delete;;
hostcost)
  timeout 600 python tools/host_cost_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/host_cost_probe.txt;;
testsk)
  # TESTS_K="expr" TESTS_FILES="tests/a.py tests/b.py"
  timeout 1500 python -m pytest ${TESTS_FILES:-tests} -m gpu -q --tb=short -p no:cacheprovider -k "${TESTS_K:-infonce}" > $OUT/tests_k.log 2>&1; echo "testsk exit $?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/tests_k.log | head -60;;
testsnce)
  timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "infonce" > $OUT/tests_nce.log 2>&1; echo "testsnce exit $?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/tests_nce.log | head -40;;
nceab)
  # InfoNCE shape constants: per alt library (tools/spmm_lab/alt/libselfrec_hip_<name>.so, built by
  # ALT_SRC=losses tools/spmm_lab/build_alt.sh ...) the loss section's per-kernel times + its error against float64
  cp selfrec_amd/lib/libselfrec_hip.so /tmp/orig.so
  for N in ${NCE_LIBS:-s8pv3 s8pv6 s16pv3 s16pv6}; do
    cp tools/spmm_lab/alt/libselfrec_hip_$N.so selfrec_amd/lib/libselfrec_hip.so
    rm -rf $OUT/nceab_$N; (cd /tmp && LOSS_PROBE_ITERS=100 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/nceab_$N -o trace -- python $OLDPWD/tools/loss_probe.py > $OLDPWD/$OUT/nceab_$N.log 2>&1); echo "nceab $N exit $?"
    echo "== $N" >> $OUT/nceab.txt; stats $OUT/nceab_$N | grep -E "nce_|bpr_" | tee -a $OUT/nceab.txt
    timeout 200 python tools/nce_precision.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/nceab.txt
    find $OUT/nceab_$N -name "*.db" -delete
  done
  cp /tmp/orig.so selfrec_amd/lib/libselfrec_hip.so;;
evalab)
  # eval kernels per alt library (ALT_SRC=eval tools/spmm_lab/build_alt.sh <name> "<flags>"): per-kernel times of eval_probe
  cp selfrec_amd/lib/libselfrec_hip.so /tmp/orig.so; : > $OUT/evalab.txt
  for N in product ${EVAL_LIBS:-}; do
    [ $N = product ] || cp tools/spmm_lab/alt/libselfrec_hip_$N.so selfrec_amd/lib/libselfrec_hip.so
    rm -rf $OUT/evalab_$N; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/evalab_$N -o trace -- python $OLDPWD/tools/eval_probe.py > $OLDPWD/$OUT/evalab_$N.log 2>&1); echo "evalab $N exit $?"
    echo "== $N" >> $OUT/evalab.txt; stats $OUT/evalab_$N | grep -E "filter16|rescore|topk_kernel|split_rows" >> $OUT/evalab.txt; grep -v amdgpu.ids $OUT/evalab_$N.log | tail -1 >> $OUT/evalab.txt
    find $OUT/evalab_$N -name "*.db" -delete
    cp /tmp/orig.so selfrec_amd/lib/libselfrec_hip.so
  done
  cat $OUT/evalab.txt;;
determinism)
  timeout 600 python tools/determinism_probe.py > $OUT/determinism.txt 2>&1; echo "determinism exit $?"; grep -v amdgpu.ids $OUT/determinism.txt | tail -12;;
nceprec)
  timeout 300 python tools/nce_precision.py > $OUT/nce_precision.txt 2>&1; echo "nceprec exit $?"; grep -v amdgpu.ids $OUT/nce_precision.txt;;
evaltrainedprof)
  # per-kernel times of the ranking on TRAINED tables (the kernels of the 1300 training steps are in the same trace: ignore them)
  rm -rf $OUT/evaltrained; (cd /tmp && EVAL_TRAIN_STEPS=1300 timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/evaltrained -o trace -- python $OLDPWD/tools/eval_breakdown.py > $OLDPWD/$OUT/evaltrained.log 2>&1); echo "evaltrainedprof exit $?"
  stats $OUT/evaltrained | grep -E "kernel  |filter16|rescore|bound_rows|mask_kernel|split_rows|item_norms|invert_order|trim_mark|DeviceRadix|radix|onesweep|histogram|hit_flags" | tee $OUT/evaltrained_kernel_stats.txt; grep -v amdgpu.ids $OUT/evaltrained.log | tail -1
  find $OUT/evaltrained -name "*.db" -size +30M -delete;;
testseval)
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_shapes.py -m gpu -q --tb=short -p no:cacheprovider -k "rank or filter or topk or score or eval or metric or ties or heap" > $OUT/tests_eval.log 2>&1; echo "testseval exit $?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/tests_eval.log | head -40;;
hostprobe)
  timeout 300 python tools/oplevel_host_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/oplevel_host_probe${HOSTPROBE_TAG:-}.txt; echo "hostprobe exit $?"; cat $OUT/oplevel_host_probe${HOSTPROBE_TAG:-}.txt;;
testslosses)
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_dropin.py -m gpu -q --tb=short -p no:cacheprovider -k "loss or dropin or client or fast or reference or embedding" > $OUT/tests_losses.log 2>&1; echo "testslosses exit $?"
  grep -E "^(FAILED|ERROR)|passed|failed|^E  " $OUT/tests_losses.log | head -40;;
oplevel)
  # per-statement step time of the op-level models + whole 2-epoch runs through main
  : > $OUT/oplevel.txt
  for m in ${OPLEVEL_MODELS:-BUIR MixGCF DirectAU SelfCF}; do timeout 200 python tools/oplevel_probe.py $m 30 2>&1 | grep -v amdgpu.ids >> $OUT/oplevel.txt; done
  PROBE_ATEN_RAND=1 timeout 200 python tools/oplevel_probe.py BUIR 15 2>&1 | grep -v amdgpu.ids >> $OUT/oplevel.txt
  mkdir -p /tmp/conf_op
  for m in ${OPLEVEL_MODELS:-BUIR MixGCF DirectAU SelfCF}; do
    sed "s/^max.epoch:.*/max.epoch: 2/" conf/$m.yaml > /tmp/conf_op/$m.yaml
    timeout 400 python -m selfrec_amd.main $m --conf /tmp/conf_op/$m.yaml --synthetic yelp2018 > $OUT/main_$m.log 2>&1
    echo "$m: 2 epochs through selfrec_amd.main, exit $?, $(grep 'Running time' $OUT/main_$m.log)" >> $OUT/oplevel.txt
  done
  cat $OUT/oplevel.txt;;
refmodelsfuse)
  timeout 1500 python tools/run_reference_models.py --ref _refstage --fuse --models ${REF_MODELS:-XSimGCL,LightGCN,SimGCL,SGL} > $OUT/refmodels_fuse.log 2>&1; echo "refmodelsfuse exit $?"
  grep -E "^#|parity|epoch\(s\)|byte-for-byte|Error|error" $OUT/refmodels_fuse.log | tail -20;;
precision)
  timeout 900 python tools/precision_probe.py > $OUT/precision_probe.log 2>&1; echo "precision exit $?"; grep -v amdgpu.ids $OUT/precision_probe.log | tail -8;;
bench)
  timeout 1200 python bench.py --steps ${BENCH_STEPS:-1300} --warmup 50 > $OUT/bench.log 2> $OUT/bench.err; echo "bench exit $?"
  tail -3 $OUT/bench.err; tail -1 $OUT/bench.log | cut -c1-3000;;
benchworld2)
  # every line of bench.py's N > 1 path with N > 1 real ranks -- on ONE GPU, over gloo (figures meaningless)
  for lay in ${W2_LAYOUTS:-auto}; do
    # (auto: the headline = fixed-batch partition, then the dp sub-record; W2_LAYOUTS="auto rows" to force others first)
    SRH_DIST_BACKEND=gloo:device SRH_SHARD_LAYOUT=$lay timeout 900 python bench.py --gpus ${W2_RANKS:-2} --steps ${W2_STEPS:-100} --warmup 10 --no-cpu-baseline > $OUT/bench_world2_$lay.log 2> $OUT/bench_world2_$lay.err
    echo "benchworld2 $lay exit $?"; grep -v "Gloo\|socket.cpp\|amdgpu.ids" $OUT/bench_world2_$lay.err | tail -5; tail -1 $OUT/bench_world2_$lay.log | cut -c1-1500
  done;;
benchdriver)
  # what the driver runs: default flags
  timeout 1200 python bench.py > $OUT/bench_default.log 2> $OUT/bench_default.err; echo "benchdriver exit $?"
  tail -3 $OUT/bench_default.err; tail -1 $OUT/bench_default.log | cut -c1-600;;
benchbig)
  # BASELINE.json configs[3] at N = 1: the full line with roofline + traffic (profiles/spmm_dense_traffic_1m-500k_d128.json)
  timeout 1200 python bench.py --shape 1m-500k --emb 128 --steps 40 --warmup 5 > $OUT/bench_1m500k.log 2> $OUT/bench_1m500k.err; echo "benchbig exit $?"
  tail -3 $OUT/bench_1m500k.err; tail -1 $OUT/bench_1m500k.log | cut -c1-2500;;
benchquick)
  timeout 600 python bench.py --steps 600 --warmup 30 --no-cpu-baseline --no-eval --no-dropin > $OUT/benchquick.log 2> $OUT/benchquick.err; echo "benchquick exit $?"
  tail -3 $OUT/benchquick.err; tail -1 $OUT/benchquick.log | cut -c1-1500;;
benchab)
  for flag in "" "--no-overlap"; do
    timeout 600 python bench.py --steps 1300 --warmup 30 --no-cpu-baseline --no-eval --no-dropin $flag > $OUT/benchab.log 2> $OUT/benchab.err; echo "benchab [$flag] exit $?"
    python -c "import json,sys; d=json.loads(open('$OUT/benchab.log').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['steady_state']['ms_per_step'], d['value'])"
  done;;
refedited)
  # the tier an EDITED / new model file gets: copies of the reference's files with one comment appended (no SHA match),
  # with and without the host-side fast paths
  timeout 900 python tools/run_reference_models.py --ref _refstage --edited --models ${REF_MODELS:-XSimGCL,LightGCN} > $OUT/refmodels_edited.log 2>&1; echo "refedited exit $?"
  grep -E "^#|parity|epoch\(s\)|Error|error" $OUT/refmodels_edited.log | tail -12
  timeout 900 python tools/run_reference_models.py --ref _refstage --edited --no-fast --models ${REF_MODELS:-XSimGCL,LightGCN} > $OUT/refmodels_edited_nofast.log 2>&1; echo "refedited (no fast paths) exit $?"
  grep -E "^#|parity|epoch\(s\)|Error|error" $OUT/refmodels_edited_nofast.log | tail -12;;
refmodels)
  timeout 1500 python tools/run_reference_models.py --ref _refstage --models ${REF_MODELS:-XSimGCL,LightGCN,SimGCL,SGL} > $OUT/refmodels.log 2>&1; echo "refmodels exit $?"
  grep -E "^#|parity|1 epoch|Error|error" $OUT/refmodels.log | tail -20;;
bprobe)
  timeout 900 python tools/b_probe.py > $OUT/b_probe.log 2>&1; echo "bprobe exit $?"; grep -v amdgpu.ids $OUT/b_probe.log | tail -70;;
splitprobe)
  timeout 600 python tools/split_probe.py > $OUT/split_probe.log 2>&1; echo "splitprobe exit $?"; grep -v amdgpu.ids $OUT/split_probe.log | tail -12;;
refprof)
  timeout 900 python tools/run_reference_models.py --ref _refstage --models ${REF_MODELS:-XSimGCL} --profile 40 ${REFPROF_ARGS:-} > $OUT/refprof.log 2>&1; echo "refprof exit $?"
  grep -v "amdgpu.ids\|^training" $OUT/refprof.log | cut -c1-230 | tail -80;;
prof)
  # per-kernel times of the captured step (hipGraph replay), the command bench.py times
  rm -rf $OUT/prof; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/prof -o trace -- python $OLDPWD/bench.py --steps 600 --warmup 50 --no-cpu-baseline --no-eval --no-dropin > $OLDPWD/$OUT/prof.log 2>&1); echo "prof exit $?"
  stats $OUT/prof > $OUT/prof_kernel_stats.txt && head -20 $OUT/prof_kernel_stats.txt; tail -1 $OUT/prof.log | cut -c1-300
  find $OUT/prof -name "*.db" -size +40M -delete;;
profeval)
  rm -rf $OUT/profeval; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/profeval -o trace -- python $OLDPWD/bench.py --steps ${PROFEVAL_STEPS:-5} --warmup 2 --no-cpu-baseline --no-dropin > $OLDPWD/$OUT/profeval.log 2>&1); echo "profeval exit $?"
  stats $OUT/profeval | grep -E "kernel  |gemm_nt|filter16|rescore|topk|mask_kernel|split_rows|metric_rows|cand_|hit_flags" > $OUT/profeval_kernel_stats.txt; cat $OUT/profeval_kernel_stats.txt
  find $OUT/profeval -name "*.db" -size +40M -delete;;
pmc)
  # HBM-side traffic + L2 hit rate + issue counters of the dense SpMM launch (separate passes, counters only)
  for c in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    tag=$(echo $c | tr ' ' '_' | cut -c1-40); rm -rf $OUT/pmc_$tag
    (cd /tmp && timeout ${PMC_TIMEOUT:-300} rocprofv3 --kernel-trace --pmc $c -d $OLDPWD/$OUT/pmc_$tag -o pmc -- python $OLDPWD/tools/spmm_pmc.py ${PMC_ARGS:-} > $OLDPWD/$OUT/pmc_$tag.log 2>&1); echo "pmc [$c] exit $?"
    f=$(find $OUT/pmc_$tag -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_summary.py "$f" --tail ${PMC_TAIL:-40} | grep -E "spmm_rows" | tee -a $OUT/pmc_summary${PMC_NAME:-}.txt
  done
  # the record bench.py quotes as roofline.traffic, stamped with the blob id of the spmm.hip it was measured on
  python tools/pmc_to_json.py $(find $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE_TCC_HIT_sum_TCC_MISS_sum $OUT/pmc_SQ_WAVES* -name "*.db") \
    --tail ${PMC_TAIL:-40} --out $OUT/spmm_dense_traffic${PMC_NAME:-}.json --summary "${PMC_SUMMARY:-profiles/r04_e_pmc_dense_spmm.txt}" \
    --what "${PMC_WHAT:-spmm_rows_kernel<16,false>, dominant launch of the step: dense value-free flavour, yelp2018-shape graph, d=64}" > /dev/null && echo "wrote $OUT/spmm_dense_traffic${PMC_NAME:-}.json"
  find $OUT/pmc_* -name "*.db" -size +30M -delete;;
evalpmc)
  # MFMA utilisation of the scoring GEMM (double-buffered gemm_nt_kernel<64, 8, filter>) + per-kernel times of the ranking
  rm -rf $OUT/evalpmc $OUT/evalprof
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc ${EVALPMC_COUNTERS:-SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES} -d $OLDPWD/$OUT/evalpmc -o pmc -- python $OLDPWD/tools/eval_probe.py > $OLDPWD/$OUT/evalpmc.log 2>&1); echo "evalpmc exit $?"
  f=$(find $OUT/evalpmc -name "*.db" | head -1); [ -n "$f" ] && python tools/pmc_summary.py "$f" | grep -E "gemm_nt|filter16|rescore|topk|cand_|mask_kernel|hit_flags|split_rows" | tee $OUT/eval_pmc.txt
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/evalprof -o trace -- python $OLDPWD/tools/eval_probe.py > $OLDPWD/$OUT/evalprof.log 2>&1); echo "evalprof exit $?"
  stats $OUT/evalprof | grep -E "kernel  |gemm_nt|topk|mask_kernel|cand_|hit_flags" | tee $OUT/eval_kernel_stats.txt; grep -v amdgpu.ids $OUT/evalprof.log | tail -1
  find $OUT/evalpmc $OUT/evalprof -name "*.db" -size +30M -delete;;
big)
  timeout 1500 python tools/big_graph.py > $OUT/big_graph.log 2>&1; echo "big exit $?"; grep -v amdgpu.ids $OUT/big_graph.log | tail -10;;
cols)
  COLS_PROBE_KERNELS_ONLY=${COLS_PROBE_KERNELS_ONLY:-0} timeout 900 python tools/cols_probe.py > $OUT/cols_probe.log 2>&1; echo "cols exit $?"; grep -v amdgpu.ids $OUT/cols_probe.log | tail -30;;
sharded1)
  for lay in ${SHARDED1_LAYOUTS:-rows cols dp}; do
    SRH_FORCE_SHARDED=1 SRH_SHARD_LAYOUT=$lay timeout 600 python bench.py --steps 300 --warmup 20 --no-cpu-baseline > $OUT/bench_sharded1_$lay.log 2> $OUT/bench_sharded1_$lay.err; echo "sharded1 $lay exit $?"
    tail -2 $OUT/bench_sharded1_$lay.err; tail -1 $OUT/bench_sharded1_$lay.log | cut -c1-500
  done;;
modelmatrix)
  timeout 1200 python tools/model_matrix.py > $OUT/model_matrix.log 2>&1; echo "modelmatrix exit $?"; grep -v amdgpu.ids $OUT/model_matrix.log | tail -12;;
rfab)
  # rows_finish (the fixed-order loss finish) part by part: tools/spmm_lab/build_alt.sh rf1 "-DSRH_RF_SKIP=1" ... first (ALT_SRC=losses)
  RF_LIBS="${RF_LIBS:-base}" bash tools/rf_ab.sh 2>&1 | grep -E "^==|rows_finish|nce_finish" | tee $OUT/rf_ab.txt;;
detab)
  # the step with the fixed-order batch-gradient reduction against the atomic scatter, one process, alternating regions
  timeout 900 python tools/det_scatter_ab.py 2>&1 | grep -v amdgpu.ids | tee $OUT/det_scatter_ab.txt | tail -14;;
benchdriver20)
  # the driver's literal command line
  timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver20.log 2> $OUT/bench_driver20.err; echo "benchdriver20 exit $?"
  tail -3 $OUT/bench_driver20.err; tail -1 $OUT/bench_driver20.log | cut -c1-900;;
ablibs)
  bash tools/spmm_lab/ab_libs.sh ${AB_LIBS:-} 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_libs.txt | cut -c1-420;;
abtest)
  # kernel + engine parity tests under an alt library (AB_TEST_LIBS), the product's restored afterwards
  cp selfrec_amd/lib/libselfrec_hip.so /tmp/orig_t.so
  for N in ${AB_TEST_LIBS:-}; do
    cp tools/spmm_lab/alt/libselfrec_hip_$N.so selfrec_amd/lib/libselfrec_hip.so
    timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_engine.py tests/test_gpu_shapes.py -m gpu -q --tb=short -p no:cacheprovider -x -k "${AB_TEST_K:-spmm or yelp or engine or infonce or adam or bpr or step}" > $OUT/abtest_$N.log 2>&1
    echo "abtest $N exit $?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/abtest_$N.log | tail -5
  done
  cp /tmp/orig_t.so selfrec_amd/lib/libselfrec_hip.so;;
evalbreakab)
  # the ranking on TRAINED embeddings under the product's library and under each alt library (EVAL_LIBS)
  cp selfrec_amd/lib/libselfrec_hip.so /tmp/orig_e.so; : > $OUT/eval_breakdown_ab.txt
  for N in product ${EVAL_LIBS:-} product; do
    [ $N = product ] || cp tools/spmm_lab/alt/libselfrec_hip_$N.so selfrec_amd/lib/libselfrec_hip.so
    echo "== $N" >> $OUT/eval_breakdown_ab.txt
    EVAL_TRAIN_STEPS=1300 timeout 300 python tools/eval_breakdown.py 2>&1 | grep -v amdgpu.ids | tail -1 >> $OUT/eval_breakdown_ab.txt
    cp /tmp/orig_e.so selfrec_amd/lib/libselfrec_hip.so
  done
  cat $OUT/eval_breakdown_ab.txt;;
evalbreak)
  timeout 600 python tools/eval_breakdown.py > $OUT/eval_breakdown.txt 2>&1; echo "evalbreak exit $?"; grep -v amdgpu.ids $OUT/eval_breakdown.txt | tail -3;;
timeline)
  timeout 600 python tools/step_timeline.py > $OUT/step_timeline.txt 2>&1; echo "timeline exit $?"; grep -v amdgpu.ids $OUT/step_timeline.txt | cut -c1-700;;
cpuref)
  # the reference's own XSimGCL.train() step beside the oracle's on this host's cores (needs _refstage/)
  (cd /tmp && timeout 900 python $OLDPWD/tools/cpu_reference_vs_port.py --ref $OLDPWD/_refstage --steps ${CPUREF_STEPS:-12} > $OLDPWD/$OUT/cpu_reference_vs_port.txt 2>&1); echo "cpuref exit $?"
  grep -v "amdgpu.ids\|UserWarning\|FloatTensor" $OUT/cpu_reference_vs_port.txt | tail -6;;
startup)
  timeout 600 python tools/startup_probe.py > $OUT/startup_yelp.txt 2>&1; echo "startup exit $?"; grep -v amdgpu.ids $OUT/startup_yelp.txt | tail -14;;
startupbig)
  timeout 1500 python tools/startup_probe.py --shape 1m-500k --emb 128 > $OUT/startup_1m500k.txt 2>&1; echo "startupbig exit $?"; grep -v amdgpu.ids $OUT/startup_1m500k.txt | tail -14;;
*) echo "unknown stage $s";;
esac
echo "-- $s: $(( $(date +%s) - t0 )) s"
done
