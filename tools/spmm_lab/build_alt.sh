#!/bin/bash
# The product library rebuilt with other compile-time switches of csrc/spmm.hip, for a file-level A/B of bench.py on
# the GPU box (no run-time knob ships in the library):
#   tools/spmm_lab/build_alt.sh <name> "<flags>" ...   ->  tools/spmm_lab/alt/libselfrec_hip_<name>.so
#   e.g.  build_alt.sh exp "-DSRH_EXP_SOMETHING=1"   (a switch the experiment adds to csrc/spmm.hip for its duration)
# (cp the file over selfrec_amd/lib/libselfrec_hip.so to use it; tools/spmm_lab/ab_libs.sh does that in a loop)
set -e
cd "$(dirname "$0")/../.."
make -C selfrec_amd/csrc > /dev/null
mkdir -p tools/spmm_lab/alt
B=selfrec_amd/csrc/build
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc $FLAGS \
    -c selfrec_amd/csrc/spmm.hip -o tools/spmm_lab/alt/spmm_$NAME.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $B/common.o $B/sampler.o $B/loader.o tools/spmm_lab/alt/spmm_$NAME.o \
    $B/graph.o $B/losses.o $B/optim.o $B/eval.o $B/exchange.o -o tools/spmm_lab/alt/libselfrec_hip_$NAME.so
  echo "built tools/spmm_lab/alt/libselfrec_hip_$NAME.so  ($FLAGS)"
done
