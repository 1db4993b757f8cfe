#!/bin/bash
# The product library rebuilt with other compile-time switches of csrc/spmm.hip, for a file-level A/B of bench.py on
# the GPU box (no run-time knob ships in the library):
#   tools/spmm_lab/build_alt.sh <name> "<flags>" ...   ->  tools/spmm_lab/alt/libselfrec_hip_<name>.so
#   e.g.  build_alt.sh exp "-DSRH_EXP_SOMETHING=1"   (a switch the experiment adds to csrc/spmm.hip for its duration)
#   ALT_SRC=losses build_alt.sh s16 "-DSRH_NCE_SPLITS=16"   rebuilds csrc/losses.hip instead (the InfoNCE shape constants)
# (cp the file over selfrec_amd/lib/libselfrec_hip.so to use it; tools/spmm_lab/ab_libs.sh does that in a loop)
set -e
cd "$(dirname "$0")/../.."
make -C selfrec_amd/csrc > /dev/null
mkdir -p tools/spmm_lab/alt
B=selfrec_amd/csrc/build
SRC=${ALT_SRC:-spmm}
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc $FLAGS \
    -c selfrec_amd/csrc/$SRC.hip -o tools/spmm_lab/alt/${SRC}_$NAME.o
  OBJS=""
  for o in common sampler loader spmm graph losses optim eval exchange; do
    if [ $o = $SRC ]; then OBJS="$OBJS tools/spmm_lab/alt/${SRC}_$NAME.o"; else OBJS="$OBJS $B/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o tools/spmm_lab/alt/libselfrec_hip_$NAME.so
  echo "built tools/spmm_lab/alt/libselfrec_hip_$NAME.so  ($FLAGS)"
done
