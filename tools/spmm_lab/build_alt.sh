#!/bin/bash
# The product library rebuilt with other compile-time switches, for a file-level A/B of bench.py on the GPU box (no
# run-time knob ships in the library):
#   tools/spmm_lab/build_alt.sh <name> "<flags>" ...   ->  tools/spmm_lab/alt/libselfrec_hip_<name>.so
#   e.g.  build_alt.sh ywt "-DSRH_Y_WT=1"              (a switch of csrc/spmm.hip)
#   ALT_SRC="spmm optim losses" build_alt.sh wt "-DSRH_Y_WT=1 -DSRH_ADAM_WT=1 -DSRH_NCE_WT=1"   rebuilds those sources
#   ALT_SRC=losses build_alt.sh s16 "-DSRH_NCE_SPLITS=16"
# (cp the file over selfrec_amd/lib/libselfrec_hip.so to use it; tools/spmm_lab/ab_libs.sh does that in a loop)
set -e
cd "$(dirname "$0")/../.."
make -C selfrec_amd/csrc > /dev/null
mkdir -p tools/spmm_lab/alt
B=selfrec_amd/csrc/build
SRCS=${ALT_SRC:-spmm}
while [ $# -ge 2 ]; do
  NAME=$1; FLAGS=$2; shift 2
  for S in $SRCS; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-gpu-rdc $FLAGS \
      -c selfrec_amd/csrc/$S.hip -o tools/spmm_lab/alt/${S}_$NAME.o &
  done
  wait
  OBJS=""
  for o in common sampler loader collectives spmm graph losses optim eval exchange; do
    if [[ " $SRCS " == *" $o "* ]]; then OBJS="$OBJS tools/spmm_lab/alt/${o}_$NAME.o"; else OBJS="$OBJS $B/$o.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -ldl -o tools/spmm_lab/alt/libselfrec_hip_$NAME.so
  echo "built tools/spmm_lab/alt/libselfrec_hip_$NAME.so  ($SRCS: $FLAGS)"
done
