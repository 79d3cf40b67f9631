#!/bin/bash
# Experiment builds of libcp_pre_hip.so that differ in one source file's -D knobs:
#   tools/exp/build_variants.sh name "-DMARCH6_NR=8 -DMARCH6_TYQ=32 ..." [source.hip]  ->  tools/exp/var/libcp_pre_hip.<name>.so
set -e
cd "$(dirname "$0")/../../cp_pre_amd/csrc"
mkdir -p ../../tools/exp/var
name=$1; flags=$2; src=${3:-star_march.hip}
base=${src%.hip}
objs=""
for o in star_march acc_march stencil_generic calib kth_axis0; do
  if [ "$o" != "$base" ]; then objs="$objs $o.o"; fi
done
extra=""
[ "$base" = "kth_axis0" ] && extra="-Wno-pass-failed"
[ "$base" = "calib" ] && extra="-ffp-contract=off"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $extra $flags -c $src -o ../../tools/exp/var/$base.$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/var/libcp_pre_hip.$name.so ../../tools/exp/var/$base.$name.o $objs
rm -f ../../tools/exp/var/$base.$name.o
echo built $name
