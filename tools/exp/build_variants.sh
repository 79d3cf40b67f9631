#!/bin/bash
# Experiment builds of libcp_pre_hip.so that differ in star_march.hip's knobs for the six-field functors:
#   tools/exp/build_variants.sh name "-DMARCH6_NR=8 -DMARCH6_TYQ=32 ..."  ->  tools/exp/var/libcp_pre_hip.<name>.so
set -e
cd "$(dirname "$0")/../../cp_pre_amd/csrc"
mkdir -p ../../tools/exp/var
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $@ -c star_march.hip -o ../../tools/exp/var/star_march.$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/exp/var/libcp_pre_hip.$name.so ../../tools/exp/var/star_march.$name.o acc_march.o stencil_generic.o calib.o kth_axis0.o
rm -f ../../tools/exp/var/star_march.$name.o
echo built $name
