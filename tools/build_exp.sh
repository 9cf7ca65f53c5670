#!/bin/bash
# Builds experiment variants of the device library: tools/exp/librtk_exp<N>.so with -DRTK_EXP=<N>.
# usage: bash tools/build_exp.sh 1 2 ...   then   RTK_LIB=tools/exp/librtk_exp1.so python tools/quick_bench.py C4
set -e
cd "$(dirname "$0")/../raytracer-rust_amd/csrc"
mkdir -p ../../tools/exp
for n in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DRTK_EXP=$n -c rtk_opaque.hip -o /tmp/rtk_opaque_exp$n.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/exp/librtk_exp$n.so rtk_api.o rtk_kernels.o /tmp/rtk_opaque_exp$n.o
done
