#!/bin/bash
# A/B of two builds of the library on the kernel-time micro bench: the tree's csrc/ against a baseline build in csrc_ab/
set -e
CONFIGS="${@:-C2 C3@3840x2160 C4@3840x2160 C5s}"
for rep in 1 2; do
  echo "baseline:"; RTK_LIB=raytracer-rust_amd/csrc_ab/librtk.so timeout -k 10 120 python -u tools/quick_bench.py $CONFIGS | cut -c1-40
  echo "tree:";     timeout -k 10 120 python -u tools/quick_bench.py $CONFIGS | cut -c1-40
done
