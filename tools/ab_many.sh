#!/bin/bash
# Kernel-time micro bench of several builds of the library, interleaved twice: the tree's csrc/ and every
# raytracer-rust_amd/csrc_ab/<name>/librtk.so given as arguments after the configs (configs first, then "--", then names).
set -e
CONFIGS=(); NAMES=()
seen=0
for a in "$@"; do if [ "$a" = "--" ]; then seen=1; elif [ $seen = 0 ]; then CONFIGS+=("$a"); else NAMES+=("$a"); fi; done
for rep in 1 2 3; do
  echo "tree:"; timeout -k 10 120 python -u tools/quick_bench.py "${CONFIGS[@]}" 2>/dev/null | cut -c1-40
  for n in "${NAMES[@]}"; do
    echo "$n:"; RTK_LIB=raytracer-rust_amd/csrc_ab/$n/librtk.so timeout -k 10 120 python -u tools/quick_bench.py "${CONFIGS[@]}" 2>/dev/null | cut -c1-40
  done
done
