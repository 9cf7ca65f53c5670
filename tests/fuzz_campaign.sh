#!/bin/bash
# One fuzz campaign against the CPU oracle on a GPU box: small scenes, glass scenes in both trace orders,
# big scenes.  Usage: tests/fuzz_campaign.sh <tag> [first seed] [small] [glass] [big]; totals land in gpurun_out/<tag>/.
set -e
TAG=${1:-fuzz}; SEED=${2:-300001}; SMALL=${3:-1500}; GLASS=${4:-120}; BIG=${5:-50}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout -k 10 500 python -u tests/fuzz_parity.py $SMALL $SEED > $OUT/fuzz_small.txt 2>&1; tail -1 $OUT/fuzz_small.txt
FUZZ_GLASS=1 RTK_TRACE_ORDER=2 timeout -k 10 300 python -u tests/fuzz_parity.py $GLASS $((SEED + 100000)) > $OUT/fuzz_glass_by_level.txt 2>&1; tail -1 $OUT/fuzz_glass_by_level.txt
FUZZ_GLASS=1 RTK_TRACE_ORDER=1 timeout -k 10 300 python -u tests/fuzz_parity.py $GLASS $((SEED + 100000)) > $OUT/fuzz_glass_depth_first.txt 2>&1; tail -1 $OUT/fuzz_glass_depth_first.txt
FUZZ_BIG=1 timeout -k 10 400 python -u tests/fuzz_parity.py $BIG $((SEED + 200000)) > $OUT/fuzz_big.txt 2>&1; tail -1 $OUT/fuzz_big.txt
