#!/bin/bash
# Copies the judged parts of a tools/profile.sh run into profiles/<name>/ (tracked):
#   bash tools/save_profile.sh gpurun_out/prof_r01e r01
set -e
SRC=$1; DST=profiles/$2
mkdir -p "$DST"
cp "$SRC"/trace/*/*_kernel_stats.csv "$DST/kernel_stats.csv"
for p in sq1 sq2 mix1 mix2 wr rd; do [ -d "$SRC/pmc_$p" ] && cp "$SRC"/pmc_$p/*/*_counter_collection.csv "$DST/pmc_$p.csv"; done
cp "$SRC/summary.json" "$SRC/summary.txt" "$DST/"
[ -f "$SRC/measured_counters.json" ] && cp "$SRC/measured_counters.json" profiles/measured_counters.json
ls -la "$DST"
