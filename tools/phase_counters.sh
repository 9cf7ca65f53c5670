#!/bin/bash
# Instruction and stall counters of the fused kernel by phase: the tree's build against phase-cut experiment builds
# (tools/build_variant.sh phasesN -DRTK_PHASES=N, see rtk_opaque.hip), one rocprofv3 PMC pass per build.
#   bash tools/phase_counters.sh <out dir> [config] [builds...]
set -u
OUT=${1:-gpurun_out/phase_counters}; CFG=${2:-C4@3840x2160}; shift 2
BUILDS=${@:-tree phases1 phases2 phases3 phases4 phases5}
mkdir -p "$OUT"; export TMPDIR=/tmp
for name in $BUILDS; do
  if [ $name = tree ]; then unset RTK_LIB; else export RTK_LIB=raytracer-rust_amd/csrc_ab/$name/librtk.so; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM \
    --output-format csv -d "$OUT/$name" -- python3 tools/quick_bench.py $CFG > "$OUT/$name.log" 2>&1 || { echo "$name failed"; tail -5 "$OUT/$name.log"; exit 1; }
  grep "^C" "$OUT/$name.log" | cut -c1-20 | sed "s/^/$name /"
done
python3 - "$OUT" $BUILDS <<'PY'
import collections, csv, glob, sys
out, builds = sys.argv[1], sys.argv[2:]
rows = {}
for name in builds:
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{name}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "render_uniform_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows[name] = {k: sum(v) / len(v) for k, v in agg.items()}
keys = [k for k in sorted(rows[builds[0]]) if k != "SQ_WAVES"]
print("per wavefront".ljust(22) + "".join(b.rjust(12) for b in builds))
for k in keys:
    print(k.ljust(22) + "".join(f"{rows[b].get(k, 0.0) / max(rows[b].get('SQ_WAVES', 1.0), 1.0):12.1f}" for b in builds))
PY
