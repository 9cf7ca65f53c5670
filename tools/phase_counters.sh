#!/bin/bash
# Instruction and stall counters of the fused kernel by phase: the tree's build against the phase-cut builds
# (tools/build_variant.sh phases1 -DRTK_PHASES=1: camera phase + hit geometry + store; phases2 -DRTK_PHASES=2: ... + shadow
# phase), one rocprofv3 PMC pass per build.  Differences between the builds are the phases' shares.
#   bash tools/phase_counters.sh <out dir> [config]
set -u
OUT=${1:-gpurun_out/phase_counters}; CFG=${2:-C4@3840x2160}
mkdir -p "$OUT"; export TMPDIR=/tmp
for name in tree phases1 phases2; do
  if [ $name = tree ]; then unset RTK_LIB; else export RTK_LIB=raytracer-rust_amd/csrc_ab/$name/librtk.so; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU \
    --output-format csv -d "$OUT/$name" -- python3 tools/quick_bench.py $CFG > "$OUT/$name.log" 2>&1 || { echo "$name failed"; tail -5 "$OUT/$name.log"; exit 1; }
  timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU \
    --output-format csv -d "$OUT/${name}_b" -- python3 tools/quick_bench.py $CFG > "$OUT/${name}_b.log" 2>&1 || { echo "$name b failed"; tail -5 "$OUT/${name}_b.log"; exit 1; }
done
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
rows = {}
for name in ("tree", "phases1", "phases2"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{name}/*/*_counter_collection.csv") + glob.glob(f"{out}/{name}_b/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "render_uniform_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    rows[name] = {k: sum(v) / len(v) for k, v in agg.items()}
keys = sorted(rows["tree"])
print(f"{'per wavefront':28s} {'camera+hit+store':>18s} {'shadow phase':>14s} {'shading':>10s} {'whole kernel':>14s}")
for k in keys:
    w = {n: rows[n].get(k, 0.0) / max(rows[n].get("SQ_WAVES", 1.0), 1.0) for n in rows}
    if k == "SQ_WAVES":
        continue
    print(f"{k:28s} {w['phases1']:18.1f} {w['phases2'] - w['phases1']:14.1f} {w['tree'] - w['phases2']:10.1f} {w['tree']:14.1f}")
PY
