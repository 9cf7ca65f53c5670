#!/bin/bash
# Two rocprofv3 PMC passes over a glass frame (level-by-level order), per-kernel means printed:
#   bash tools/glass_counters.sh <out dir> [config]
set -u
OUT=${1:-gpurun_out/glass_counters}; CFG=${2:-C4g}
mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/a" -- python3 tools/quick_bench.py $CFG > "$OUT/a.log" 2>&1 || { tail -5 "$OUT/a.log"; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/b" -- python3 tools/quick_bench.py $CFG > "$OUT/b.log" 2>&1 || { tail -5 "$OUT/b.log"; exit 1; }
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/[ab]/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].split("::")[-1][:24]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[n]["VGPR"] = [float(r["VGPR_Count"])]
        agg[n]["scratch"] = [float(r["Scratch_Size"])]
for n, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    if m.get("SQ_WAVES", 0) < 100:
        continue
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0.0 or m.get("SQ_ACTIVE_INST_VALU", 0.0) <= 0.0:
        continue
    print(f"{n:26s} launches {len(c['SQ_WAVES']):3d} waves {m['SQ_WAVES']:8.0f} VGPR {m['VGPR']:.0f} scratch {m['scratch']:.0f} cycles {cyc:9.0f}  VALU/wave {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:8.0f} SALU/wave {m['SQ_INSTS_SALU'] / m['SQ_WAVES']:8.0f} "
          f"vmem_rd/wave {m['SQ_INSTS_VMEM_RD'] / m['SQ_WAVES']:7.0f} flat/wave {m['SQ_INSTS_FLAT'] / m['SQ_WAVES']:7.0f} smem/wave {m['SQ_INSTS_SMEM'] / m['SQ_WAVES']:7.0f} "
          f"valu_busy {m['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * cyc):.2f} lane_util {m['SQ_THREAD_CYCLES_VALU'] / (m['SQ_ACTIVE_INST_VALU'] * 64):.2f} "
          f"waves/simd {m['SQ_WAVE_CYCLES'] * 4 / (1024 * cyc):.2f} wait_any {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.2f} issue_wait {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.2f}")
PY
