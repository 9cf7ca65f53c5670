#!/bin/bash
# One rocprofv3 PMC pass over a glass frame for the memory pipeline of each kernel: how busy the texture
# addressers are, cache-line accesses of the L1s, requests to L2:
#   bash tools/glass_mem_counters.sh <out dir> [config]
set -u
OUT=${1:-gpurun_out/glass_mem}; CFG=${2:-C4g}
mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc TA_BUSY_avr TA_FLAT_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE TD_TD_BUSY_sum SQ_WAVES --output-format csv -d "$OUT/m" -- python3 tools/quick_bench.py $CFG > "$OUT/m.log" 2>&1 || { tail -5 "$OUT/m.log"; exit 1; }
python3 - "$OUT" <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/m/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].split("::")[-1][:24]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in agg.items():
    m = {k: sum(v) / len(v) for k, v in c.items()}
    if m.get("SQ_WAVES", 0) < 100:
        continue
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if cyc <= 0:
        continue
    w = m["SQ_WAVES"]
    print(f"{n:26s} launches {len(c['SQ_WAVES']):3d} waves {w:8.0f} cycles {cyc:9.0f}  TA busy {m['TA_BUSY_avr'] / cyc:.2f}  TD busy {m['TD_TD_BUSY_sum'] / 256 / cyc:.2f}  "
          f"vmem instr/wave {m['TA_FLAT_WAVEFRONTS_sum'] / w:6.0f}  L1 line accesses/wave {m['TCP_TOTAL_CACHE_ACCESSES_sum'] / w:7.0f}  "
          f"L2 read req/wave {m['TCP_TCC_READ_REQ_sum'] / w:6.0f} write req/wave {m['TCP_TCC_WRITE_REQ_sum'] / w:6.0f}  L1 pending-stall {m['TCP_PENDING_STALL_CYCLES_sum'] / 256 / cyc:.2f}")
PY
