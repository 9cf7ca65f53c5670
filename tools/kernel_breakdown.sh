#!/bin/bash
# Per-kernel average times (rocprofv3 kernel trace) for the given workloads.
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
for w in "$@"; do
  rm -rf gpurun_out/tr_$w
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tr_$w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --workload $w > gpurun_out/tr_$w.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/tr_$w/*/*kernel_stats.csv")[0]
tot=0
for r in csv.DictReader(open(f)):
    if "rtk::" in r["Name"]:
        ms=float(r["AverageNs"])/1e6; tot+=ms
        print("$w %-28s %8.3f ms" % (r["Name"].split("(")[0].replace("void ","")[-28:], ms))
print("$w total %.3f ms" % tot)
PY
done
