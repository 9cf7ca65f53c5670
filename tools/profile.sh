#!/bin/bash
# rocprofv3 passes for the bench workload.  Run on the GPU box from the repo root:
#   bash tools/profile.sh <tag>        (results under gpurun_out/prof_<tag>/)
# Each pass is its own process: kernel trace + stats first, then PMC passes that stay
# within the per-block counter limits (8 SQ, 4 TCC, 2 GRBM).
set -u
TAG=${1:-r01}
OUT=gpurun_out/prof_${TAG}
mkdir -p "$OUT"
cd /tmp 2>/dev/null && export TMPDIR=/tmp; cd - >/dev/null
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra ${BENCH_ARGS:-}"
# the kernel-trace pass runs the default step counts so that its averages are comparable with bench.py's own
TRACE="python3 bench.py --no-cpu-baseline --no-extra ${BENCH_ARGS:-}"

timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $TRACE > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/pmc_sq1" -- $BENCH > "$OUT/pmc_sq1.log" 2>&1 || { echo "pmc sq1 failed"; tail -5 "$OUT/pmc_sq1.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d "$OUT/pmc_sq2" -- $BENCH > "$OUT/pmc_sq2.log" 2>&1 || { echo "pmc sq2 failed"; tail -5 "$OUT/pmc_sq2.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d "$OUT/pmc_mix1" -- $BENCH > "$OUT/pmc_mix1.log" 2>&1 || { echo "pmc mix1 failed"; tail -5 "$OUT/pmc_mix1.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU --output-format csv -d "$OUT/pmc_mix2" -- $BENCH > "$OUT/pmc_mix2.log" 2>&1 || { echo "pmc mix2 failed"; tail -5 "$OUT/pmc_mix2.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_wr" -- $BENCH > "$OUT/pmc_wr.log" 2>&1 || { echo "pmc wr failed"; tail -5 "$OUT/pmc_wr.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_rd" -- $BENCH > "$OUT/pmc_rd.log" 2>&1 || { echo "pmc rd failed"; tail -5 "$OUT/pmc_rd.log"; exit 1; }
find "$OUT" -name "*.csv" | head -40
echo done
