#!/bin/bash
# Collect PMC counters for one coarse query_kernel launch (tools/perf_query.py); separate passes, no trace domains beyond kernel-trace.
# usage: tools/pmc_mode1.sh <mode> <outdir>
set -u
MODE=${1:-1}; OUT=${2:-gpurun_out/pmc_mode$MODE}
ROOT=$(pwd); mkdir -p "$OUT"; export TMPDIR=/tmp
pass() { n=$1; shift; echo "pass $n: $*"; (cd /tmp && timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/p$n" -- python3 "$ROOT/tools/perf_query.py" --iters 2 --mode "$MODE" > "$ROOT/$OUT/p$n.log" 2>&1) || echo "pass $n failed"; }
pass 1 SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass 2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU
pass 3 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass 4 SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS
pass 6 FETCH_SIZE WRITE_SIZE
find "$OUT" -name "*counter_collection.csv" | xargs python3 "$ROOT/tools/pmc_summary.py" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
