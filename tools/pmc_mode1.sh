#!/bin/bash
# Collect PMC counters for one launch of a kernel under its timing tool; separate passes, no trace domains beyond kernel-trace.
# usage: tools/pmc_mode1.sh <mode> <outdir>            query_kernel<mode> under tools/perf_query.py (default)
#        PMC_TOOL=tools/perf_mesh.py tools/pmc_mode1.sh - <outdir>   mesh_query_accel_kernel under tools/perf_mesh.py
set -u
MODE=${1:-1}; OUT=${2:-gpurun_out/pmc_mode$MODE}
ROOT=$(pwd); mkdir -p "$OUT"; export TMPDIR=/tmp
TOOL=${PMC_TOOL:-tools/perf_query.py}
if [ "$TOOL" = "tools/perf_query.py" ]; then ARGS="--iters 2 --mode $MODE"; else ARGS="--hint-only"; fi
pass() { n=$1; shift; echo "pass $n: $*"; (cd /tmp && timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOT/$OUT/p$n" -- python3 "$ROOT/$TOOL" $ARGS > "$ROOT/$OUT/p$n.log" 2>&1) || echo "pass $n failed"; }
pass 1 SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE
pass 2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SALU
pass 3 TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass 4 SQ_IFETCH SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_INSTS_LDS
pass 5 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
pass 6 FETCH_SIZE
pass 7 WRITE_SIZE
find "$OUT" -name "*counter_collection.csv" | xargs python3 "$ROOT/tools/pmc_summary.py" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
