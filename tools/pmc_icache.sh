#!/bin/bash
# Instruction-cache counters of the two big kernels (separate rocprofv3 --pmc passes, kernel trace only).  usage: bash tools/pmc_icache.sh <outdir>
set -u
OUT=${1:-gpurun_out/pmc_icache}; ROOT=$(pwd); mkdir -p "$OUT"; export TMPDIR=/tmp
run() { name=$1; shift; tool=$1; shift; (cd /tmp && timeout -k 10 150 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d "$ROOT/$OUT/$name" -- python3 "$ROOT/$tool" "$@" > "$ROOT/$OUT/$name.log" 2>&1) || echo "$name failed"; }
run query tools/perf_query.py --iters 2 --mode 1
run mesh tools/perf_mesh.py --hint-only
find "$OUT" -name "*counter_collection.csv" | xargs python3 "$ROOT/tools/pmc_summary.py" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
