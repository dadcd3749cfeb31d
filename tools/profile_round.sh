#!/bin/bash
# Round-end evidence: rocprofv3 kernel-trace stats of the default bench command (both precisions), plain bench lines, and the PMC passes of the
# dominant kernel and of the mesh query.   usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>   -> gpurun_out/<tag>_*
set -u
TAG=${1:-r02}; ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
for prec in bf16x3 fp32; do
  echo "== kernel trace, $prec"
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/${TAG}_trace_$prec" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --precision $prec --cpu-rays-side 0 > "$ROOT/gpurun_out/${TAG}_bench_${prec}_profiled.json" 2> "$ROOT/gpurun_out/${TAG}_trace_$prec.err") || echo "trace $prec failed"
  find "gpurun_out/${TAG}_trace_$prec" -name "*kernel_stats.csv" -exec cp {} "gpurun_out/${TAG}_kernel_stats_$prec.csv" \;
  head -8 "gpurun_out/${TAG}_kernel_stats_$prec.csv"
  python3 bench.py --precision $prec > "gpurun_out/${TAG}_bench_$prec.json" 2> "gpurun_out/${TAG}_bench_$prec.err"; tail -c 400 "gpurun_out/${TAG}_bench_$prec.json"; echo
done
bash tools/pmc_mode1.sh 1 gpurun_out/${TAG}_pmc_q1 > gpurun_out/${TAG}_pmc_q1.log 2>&1
PMC_TOOL=tools/perf_mesh.py bash tools/pmc_mode1.sh - gpurun_out/${TAG}_pmc_mesh > gpurun_out/${TAG}_pmc_mesh.log 2>&1
cp gpurun_out/${TAG}_pmc_q1/summary.txt gpurun_out/${TAG}_pmc_q1_summary.txt; cp gpurun_out/${TAG}_pmc_mesh/summary.txt gpurun_out/${TAG}_pmc_mesh_summary.txt
