#!/bin/bash
# A/B timing of kernel builds from tools/build_variants.py on the GPU box: [MODE=1] [PQ_ARGS=..] tools/ab_variants.sh name1 name2 ...  -> gpurun_out/ab_<name>.log
set -u
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v (mode ${MODE:-1} ${PQ_ARGS:-})"
  VANERF_HIP_LIB=$PWD/exp/libvanerf_$v.so timeout -k 10 200 python3 tools/perf_query.py --mode ${MODE:-1} --iters 6 ${PQ_ARGS:-} > gpurun_out/ab_$v.log 2>&1 || { echo "variant $v failed"; tail -5 gpurun_out/ab_$v.log; exit 1; }
  tail -2 gpurun_out/ab_$v.log
done
