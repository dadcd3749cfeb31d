set -u
ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
for prec in bf16x3 fp32; do
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r03_trace_$prec" -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --precision $prec --cpu-rays-side 0 > "$ROOT/gpurun_out/r03_bench_${prec}_profiled.json" 2> "$ROOT/gpurun_out/r03_trace_$prec.err") || echo "trace $prec failed"
  find "gpurun_out/r03_trace_$prec" -name "*kernel_stats.csv" -exec cp {} "gpurun_out/r03_kernel_stats_$prec.csv" \;
  head -6 "gpurun_out/r03_kernel_stats_$prec.csv"
  timeout -k 10 600 python3 bench.py --precision $prec > "gpurun_out/r03_bench_$prec.json" 2> "gpurun_out/r03_bench_$prec.err"; tail -c 300 "gpurun_out/r03_bench_$prec.json"; echo
done
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r03_trace_train" -- python3 "$ROOT/tools/perf_train_step.py" > "$ROOT/gpurun_out/r03_train_profiled.txt" 2> "$ROOT/gpurun_out/r03_trace_train.err") || echo "trace train failed"
find "gpurun_out/r03_trace_train" -name "*kernel_stats.csv" -exec cp {} "gpurun_out/r03_train_step_kernel_stats.csv" \;
head -12 gpurun_out/r03_train_step_kernel_stats.csv
timeout -k 10 300 python3 tools/timeline_train_step.py > gpurun_out/r03_train_step_timeline.txt 2>&1
for i in 1 2 3; do timeout -k 10 300 python3 tools/perf_train_step.py 2>&1 | grep "training step"; done > gpurun_out/r03_train_step_times.txt
timeout -k 10 300 python3 tools/perf_train_step.py --torch_graph 2>&1 | grep "training step" >> gpurun_out/r03_train_step_times.txt
cat gpurun_out/r03_train_step_times.txt
timeout -k 10 200 python3 tools/perf_spill_kernels.py 2>&1 | tail -1 > gpurun_out/r03_spill_kernels.txt; timeout -k 10 200 python3 tools/bench_dw_products.py 2>&1 | tail -1 >> gpurun_out/r03_spill_kernels.txt; cat gpurun_out/r03_spill_kernels.txt
rm -rf gpurun_out/r03_trace_bf16x3 gpurun_out/r03_trace_fp32 gpurun_out/r03_trace_train
