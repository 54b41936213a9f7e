#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench run -> gpurun_out/prof_bench/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_bench
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench --output-format csv -- python3 $R/bench.py --steps ${STEPS:-16} --warmup 2 --no-cpu-baseline --alt-legs "" ${BENCH_ARGS:-} > $R/gpurun_out/prof_bench.log 2>&1
echo "rc=$?"
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-400
f=$(ls $R/gpurun_out/prof_bench/*/*kernel_stats.csv | head -1)
cp "$f" $R/gpurun_out/kernel_stats.csv
head -20 "$f"
t=$(ls $R/gpurun_out/prof_bench/*/*kernel_trace.csv | head -1)
python3 $R/tools/gpu_busy.py "$t"
# the trace itself is large: keep only the summary
find $R/gpurun_out/prof_bench -name "*kernel_trace.csv" -delete
