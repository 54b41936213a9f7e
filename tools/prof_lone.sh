#!/bin/bash
# rocprofv3 kernel statistics of the bench with ONE call in flight (the kernels' durations when a call has the chip to itself)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_lone
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_lone --output-format csv -- python3 $R/bench.py --steps ${STEPS:-6} --warmup 1 --in-flight 1 --no-cpu-baseline --alt-legs "" --quick ${BENCH_ARGS:-} > $R/gpurun_out/prof_lone.log 2>&1
echo "rc=$?"
f=$(ls $R/gpurun_out/prof_lone/*/*kernel_stats.csv | head -1)
cp "$f" $R/gpurun_out/kernel_stats_lone.csv
find $R/gpurun_out/prof_lone -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$R/gpurun_out/kernel_stats_lone.csv")))
for r in rows[:${TOP:-22}]:
    name=r["Name"]; short=name.split("(")[0][-50:]
    if "heavy_kernel" in name: short="chain_heavy"+name[name.index("heavy_kernel")+12:name.index("heavy_kernel")+18]
    if "msw_kernel" in name: short="msw_kernel"
    if "chain_kernel<" in name: short=name[name.index("chain_kernel<"):name.index("chain_kernel<")+22]
    if "chain_pick" in name: short="chain_pick_kernel"
    print("%-52s calls %5s total %8.1f ms avg %8.3f ms max %8.2f" % (short, r["Calls"], int(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, int(r["MaxNs"])/1e6))
PY
python3 $R/tools/benchline.py < $R/gpurun_out/prof_lone.log
