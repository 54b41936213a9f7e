#!/bin/bash
# usage: tools/sweep_fly.sh "1" "2" "3" ... -> bench value per number of calls in flight
mkdir -p gpurun_out
for f in "$@"; do
  out=$(MPIBWA_BENCH_CPU_PAIRS=${CPU_PAIRS:-60000} timeout 600 python bench.py --steps ${STEPS:-9} --warmup 1 --in-flight $f 2>>gpurun_out/sweep_err.log | tail -1)
  echo "in_flight=$f => $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["launch_ms"], d.get("parity_on_sample"))')"
done
