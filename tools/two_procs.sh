#!/bin/bash
# throughput of two processes sharing one GPU (what 2 ranks per GPU, or two calls in flight, would give)
mkdir -p gpurun_out
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/two_single.json
T=${1:-16}
(MPIBWA_THREADS=$T timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/two_a.json) &
A=$!
(MPIBWA_THREADS=$T timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/two_b.json) &
B=$!
wait $A $B
for f in single a b; do python - "$f" <<'PY'
import sys, json
d = json.load(open("gpurun_out/two_%s.json" % sys.argv[1]))
print(sys.argv[1], d["value"], d["ms_per_step"])
PY
done
