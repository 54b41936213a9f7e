#!/bin/bash
# usage: tools/sweep_env.sh "A=1 B=2" "A=2 B=2" ...   -> one bench line per environment, value + stage split
mkdir -p gpurun_out
for cfg in "$@"; do
  out=$(env $cfg timeout 600 python bench.py --steps 3 --warmup 1 2>>gpurun_out/sweep_err.log | tail -1)
  echo "$cfg => $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); s=d["stage_ms_per_step"]; print(d["value"], {k: round(v,1) for k,v in s.items()})')"
done
