#!/bin/bash
# usage: tools/sweep_env_fly.sh "A=1" "A=2" ... -> default bench (calls in flight) per environment
mkdir -p gpurun_out
for cfg in "$@"; do
  out=$(env $cfg MPIBWA_BENCH_CPU_PAIRS=${CPU_PAIRS:-60000} timeout 600 python bench.py --steps ${STEPS:-9} --warmup 1 2>>gpurun_out/sweep_err.log | tail -1)
  echo "$cfg => $(echo "$out" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); s=d["stage_ms_per_step"]; print(d["value"], d["ms_per_step"], d.get("host_cpu_s_per_step"), d.get("host_cpu_busy_frac"), d.get("parity_on_sample"), {k: round(s[k],1) for k in ("phase1_ms","sam_ms","k_smem_ms","k_ext_ms","k_msw_ms","k_aln_ms")})')"
done
