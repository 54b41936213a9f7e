#!/bin/bash
# the seeding kernels alone on the chip (production variant: MPIBWA_SMEM_COUNT=0) for several depths of the k-mer tables
# usage: tools/kmt_sweep.sh [genome-model-args for bigindex are fixed: uniform 3100 Mbp]   (run on the GPU box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for k in ${KMT_LIST:-0 12 13 14 15}; do
  echo "== MPIBWA_KMT=$k"
  MPIBWA_KMT=$k MPIBWA_SMEM_COUNT=0 python tools/bench_smem.py 3100 333334 4 2>&1 | grep -E "^smem"
done
