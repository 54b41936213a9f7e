#!/bin/bash
# seeding kernel pair alone on one chunk at 1..5 workgroups per CU (is it issue-bound or latency-bound?)
R=$GRAFT_REPO_ROOT
export MPIBWA_SMEM_COUNT=0
for w in 5 4 3 2 1; do
  echo "== MPIBWA_SMEM_WG_PER_CU=$w"
  MPIBWA_SMEM_WG_PER_CU=$w python3 $R/tools/bench_smem.py ${MBP:-3100} ${PAIRS:-333334} 2 2>&1 | grep "smem:" | tail -1
done
