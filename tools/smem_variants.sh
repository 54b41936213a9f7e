#!/bin/bash
# the seeding kernels alone on SURVEY 8d's genome for a few compile-time footprints (run on the GPU box; rebuilds fm_kernels.hip per variant)
cd "$(dirname "$0")/.."
run() { echo "== $1 | $2"; touch mpibwa_amd/csrc/fm_kernels.hip; MPIBWA_CXXFLAGS="$1" python -m mpibwa_amd.build > /dev/null 2>&1; env $2 SMEM_ONLY=1 MPIBWA_SMEM_COUNT=0 python tools/bench_smem.py 3100 333334 3 2>&1 | grep -E "^smem"; }
run "" "X=1"
run "" "MPIBWA_SMEM_WG_PER_CU=4"
run "-DLCAP_S=28 -DSMEM_WG_S=4" "X=1"
run "-DLCAP_S=36 -DSMEM_WG_S=3" "X=1"
run "" "MPIBWA_KMT=0"
touch mpibwa_amd/csrc/fm_kernels.hip; python -m mpibwa_amd.build > /dev/null 2>&1
MPIBWA_SMEM_COUNT=1 SMEM_ONLY=1 python tools/bench_smem.py 3100 333334 1 2>&1 | grep -E "^smem"
