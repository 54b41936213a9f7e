#!/bin/bash
# quick loop for work on the seeding kernels: parity first, then the stage alone on the full-size index
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -x -q -k "smem or golden or pe_default or se_variable or edge" 2>&1 | grep -v "^\[M::" | tail -5
timeout 600 python tools/bench_smem.py 3100 ${1:-333334} 3 2>&1 | grep -E "^smem|^sa|rror" | head -8
