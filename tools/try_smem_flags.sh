#!/bin/bash
# recompile fm_kernels.hip with extra flags on the GPU box, relink, time the seeding kernel standalone
R=$GRAFT_REPO_ROOT; cd $R
COMMON="-O3 -std=c++17 -fPIC -ffp-contract=off -g1 -I mpibwa_amd/csrc -I include"
OBJS=$(for f in mpibwa_amd/csrc/*.cpp mpibwa_amd/csrc/*.hip; do b=$(basename $f); [ "$b" = fm_kernels.hip ] || echo mpibwa_amd/build/$b.o; done)
for flags in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -x hip $COMMON $flags -c mpibwa_amd/csrc/fm_kernels.hip -o /tmp/fm_try.o 2>/tmp/fm_try.err || { echo "[$flags] compile failed"; tail -3 /tmp/fm_try.err; continue; }
  /opt/rocm/bin/hipcc -shared -o mpibwa_amd/libmpibwa_amd.so $OBJS /tmp/fm_try.o -Wl,-Bsymbolic -lpthread -lm
  echo "== [$flags]"
  timeout 300 python tools/bench_smem.py 3100 333334 3 2>/dev/null | grep "^smem" | tail -2
done
