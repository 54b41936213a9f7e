#!/bin/bash
# The library's host sources (everything that is not a kernel or a launch: boundary, options, FASTQ, chaining, ksw, pairing, regions /
# CIGAR / SAM text, index builder, the passes behind the call) built alone with AddressSanitizer + UndefinedBehaviorSanitizer, and the
# host-logic tests of the CPU suite run on that build (GPU AddressSanitizer is not available on the pool: sanitizers run on the CPU build).
#   tools/san_host.sh [pytest arguments]        -> /tmp/mpibwa_san/libhost_san.so, the tests' output
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
O=/tmp/mpibwa_san
mkdir -p $O
cat > $O/stub.cpp <<'EOF'
// what the host sources take from the .hip side of the library
extern "C" int mi355x_host_cpus(void) { return 4; }
EOF
cd $R/mpibwa_amd/csrc
g++ -std=c++17 -O1 -g -fPIC -shared -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -ffp-contract=off -w \
    -I. -I../../include boundary.cpp common.cpp fastq.cpp host_chain.cpp host_ksw.cpp host_pair.cpp host_regs.cpp index.cpp sampost.cpp $O/stub.cpp \
    -o $O/libhost_san.so -lz -lpthread -lm -ldl
cd $R
export MPIBWA_SANITIZER_LIB=$O/libhost_san.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
# (not on this build: tests that go through entry points living in .hip files — mi355x_chain_batch, mem_process_seqs — or that list every export)
exec python -m pytest -q -m "not gpu" -p no:cacheprovider -k "not exports and not exported and not align_files" ${@:-tests/test_sampost.py tests/test_host_pair.py tests/test_host_ksw.py tests/test_fastq.py tests/test_index.py tests/test_boundary.py}
