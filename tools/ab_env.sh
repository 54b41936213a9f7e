#!/bin/bash
# A/B of environment switches on the default bench (quick mode: no CPU baseline, no alt legs); usage: tools/ab_env.sh "VAR=x" "VAR2=y" ...
cd "$(dirname "$0")/.."
for e in "X=1" "$@"; do
  echo "== $e"
  env $e python bench.py --quick --alt-legs "" --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python tools/benchline.py | cut -c1-200
done
