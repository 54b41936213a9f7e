# A/B series on one box: environment settings (one per argument, "-" = none, "A=1,B=2" = several) through a 20-step bench each
for e in "$@"; do
  echo "== $e"
  ( [ "$e" != "-" ] && export ${e//,/ }; timeout 500 python bench.py --steps 20 --no-cpu-baseline 2>/dev/null ) | python tools/benchline.py
done
