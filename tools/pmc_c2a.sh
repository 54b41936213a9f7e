#!/bin/bash
# PMC counters of the chain->region kernel, two small passes (never combined with tracing)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --pmc $set --kernel-include-regex "${KREGEX:-c2a_lane}" -d $R/gpurun_out/pmc_${KTAG:-c2a}/$tag --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 > $R/gpurun_out/pmc_${KTAG:-c2a}_$tag.log 2>&1
  echo "== $set rc=$?"
  f=$(ls $R/gpurun_out/pmc_${KTAG:-c2a}/$tag/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in acc: print(k, acc[k], "dispatch-rows", n[k])
PY
done
