#!/bin/bash
# second-level counters of the seeding stage: clock, TA/TCP busy and stalls, VALU thread cycles
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof_smem2}
rm -rf $O; mkdir -p $O
run() { d=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-include-regex "smem_" -d $O/$d --output-format csv -- python3 $R/tools/bench_smem.py 3100 333334 1 > $O/$d.log 2>&1; echo "$d rc=$?"; }
run g GRBM_GUI_ACTIVE GRBM_TA_BUSY SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_INT32 SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH
# (TA_* counters are not collected: a pass with them did not finish within 10 minutes on this pool)
run c TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum
python3 - $O <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
acc = collections.defaultdict(float)
for d in "gc":
    for f in glob.glob(O + "/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
print(json.dumps(acc, indent=1))
import subprocess
for d in "gc":
    print(d, open(O + "/%s.log" % d).read()[-300:])
PY
