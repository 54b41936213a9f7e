#!/bin/bash
# Per-kernel time and PMC counters of the SMEM seeding stage alone (tools/bench_smem.py: one chunk of 666 668 reads, 3 repetitions).
# Separate rocprofv3 runs: kernel trace + stats; SQ counters; FETCH_SIZE; TCC hit/miss (counter runs carry no tracing).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof_smem}
PAIRS=${2:-333334}
rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $R/tools/bench_smem.py 3100 $PAIRS 3 > $O/trace.log 2>&1
echo "trace rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-include-regex "smem_" -d $O/sq1 --output-format csv -- python3 $R/tools/bench_smem.py 3100 $PAIRS 1 > $O/sq1.log 2>&1
echo "sq1 rc=$?"
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD --kernel-include-regex "smem_" -d $O/sq2 --output-format csv -- python3 $R/tools/bench_smem.py 3100 $PAIRS 1 > $O/sq2.log 2>&1
echo "sq2 rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "smem_" -d $O/fetch --output-format csv -- python3 $R/tools/bench_smem.py 3100 $PAIRS 1 > $O/fetch.log 2>&1
echo "fetch rc=$?"
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-include-regex "smem_" -d $O/tcc --output-format csv -- python3 $R/tools/bench_smem.py 3100 $PAIRS 1 > $O/tcc.log 2>&1
echo "tcc rc=$?"
python3 - $O <<'PY'
import csv, glob, json, sys, collections
O = sys.argv[1]
out = {}
st = glob.glob(O + "/trace/*/*kernel_stats.csv")
if st:
    out["kernel_stats"] = [r for r in csv.DictReader(open(st[0])) if "smem" in r["Name"] or "seed" in r["Name"]]
for d in ("sq1", "sq2", "fetch", "tcc"):
    fs = glob.glob(O + "/%s/*/*counter_collection.csv" % d)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(int)
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        out.setdefault("pmc", {}).setdefault(k, {}).update(v)
out["bench_lines"] = [l.strip() for l in open(O + "/trace.log") if l.startswith("smem:")]
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
