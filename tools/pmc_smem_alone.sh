#!/bin/bash
# PMC counters of the seeding kernel pair running alone on one chunk (tools/bench_smem.py, production variant: no block
# counting), one counter set per run, counters only (never combined with tracing).  Summary -> gpurun_out/pmc_smem_alone.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_smem_alone
rm -rf $O; mkdir -p $O
export MPIBWA_SMEM_COUNT=0
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "smem_" -d $O/s$i --output-format csv -- python3 $R/tools/bench_smem.py ${MBP:-3100} ${PAIRS:-333334} 1 > $O/s$i.log 2>&1
  echo "== set $i ($set) rc=$?"
done
python3 - $O <<'PY'
import csv, glob, json, os, sys, collections
out = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(sys.argv[1], "s[0-9]"))):
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = "smem_p3_kernel" if "smem_p3" in r["Kernel_Name"] else "smem_kernel"
            out[k][r["Counter_Name"]] = out[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, o in out.items():
    if o.get("SQ_WAVE_CYCLES"):
        o["valu_active_frac_of_wave_cycles"] = round(o.get("SQ_ACTIVE_INST_VALU", 0) / o["SQ_WAVE_CYCLES"], 3)
        o["wait_any_frac"] = round(o.get("SQ_WAIT_ANY", 0) / o["SQ_WAVE_CYCLES"], 3)
        o["wait_inst_any_frac"] = round(o.get("SQ_WAIT_INST_ANY", 0) / o["SQ_WAVE_CYCLES"], 3)
logs = [l.strip() for l in open(os.path.join(sys.argv[1], "s1.log")) if l.startswith("smem:")]
json.dump({"note": "one launch of the seeding kernel pair over one chunk (666 668 reads of 150 bp, 3.1 Gbp index), alone on the chip; SQ_* in the units rocprofv3 reports",
           "bench_smem": logs, "kernels": out}, open(os.path.join(os.path.dirname(sys.argv[1]), "pmc_smem_alone.json"), "w"), indent=1)
print(json.dumps({"bench_smem": logs, "kernels": out}, indent=1))
PY
