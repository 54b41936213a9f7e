#!/bin/bash
# PMC counters of the four big kernels, one kernel and one counter set per run (never combined with tracing);
# summary -> gpurun_out/pmc_kernels.json (per kernel: totals over the dispatches of two mem_process_seqs calls)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_kernels
rm -rf $OUT; mkdir -p $OUT
for k in ${PMC_KERNELS:-smem_kernel c2a_kernel msw2_kernel chain_heavy_kernel}; do
  i=0
  for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    timeout 400 rocprofv3 --pmc $set --kernel-include-regex "${k}[^_]" -d $OUT/${k}_$i --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --in-flight 1 --no-cpu-baseline --alt-legs "" --quick > $OUT/${k}_$i.log 2>&1
    echo "== $k set $i rc=$?"
  done
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys, collections
out = {}
for d in sorted(glob.glob(os.path.join(sys.argv[1], "*_[12]"))):
    k = os.path.basename(d).rsplit("_", 1)[0]
    acc = collections.defaultdict(float); n = 0
    for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")):
        disp = set()
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        n = max(n, len(disp))
    o = out.setdefault(k, {})
    o.update({c: v for c, v in acc.items()})
    o["dispatches"] = n
for k, o in out.items():
    if o.get("SQ_WAVE_CYCLES"):
        o["valu_active_frac_of_wave_cycles"] = round(o.get("SQ_ACTIVE_INST_VALU", 0) / o["SQ_WAVE_CYCLES"], 3)
    if o.get("SQ_WAVES"):
        o["valu_insts_per_wave"] = round(o.get("SQ_INSTS_VALU", 0) / o["SQ_WAVES"], 1)
json.dump({"note": "four mem_process_seqs calls of 333 334 pairs on SURVEY 8d's config 1 (bench.py --steps 1 --warmup 1 --in-flight 1 --quick: three warm-up calls + the timed one); "
                   "SQ_* summed over all dispatches of the kernel, cycle counters in the units rocprofv3 reports",
           "kernels": out}, open(os.path.join(os.path.dirname(sys.argv[1]), "pmc_kernels.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
PY
