#!/bin/bash
# HBM traffic of smem_kernel from FETCH_SIZE, calibrated on a random 64-B gather of known size in the same access shape
# (MI355X_MICROARCH.md: FETCH_SIZE is only calibrated for wide streaming reads).  Counter passes only, no tracing.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_smem
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gather_probe" -d $O/probe --output-format csv -- python3 $R/tools/gather_probe_one.py > $O/probe.log 2>&1
echo "probe rc=$?"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "smem_kernel" -d $O/smem --output-format csv -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/smem.log 2>&1
echo "smem rc=$?"
python3 - $O <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
def rows(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    return [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
probe = [json.loads(l) for l in open(O + "/probe.log") if l.startswith("{")][-1]
pr = rows(O + "/probe")
# the probe entry point launches a short warm-up dispatch and the timed one: the timed one is the larger
p_fetch = max(float(r["Counter_Value"]) for r in pr)
bench = [json.loads(l) for l in open(O + "/smem.log") if l.startswith("{")][-1]
sr = rows(O + "/smem")
s_fetch = sum(float(r["Counter_Value"]) for r in sr) / len(sr)
out = {"probe": probe, "probe_dispatches": len(pr), "probe_FETCH_SIZE_per_dispatch": p_fetch,
       "smem_dispatches": len(sr), "smem_FETCH_SIZE_per_dispatch": s_fetch,
       "bench_roofline": bench["roofline"], "pairs_per_step": bench["config"]["pairs_per_step_per_gpu"]}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
