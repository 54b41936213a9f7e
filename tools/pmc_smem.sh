#!/bin/bash
# HBM traffic of the seeding kernels (smem_p3_kernel + smem_kernel) from FETCH_SIZE, calibrated on a random 64-B gather of known
# size in the same access shape (MI355X_MICROARCH.md: FETCH_SIZE is only calibrated for wide streaming reads).  Counter passes
# only, no tracing.  Writes gpurun_out/pmc_smem/r04_pmc_smem.json, to be committed as profiles/r04_pmc_smem.json: bench.py
# scales its traffic_bytes_per_read to the reads of a launch as long as the kernel sources are the ones recorded here.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_smem
rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "gather_probe" -d $O/probe --output-format csv -- python3 $R/tools/gather_probe_one.py > $O/probe.log 2>&1
echo "probe rc=$?"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "smem_" -d $O/smem --output-format csv -- python3 $R/bench.py --steps 1 --warmup 1 --in-flight 1 --no-cpu-baseline --alt-legs "" > $O/smem.log 2>&1
echo "smem rc=$?"
python3 - $O $R <<'PY'
import csv, glob, json, sys
O, R = sys.argv[1], sys.argv[2]
sys.path.insert(0, R)
from bench import kernel_sources_sha256
def rows(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    return [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE"]
probe = [json.loads(l) for l in open(O + "/probe.log") if l.startswith("{")][-1]
pr = rows(O + "/probe")
p_fetch = max(float(r["Counter_Value"]) for r in pr)        # warm-up dispatch + the timed one: the timed one is the larger
bench = [json.loads(l) for l in open(O + "/smem.log") if l.startswith("{")][-1]
sr = rows(O + "/smem")
# one call in flight = two sub-batches per call = two dispatches of smem_p3_kernel per mem_process_seqs call (warm-up, the timed step,
# the one-call pass and the counting pass alike)
calls = sum(1 for r in sr if "smem_p3_kernel" in r["Kernel_Name"]) // 2
reads = calls * 2 * bench["config"]["pairs_per_step_per_gpu"]
total_kib = sum(float(r["Counter_Value"]) for r in sr)
per_kernel = {}
for r in sr:
    k = r["Kernel_Name"].split("(")[0][-40:]
    per_kernel[k] = per_kernel.get(k, 0.0) + float(r["Counter_Value"])
known = float(probe.get("bytes", probe.get("known_bytes", 0)))
scale = known / (p_fetch * 1024.0) if p_fetch and known else 1.0
out = {"what": "HBM-side read traffic of the seeding kernels from rocprofv3 --pmc FETCH_SIZE (own pass, no tracing), calibrated on a random 64-B gather "
               "of known size in the same access shape (tools/pmc_smem.sh, tools/gather_probe_one.py)",
       "calibration": {"probe": probe, "FETCH_SIZE_KiB": p_fetch, "bytes_per_reported_KiB_byte": scale},
       "smem_kernel": {"dispatches": len(sr), "mem_process_seqs_calls": calls, "reads": reads, "FETCH_SIZE_KiB_total": total_kib,
                       "FETCH_SIZE_KiB_by_kernel": per_kernel, "traffic_bytes_per_read": total_kib * 1024.0 * scale / reads,
                       "algo_bytes_per_read": bench["roofline"]["algo_bytes_per_launch"] * bench["roofline"]["launches_per_step"] / (2 * bench["config"]["pairs_per_step_per_gpu"])},
       "workload": bench["config"]["workload"], "kernel_sources_sha256": kernel_sources_sha256(R)}
out["smem_kernel"]["traffic_over_algorithmic"] = out["smem_kernel"]["traffic_bytes_per_read"] / out["smem_kernel"]["algo_bytes_per_read"]
json.dump(out, open(O + "/r04_pmc_smem.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
