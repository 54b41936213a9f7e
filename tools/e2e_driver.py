"""File to file through the C driver: `mpiexec -n 1 mpibwa_gpu mem -K 100000000 -P {1,4,8}` on 2 x N reads of 150 bp against
the 3.1 Gbp synthetic reference, timed by the driver's own chunk-loop clock (what the reference brackets with MPI_Wtime,
src/mainParallel.c:1238-1319).  Checks that the sorted SAM body is the same for every -P and equals the Python loop's
(fastq.align_files).  usage: python tools/e2e_driver.py [pairs=2000000]  -> gpurun_out/e2e_driver.json"""
import hashlib, json, os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpibwa_amd import abi, api, bigindex, fastq
from test_driver import EXE, mpiexec

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
wd = "/tmp/mpibwa_bench"
os.makedirs(wd, exist_ok=True)
idx = bigindex.make_or_get(wd, genome_mbp=3100, seed=38, log=lambda *a: print(*a, flush=True),
                           model=os.environ.get("MPIBWA_BENCH_GENOME_MODEL", "grch38like"), repeat_frac=float(os.environ.get("MPIBWA_BENCH_REPEAT_FRAC", "0.05")))
eng = idx.engine
r1, r2 = os.path.join(wd, "e2e_R1.fastq"), os.path.join(wd, "e2e_R2.fastq")
with open(r1, "wb") as f1, open(r2, "wb") as f2:
    for part in range(0, pairs, 250_000):
        n = min(250_000, pairs - part)
        for name, a, b in idx.simulate_pairs(n, seed=900 + part):
            nm = ("@p%d_%s" % (part, name)).encode()
            f1.write(nm + b"/1\n" + a + b"\n+\n" + b"I" * len(a) + b"\n")
            f2.write(nm + b"/2\n" + b + b"\n+\n" + b"I" * len(b) + b"\n")
cores = int(eng.lib.mi355x_host_cpus())
import ctypes as C
C.c_int.in_dll(eng.lib, "bwa_verbose").value = 1
opt = eng.opt(flag=abi.MEM_F_PE, n_threads=cores)
out = os.path.join(wd, "e2e_py.sam")
t0 = time.time()
with open(out, "wb") as fo:
    _, counts = fastq.align_files(eng, opt, r1, r2, out=fo, K=100_000_000, in_flight=4)
py_s = time.time() - t0


def body_md5(path):
    lines = [ln for ln in open(path, "rb").read().splitlines(keepends=True) if not ln.startswith(b"@")]
    lines.sort()
    return hashlib.md5(b"".join(lines)).hexdigest(), len(lines)


want, n_lines = body_md5(out)
res = {"reads": 2 * pairs, "chunks": len(counts), "python_loop": {"in_flight": 4, "seconds": round(py_s, 3), "Mreads_per_s": round(2 * pairs / py_s / 1e6, 3)},
       "sorted_body_md5": want, "records": n_lines, "driver": []}
print("python loop: %.2f s" % py_s, flush=True)
api.load_library().mi355x_finalize()          # the driver process brings its own copy of the index onto the GPU
env = dict(os.environ); env.pop("LD_LIBRARY_PATH", None); env["MPIBWA_DRV_PROF"] = "1"
# E2E_P items: "8": eight chunks in flight; "8n": the same with --no-prewarm; "8:-f", "8:-f -g", "8:--by-chr": the driver's output options
# (the records are then other records or other files: the md5 check is skipped for those, tests/test_gpu_driver.py checks them)
for item in os.environ.get("E2E_P", "1,4,8").split(","):
    item, _, opts = item.partition(":")
    P, extra = int(item.rstrip("n")), (["--no-prewarm"] if item.endswith("n") else []) + opts.split()
    o = os.path.join(wd, "e2e_drv.sam")
    if "--by-chr" in extra:
        os.makedirs(os.path.join(wd, "e2e_bychr"), exist_ok=True)
        o = os.path.join(wd, "e2e_bychr", "x.sam")
    t0 = time.time()
    r = subprocess.run([mpiexec(), "-n", "1", EXE, "mem", "-t", str(cores), "-K", "100000000", "--in-flight", str(P)] + extra + ["-o", o, idx.prefix, r1, r2],
                       capture_output=True, text=True, env=env, timeout=1500)
    wall = time.time() - t0
    m = re.search(r"chunk loop: (\d+) reads in (\d+) chunks.* ([\d.]+) s = ([\d.]+) Mreads/s", r.stderr)
    done = sorted(float(x) for x in re.findall(r"chunk \d+ done at ([\d.]+)", r.stderr))
    steady = None
    if len(done) > 2 * P:   # the rate once every worker has its buffers and its call context: chunks finishing after the first P
        steady = round((len(done) - P) * (2 * pairs / len(done)) / (done[-1] - done[P - 1]) / 1e6, 3)
    if os.environ.get("E2E_SHOW"):
        print("\n".join(l for l in r.stderr.splitlines() if "chunk " in l or "warmed" in l), flush=True)
    wm = re.search(r"(\d+) call contexts warmed on (\d+) sampled reads each in ([\d.]+) s", r.stderr)
    if r.returncode != 0 or not m:
        print(r.stderr[-3000:]); raise SystemExit("driver failed")
    md5, nl = body_md5(o) if not opts else (want, n_lines)
    out_bytes = sum(os.path.getsize(os.path.join(os.path.dirname(o), f)) for f in os.listdir(os.path.dirname(o))) if "--by-chr" in extra else os.path.getsize(o)
    res["driver"].append({"P": P, "args": extra, "output_bytes": out_bytes, "chunk_loop_s": float(m.group(3)), "Mreads_per_s": float(m.group(4)), "process_wall_s": round(wall, 2), "Mreads_per_s_after_first_P_chunks": steady,
                          "prewarm": {"contexts": int(wm.group(1)), "reads_each": int(wm.group(2)), "seconds_beside_the_fastq_scan": float(wm.group(3))} if wm else None,
                          "same_records_as_python_loop": md5 == want})
    print(res["driver"][-1], flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "e2e_driver.json"), "w"), indent=1)
print(json.dumps(res))
