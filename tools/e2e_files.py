"""File to file: R1.fastq + R2.fastq -> SAM body through mpiBWA's chunking rule (-K 1e8), three chunks in flight.
What the caller's side of mem_process_seqs costs on top of the bench figure: FASTQ scan, bseq1_t fill, SAM concatenation and
write.  usage: python tools/e2e_files.py [pairs=2000000]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpibwa_amd import abi, bigindex, fastq

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
wd = "/tmp/mpibwa_bench"
os.makedirs(wd, exist_ok=True)
idx = bigindex.make_or_get(wd, genome_mbp=3100, seed=38, log=lambda *a: None)
eng = idx.engine
r1, r2 = os.path.join(wd, "e2e_R1.fastq"), os.path.join(wd, "e2e_R2.fastq")
t0 = time.time()
with open(r1, "wb") as f1, open(r2, "wb") as f2:
    for part in range(0, pairs, 250_000):
        n = min(250_000, pairs - part)
        for name, a, b in idx.simulate_pairs(n, seed=900 + part):
            nm = ("@p%d_%s" % (part, name)).encode()
            q = b"I" * len(a)
            f1.write(nm + b"/1\n" + a + b"\n+\n" + q + b"\n")
            f2.write(nm + b"/2\n" + b + b"\n+\n" + b"I" * len(b) + b"\n")
print("wrote 2 x %d reads in %.1f s" % (pairs, time.time() - t0), flush=True)
import ctypes as C
C.c_int.in_dll(eng.lib, "bwa_verbose").value = 1
opt = eng.opt(flag=abi.MEM_F_PE, n_threads=int(eng.lib.mi355x_host_cpus()))
class Null:
    def write(self, b):
        return len(b)
t0 = time.time()
src = fastq.FastqSource(eng.lib, r1, r2, K=100_000_000)
print("FastqSource (map + scan both files + chunk table): %.2f s, %d chunks" % (time.time() - t0, src.n_chunks), flush=True)
t0 = time.time(); rec, n = src.chunk(0); print("fill of chunk 0: %.3f s" % (time.time() - t0), flush=True)
del src, rec
t0 = time.time()
_, counts = fastq.align_files(eng, opt, r1, r2, out=Null(), K=100_000_000, in_flight=3)
print("in_flight=3, SAM discarded: %.2f s" % (time.time() - t0), flush=True)
for fly in (3, 1):
    out = os.path.join(wd, "e2e_out.sam")
    t0 = time.time()
    with open(out, "wb") as fo:
        _, counts = fastq.align_files(eng, opt, r1, r2, out=fo, K=100_000_000, in_flight=fly)
    dt = time.time() - t0
    md5 = hashlib.md5(open(out, "rb").read()).hexdigest()
    print("in_flight=%d: %d reads in %d chunks, %.2f s file to file = %.2f Mreads/s, SAM %.1f MB md5 %s" %
          (fly, sum(counts), len(counts), dt, sum(counts) / dt / 1e6, os.path.getsize(out) / 1e6, md5), flush=True)
