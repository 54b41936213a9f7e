"""Stage-level benchmark of the SMEM / SA kernels on the GRCh38-size synthetic index (for rocprofv3 / PMC runs)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpibwa_amd import api, bigindex, abi
mbp = float(sys.argv[1]) if len(sys.argv) > 1 else 3100
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 333334
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
os.makedirs("/tmp/mpibwa_bench", exist_ok=True)
idx = bigindex.make_or_get("/tmp/mpibwa_bench", genome_mbp=mbp, seed=38, log=lambda *a: print(*a, flush=True),
                           model=os.environ.get("MPIBWA_BENCH_GENOME_MODEL", "grch38like"), repeat_frac=float(os.environ.get("MPIBWA_BENCH_REPEAT_FRAC", "0.05")))
eng = idx.engine
reads = idx.simulate_pairs(pairs, seed=1000)
tr = bytes.maketrans(b"ACGTN", bytes([0, 1, 2, 3, 4]))
seqs = []
for n, a, b in reads:
    seqs.append(np.frombuffer(a.translate(tr), dtype=np.uint8)); seqs.append(np.frombuffer(b.translate(tr), dtype=np.uint8))
opt = eng.opt()
if os.environ.get("SMEM_ONLY_HALF"):
    seqs = seqs[:len(seqs) // 2]
for r in range(reps):
    out, ms, nb = eng.smem(opt, seqs, cap=int(os.environ.get("SMEM_CAP", "96")))
    print("smem: %.2f ms  %.1f GB/s algorithmic (%.1f KB/read)" % (ms, nb / ms / 1e6, nb / len(seqs) / 1e3), flush=True)
if os.environ.get("SMEM_ONLY"):
    sys.exit(0)
rows = np.concatenate([o[:, 0] for o in out[:200000] if len(o)])[:8000000]
for r in range(reps):
    sa, ms, nb = eng.sa(rows)
    print("sa: %d lookups %.2f ms  %.1f GB/s algorithmic" % (len(rows), ms, nb / ms / 1e6), flush=True)
