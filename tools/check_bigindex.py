"""Build a big synthetic index on the GPU and validate it on the host with the oracle: random genome substrings must be
found by backward search and their SA values must point at an identical substring."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpibwa_amd import api, bigindex
from mpibwa_amd.build import build
from oracle import pyoracle as po
build(); po.build()
mbp = float(sys.argv[1])
lib = api.load_library()
d = "/tmp/chk_%d" % int(mbp); os.makedirs(d, exist_ok=True)
pac, lens = bigindex.synth_packed_genome(mbp * 1e6, seed=38)
prefix = os.path.join(d, "g.fa")
bigindex.write_meta_files(prefix, pac, lens)
secs = C.c_double(0)
lib.mi355x_index_build_gpu(0, pac.ctypes.data, int(lens.sum()), prefix.encode(), C.byref(secs))
print("built %.0f Mbp in %.1f s" % (mbp, secs.value), flush=True)
fm = po.OracleFM(prefix)
L = int(lens.sum()); N = 2 * L
print("seq_len", fm.fm.seq_len, "expected", N, "primary", fm.fm.primary, "L2", list(fm.fm.L2))
assert fm.fm.seq_len == N
rng = np.random.default_rng(1)
bad = 0
for it in range(300):
    p = int(rng.integers(0, L - 60))
    s = bigindex.unpack_windows(pac, np.array([p], dtype=np.int64), 50)[0]
    iv = fm.collect_intv(s)
    full = [r for r in iv if (int(r[3]) >> 32) == 0 and (int(r[3]) & 0xffffffff) == 50]
    if not full:
        bad += 1; print("no full-length SMEM for pos", p); continue
    x0, x1, x2 = int(full[0][0]), int(full[0][1]), int(full[0][2])
    hits = [fm.sa_lookup(x0 + k) for k in range(min(x2, 5))]
    ok = False
    for h in hits:
        if h < 0 or h + 50 > N: continue
        if h < L: t = bigindex.unpack_windows(pac, np.array([h], dtype=np.int64), 50)[0]
        else:
            f = N - h - 50
            t = (3 - bigindex.unpack_windows(pac, np.array([f], dtype=np.int64), 50)[0])[::-1]
        if (t == s).all(): ok = True
    if not ok:
        bad += 1; print("SA mismatch for pos", p, hits[:3])
print("bad", bad, "of 300")
