import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from mpibwa_amd import api
from oracle import pyoracle as po
from golden_util import golden_index, kernel_vectors, ragged
import tempfile
d = tempfile.mkdtemp()
prefix = golden_index(d)
print("index built", flush=True)
eng = api.Engine(prefix, device=0)
print("uploaded", flush=True)
fm = po.OracleFM(prefix)
kv = kernel_vectors()
reads = ragged(kv, "intv_reads")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
if len(sys.argv) > 2:
    reads = [reads[int(sys.argv[2])]]; n = 1
got, ms, nb = eng.smem(eng.opt(), reads[:n], cap=512)
print("smem ran", ms, flush=True)
bad = 0
for ri, (r, g) in enumerate(zip(reads[:n], got)):
    w = fm.collect_intv(r)
    if g.shape != w.shape or not (g == w).all():
        bad += 1
        print("BAD read", ri, "len", len(r), "n_got", len(g), "n_want", len(w))
        if bad < 3:
            print("read", "".join("ACGTN"[int(v)] for v in r))
            f = lambda a: [(int(x[0]), int(x[1]), int(x[2]), int(x[3]) >> 32, int(x[3]) & 0xffffffff) for x in a]
            print("got", f(g)); print("want", f(w))
print("bad", bad, "of", n, flush=True)
