import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpibwa_amd import simulate, api
from mpibwa_amd.build import build
from oracle import pyoracle as po
build(); po.build()
os.makedirs('/tmp/g2', exist_ok=True)
names, seqs = simulate.make_genome(360_000, 3, seed=7)
simulate.write_fasta('/tmp/g2/g.fa', names, seqs)
api.build_index('/tmp/g2/g.fa', '/tmp/g2/g.fa')
eng = api.Engine('/tmp/g2/g.fa')
fm = po.OracleFM('/tmp/g2/g.fa')
reads = simulate.simulate_reads(seqs, 200, 150, paired=True, seed=11)
flat = []
for _, a, b in reads: flat += [a, b]
got, ms, nb = eng.smem(eng.opt(), flat, cap=512)
nbad = 0
for i, (s, a) in enumerate(zip(flat, got)):
    b = fm.collect_intv(s)
    if a.shape != b.shape or not (a == b).all():
        nbad += 1
        if nbad <= 3:
            print("read", i, "len", len(s), a.shape, b.shape)
            for r in range(max(len(a), len(b))):
                ra = a[r] if r < len(a) else None; rb = b[r] if r < len(b) else None
                flag = "" if (ra is not None and rb is not None and (ra == rb).all()) else "  <<<"
                f = lambda x: None if x is None else (int(x[0]), int(x[1]), int(x[2]), int(x[3]) >> 32, int(x[3]) & 0xffffffff)
                print("  ", f(ra), f(rb), flag)
print("bad", nbad, "of", len(flat), "ms", ms)
