import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mpibwa_amd import simulate, api, abi
from mpibwa_amd.build import build
from oracle import pyoracle as po
build(); po.build()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 500
mode = sys.argv[2] if len(sys.argv) > 2 else "pe"
os.makedirs('/tmp/g3', exist_ok=True)
names, seqs = simulate.make_genome(360_000, 3, seed=7)
simulate.write_fasta('/tmp/g3/g.fa', names, seqs)
api.build_index('/tmp/g3/g.fa', '/tmp/g3/g.fa')
eng = api.Engine('/tmp/g3/g.fa')
ref = po.RefIndex('/tmp/g3/g.fa')
if mode == "pe":
    reads = simulate.simulate_reads(seqs, n, 150, paired=True, seed=11)
    flag = abi.MEM_F_PE
else:
    reads = simulate.simulate_reads(seqs, n, 150, paired=False, seed=12, var_len=(30, 300))
    flag = 0
ra = simulate.reads_to_ascii(reads)
t0 = time.time(); want = ref.process(ref.opt(flag=flag), ra); t1 = time.time()
got = eng.process(eng.opt(flag=flag), ra); t2 = time.time()
bad = 0
for i, (a, b) in enumerate(zip(got, want)):
    if a != b:
        bad += 1
        if bad <= 4:
            print("MISMATCH read", i)
            print(" got :", a.decode()[:600])
            print(" want:", b.decode()[:600])
print("bad", bad, "of", len(want), "ref %.2fs mine %.2fs" % (t1 - t0, t2 - t1))
print(eng.stats())
