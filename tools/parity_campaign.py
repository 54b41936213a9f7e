"""Randomised parity campaign at full index size: fresh read sets (length, error rates, fragment size, options drawn per
case) through mem_process_seqs on the GPU and through the compiled reference on the host; every SAM record must be equal.
usage: python tools/parity_campaign.py [cases=12] [pairs=60000] [rng seed=2026]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
from mpibwa_amd import abi, api, bigindex
from oracle import pyoracle as po

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
wd = "/tmp/mpibwa_bench"
os.makedirs(wd, exist_ok=True)
idx = bigindex.make_or_get(wd, genome_mbp=3100, seed=38, log=lambda *a: None)
eng = idx.engine
ref = po.RefIndex(idx.prefix)
for lib in (eng.lib, ref.lib):
    C.c_int.in_dll(lib, "bwa_verbose").value = 1
cores = int(eng.lib.mi355x_host_cpus())
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 2026)
bad = 0
for c in range(cases):
    L = int(rng.choice([76, 100, 125, 150, 150, 200, 250]))
    sub = float(rng.choice([0.002, 0.01, 0.03]))
    indel = float(rng.choice([0.0002, 0.001, 0.004]))
    fm = float(rng.uniform(2.0, 3.5)) * L
    kw = dict(flag=abi.MEM_F_PE)
    pick = int(rng.integers(0, 10))
    if pick == 1: kw.update(T=int(rng.integers(20, 60)))
    if pick == 2: kw.update(flag=abi.MEM_F_PE | abi.MEM_F_NO_RESCUE)
    if pick == 3: kw.update(max_matesw=int(rng.integers(5, 100)), w=int(rng.integers(40, 160)))
    if pick == 4: kw.update(flag=abi.MEM_F_PE | abi.MEM_F_ALL)
    if pick == 5: kw.update(flag=abi.MEM_F_PE | abi.MEM_F_SOFTCLIP | abi.MEM_F_NO_MULTI, pen_unpaired=int(rng.integers(5, 30)))
    if pick == 6: kw.update(o_del=int(rng.integers(3, 9)), e_del=int(rng.integers(1, 3)), o_ins=int(rng.integers(3, 9)), e_ins=int(rng.integers(1, 3)), pen_clip5=int(rng.integers(0, 8)))
    if pick == 7: kw.update(XA_drop_ratio=float(rng.choice([0.5, 0.9])), mask_level=float(rng.choice([0.3, 0.7])), min_seed_len=int(rng.integers(15, 24)))
    if pick == 8: kw.update(flag=abi.MEM_F_PE | abi.MEM_F_PRIMARY5 | abi.MEM_F_KEEP_SUPP_MAPQ, drop_ratio=float(rng.choice([0.3, 0.6])))
    if pick == 9: kw.update(mapQ_coef_len=float(rng.choice([0, 30, 80])), max_XA_hits=int(rng.integers(1, 8)))
    reads = idx.simulate_pairs(pairs, seed=7000 + c, read_len=L, frag_mean=fm, frag_sd=fm / 8, sub=sub, indel=indel,
                               frac_random=float(rng.choice([0.0, 0.02, 0.1])))
    shape = int(rng.integers(0, 4))
    if shape == 1:      # quality-trimmed reads: ragged lengths down to 30 bp
        cut = rng.integers(30, L + 1, size=(len(reads), 2))
        reads = [(n, a[:int(c[0])], b[:int(c[1])]) for (n, a, b), c in zip(reads, cut)]
    if shape == 2:      # single-end
        reads = [(n, a, None) for n, a, b in reads]
        kw["flag"] = kw["flag"] & ~abi.MEM_F_PE
    kw_show = dict(kw, shape=["pe", "pe_trimmed", "se", "pe"][shape])
    t0 = time.time()
    got = eng.process(eng.opt(n_threads=cores, **kw), reads)
    t1 = time.time()
    want = ref.process(ref.opt(n_threads=cores, **kw), reads)
    t2 = time.time()
    n_bad = sum(1 for a, b in zip(got, want) if a != b) + abs(len(got) - len(want))
    bad += n_bad
    print("case %2d: L=%d sub=%.3f indel=%.4f frag=%.0f %s -> %d / %d records differ (GPU %.2f s, reference %.2f s)" %
          (c, L, sub, indel, fm, kw_show, n_bad, len(want), t1 - t0, t2 - t1), flush=True)
print("TOTAL differing records:", bad)
sys.exit(1 if bad else 0)
