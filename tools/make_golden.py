"""Generate tests/golden/* from the REAL reference (oracle/_ref/libbwaref.so, built from /root/reference by
oracle/Makefile).  Run in the build container only; the outputs are committed.  Fixtures are data: inputs
(seeded synthetic genome / reads, random DP tuples) and the reference's outputs for them."""
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mpibwa_amd import simulate, api, abi  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")
os.makedirs(G, exist_ok=True)
tmp = "/tmp/golden_build"
os.makedirs(tmp, exist_ok=True)

# ---- genome (stored, so the fixture does not depend on the RNG implementation) ----
names, seqs = simulate.make_genome(120_000, 2, seed=101, n_runs=1)
fa = os.path.join(tmp, "gold.fa")
simulate.write_fasta(fa, names, seqs)
with open(fa, "rb") as f, gzip.open(os.path.join(G, "genome.fa.gz"), "wb", 9) as g:
    g.write(f.read())
api.build_index(fa, fa)
ref = po.RefIndex(fa)

# ---- end-to-end SAM: PE 2x150, SE variable length ----
pe = simulate.reads_to_ascii(simulate.simulate_reads(seqs, 300, 150, paired=True, seed=5))
se = simulate.reads_to_ascii(simulate.simulate_reads(seqs, 200, 150, paired=False, seed=6, var_len=(25, 300)))


def dump_reads(path, reads):
    with gzip.open(path, "wt") as g:
        for n, a, b in reads:
            g.write(n + "\t" + a.decode() + "\t" + ("" if b is None else b.decode()) + "\n")


dump_reads(os.path.join(G, "reads_pe150.tsv.gz"), pe)
dump_reads(os.path.join(G, "reads_se_var.tsv.gz"), se)
cases = {
    "pe_default": (pe, dict(flag=abi.MEM_F_PE)),
    "pe_M_Y": (pe, dict(flag=abi.MEM_F_PE | abi.MEM_F_NO_MULTI | abi.MEM_F_SOFTCLIP)),
    "pe_all": (pe, dict(flag=abi.MEM_F_PE | abi.MEM_F_ALL)),
    "pe_norescue_nopair": (pe, dict(flag=abi.MEM_F_PE | abi.MEM_F_NO_RESCUE | abi.MEM_F_NOPAIRING)),
    "se_default": (se, dict(flag=0)),
    "se_primary5_T20": (se, dict(flag=abi.MEM_F_PRIMARY5 | abi.MEM_F_KEEP_SUPP_MAPQ, T=20)),
}
for name, (reads, kw) in cases.items():
    sam = ref.process(ref.opt(**kw), reads)
    with gzip.open(os.path.join(G, "sam_%s.txt.gz" % name), "wb", 9) as g:
        g.write(b"".join(sam))
json.dump({k: {kk: vv for kk, vv in v[1].items()} for k, v in cases.items()}, open(os.path.join(G, "sam_cases.json"), "w"), indent=1)

# ---- kernel-level vectors ----
opt = ref.opt()
rng = np.random.default_rng(77)
intv_reads, intv_out = [], []
for n, a, b in pe[:60] + se[:60]:
    for s in (a, b):
        if s is None:
            continue
        codes = np.frombuffer(s.translate(bytes.maketrans(b"ACGTN", bytes([0, 1, 2, 3, 4]))), dtype=np.uint8).copy()
        iv = ref.collect_intv(opt, codes.copy())
        iv = np.array(sorted(map(tuple, iv.tolist()), key=lambda r: (r[3], r[0], r[1], r[2])), dtype=np.uint64).reshape(-1, 4)
        intv_reads.append(codes)
        intv_out.append(iv)
fm = po.OracleFM(fa)
ks = np.concatenate([rng.integers(0, fm.fm.seq_len + 1, size=800).astype(np.uint64),
                     np.array([0, 1, fm.fm.primary, fm.fm.primary + 1, fm.fm.seq_len], dtype=np.uint64)])
sa = np.array([ref.sa_lookup(int(k)) for k in ks], dtype=np.uint64)
mat = np.array(list(opt.contents.mat), dtype=np.int8)


def rand_pair(qlen, div):
    q = rng.integers(0, 4, size=qlen, dtype=np.uint8)
    out = []
    for b in q:
        u = rng.random()
        if u < div:
            out.append((b + 1 + rng.integers(0, 3)) & 3)
        elif u < div * 1.3:
            continue
        elif u < div * 1.6:
            out += [b, rng.integers(0, 4)]
        else:
            out.append(b)
    out += list(rng.integers(0, 4, size=rng.integers(0, 50)))
    return q, np.array(out if out else [0], dtype=np.uint8)


ext_q, ext_t, ext_p, ext_o = [], [], [], []
for it in range(500):
    qlen = int(rng.choice([1, 5, 63, 64, 65, 128, 131, 150, 231, 290, int(rng.integers(1, 300))]))
    q, t = rand_pair(qlen, float(rng.choice([0.0, 0.02, 0.1, 0.3])))
    if rng.random() < 0.1:
        q[rng.integers(0, qlen)] = 4
    w, h0, eb, zd = int(rng.choice([100, 200, 9, 33])), int(rng.integers(1, 170)), int(rng.choice([5, 0])), int(rng.choice([100, 100, 0, 15]))
    o = ref.extend2(q, t, mat, 6, 1, 6, 1, w, eb, zd, h0)
    ext_q.append(q); ext_t.append(t); ext_p.append([w, h0, eb, zd]); ext_o.append(o)
glo_q, glo_t, glo_w, glo_s, glo_c = [], [], [], [], []
for it in range(200):
    qlen = int(rng.integers(1, 180))
    q, t = rand_pair(qlen, float(rng.choice([0.0, 0.03, 0.1])))
    t = t[:max(1, len(q) + int(rng.integers(-6, 7)))]
    w = int(abs(len(t) - len(q)) + rng.integers(3, 30))
    sc, cg = ref.global2(q, t, mat, 6, 1, 6, 1, w)
    glo_q.append(q); glo_t.append(t); glo_w.append(w); glo_s.append(sc); glo_c.append(cg)


def ragged(lst, dtype):
    off = np.zeros(len(lst) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(x) for x in lst])
    flat = np.concatenate([np.asarray(x, dtype=dtype).reshape(-1) for x in lst]) if lst else np.zeros(0, dtype)
    return flat, off


kv = {}
kv["intv_reads"], kv["intv_reads_off"] = ragged(intv_reads, np.uint8)
kv["intv_out"], kv["intv_out_off"] = ragged([x.reshape(-1) for x in intv_out], np.uint64)
kv["sa_k"], kv["sa_v"] = ks, sa
kv["ext_q"], kv["ext_q_off"] = ragged(ext_q, np.uint8)
kv["ext_t"], kv["ext_t_off"] = ragged(ext_t, np.uint8)
kv["ext_p"], kv["ext_o"] = np.array(ext_p, dtype=np.int32), np.array(ext_o, dtype=np.int32)
kv["glo_q"], kv["glo_q_off"] = ragged(glo_q, np.uint8)
kv["glo_t"], kv["glo_t_off"] = ragged(glo_t, np.uint8)
kv["glo_w"], kv["glo_s"] = np.array(glo_w, dtype=np.int32), np.array(glo_s, dtype=np.int32)
kv["glo_c"], kv["glo_c_off"] = ragged(glo_c, np.uint32)
np.savez_compressed(os.path.join(G, "kernel_vectors.npz"), **kv)
print("golden written:", sorted(os.listdir(G)))
