"""Generate tests/golden/fixmate_cases.json.gz from the REAL reference: the SAM text the reference's mem_process_seqs writes for pairs of
every kind on the committed golden genome (tests/golden/genome.fa.gz), and what the reference's own fixmate() (src/fixmate.c compiled in
place: oracle/_ref/libfixmateref.so) makes of each pair.  Run in the build container only; the output is committed.  Data only: input
texts and the reference's output texts."""
import ctypes as C
import gzip
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mpibwa_amd import abi, api, simulate  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from golden_util import golden_index  # noqa: E402
from test_sampost import _pairs_of_every_kind, _with_qualities, _Pair  # noqa: E402

tmp = "/tmp/golden_fixmate"
os.makedirs(tmp, exist_ok=True)
fa = golden_index(tmp)
ref = po.RefIndex(fa)
names = [ref.bns.contents.anns[i].name.decode() for i in range(ref.bns.contents.n_seqs)]
seqs = []
with gzip.open(os.path.join(ROOT, "tests", "golden", "genome.fa.gz"), "rt") as g:
    cur = []
    for ln in g:
        if ln.startswith(">"):
            if cur:
                seqs.append(np.frombuffer("".join(cur).encode(), np.uint8))
            cur = []
        else:
            cur.append(ln.strip())
    seqs.append(np.frombuffer("".join(cur).encode(), np.uint8))
lut = np.full(256, 4, np.uint8)
for i, c in enumerate(b"ACGT"):
    lut[c] = i
seqs = [lut[s] for s in seqs]
fx = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libfixmateref.so"))
fx.fixmate.argtypes = [C.c_int, C.POINTER(abi.bseq1_t), C.POINTER(abi.bseq1_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(abi.bwaidx_t)]
reads = simulate.reads_to_ascii(_pairs_of_every_kind({"seqs": seqs}, n=96, seed=31))
rng = np.random.default_rng(32)
cases = []
for tag, flag in (("default", abi.MEM_F_PE), ("M", abi.MEM_F_PE | abi.MEM_F_NO_MULTI), ("a", abi.MEM_F_PE | abi.MEM_F_ALL)):
    sams = ref.process(ref.opt(flag=flag), reads)
    for p, (name, _, _) in enumerate(reads):
        s1, s2 = _with_qualities(sams[2 * p], rng), _with_qualities(sams[2 * p + 1], rng)
        pr = _Pair(name.encode(), s1, s2)
        a, b = C.c_int(0), C.c_int(0)
        assert fx.fixmate(0, C.byref(pr.arr[0]), C.byref(pr.arr[1]), C.byref(a), C.byref(b), ref.idx) == 0
        o1, o2 = pr.take()
        cases.append({"opt": tag, "name": name, "in": [s1.decode(), s2.decode()], "out": [o1.decode(), o2.decode()]})
out = os.path.join(ROOT, "tests", "golden", "fixmate_cases.json.gz")
with gzip.open(out, "wt", 9) as g:
    json.dump({"contigs": names, "cases": cases}, g)
print(out, len(cases), "pairs,", os.path.getsize(out), "bytes")
