"""Mutated FASTQ text (cut lines, missing '+', records without a final newline, stray '@', random bytes, empty files) through the caller's side —
mi355x_fastq_scan, mi355x_fastq_chunks, mi355x_fastq_fill in all three modes — on the sanitizer build of the host sources (see
tools/fuzz_sampost.py for the command line).  Nothing may crash or trip a sanitizer: a malformed record is reported by its position."""
import ctypes as C, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mpibwa_amd import abi, api
lib = api.load_library()
rnd = random.Random(int(os.environ.get("FUZZ_SEED", "7")))


def fastq(n, trimmed):
    out = []
    for i in range(n):
        L = rnd.randrange(1, 160) if trimmed else 100
        name = b"@r%d" % i + (b"/1" if rnd.random() < 0.3 else b"") + (b" comment %d" % i if rnd.random() < 0.4 else b"")
        out.append(name + b"\n" + bytes(rnd.choice(b"ACGTN") for _ in range(L)) + b"\n+\n" + bytes(rnd.randrange(33, 75) for _ in range(L)) + b"\n")
    return b"".join(out)


def mutate(t):
    b = bytearray(t)
    k = rnd.randrange(9)
    if not b:
        return bytes(b)
    if k == 0: del b[rnd.randrange(len(b)):]
    elif k == 1:
        i = rnd.randrange(len(b)); del b[i:i + rnd.randrange(1, 60)]
    elif k == 2: b = b.replace(b"\n+\n", b"\n", 1)
    elif k == 3: b = b.replace(b"\n", b"", rnd.randrange(1, 4))
    elif k == 4:
        for _ in range(rnd.randrange(1, 8)): b[rnd.randrange(len(b))] = rnd.randrange(1, 256)
    elif k == 5: b = b[:-1]
    elif k == 6: b = b"\n\n" + b
    elif k == 7: b = bytearray(b"@")
    return bytes(b)


n_good = n_bad = 0
for it in range(int(os.environ.get("FUZZ_N", "4000"))):
    n = rnd.randrange(1, 12)
    paired, trimmed = rnd.random() < 0.7, rnd.random() < 0.5
    texts = [fastq(n, trimmed), fastq(n, trimmed) if paired else None]
    texts = [mutate(t) if t is not None and rnd.random() < 0.6 else t for t in texts]
    bufs, offs, counts = [], [], []
    ok = True
    for t in texts:
        if t is None:
            bufs.append(None); offs.append(None); continue
        buf = np.frombuffer(t + b"\0", dtype=np.uint8).copy()
        cap = len(t) // 4 + 4
        off = np.zeros(cap + 1, dtype=np.int64); bases = np.zeros(cap, dtype=np.int32)
        r = lib.mi355x_fastq_scan(buf.ctypes.data, len(t), cap, off.ctypes.data, bases.ctypes.data)
        if r < 0:
            ok = False
        bufs.append(buf); offs.append(off); counts.append((r, bases))
    if not ok or (paired and counts[0][0] != counts[1][0]) or counts[0][0] <= 0:
        n_bad += 1
        continue
    cnt = counts[0][0]
    first = np.zeros(cnt + 2, dtype=np.int64)
    lib.mi355x_fastq_chunks(counts[0][1].ctypes.data, counts[1][1].ctypes.data if paired and trimmed else None, cnt, rnd.randrange(1, 600), cnt + 1, first.ctypes.data)
    seqs = (abi.bseq1_t * (cnt * (2 if paired else 1)))()
    r = lib.mi355x_fastq_fill(bufs[0].ctypes.data, offs[0].ctypes.data, bufs[1].ctypes.data if paired else None, offs[1].ctypes.data if paired else None, 0, cnt,
                              rnd.randrange(2), int(paired and not trimmed and rnd.random() < 0.7), seqs)
    if r >= 0:
        n_good += 1
        for s in seqs:   # every string the caller will read is inside its buffer and terminated
            for p in (s.name, s.seq, s.qual, s.comment):
                if p:
                    C.string_at(p)
    else:
        n_bad += 1
print("filled", n_good, "refused", n_bad)
