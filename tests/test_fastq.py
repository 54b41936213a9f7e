"""Caller-side row, host logic only: mpiBWA's FASTQ record scan, chunk rule and bseq1_t filling."""
import ctypes as C
import os

import numpy as np
import pytest

from mpibwa_amd import abi, api, fastq


@pytest.fixture(scope="module")
def lib(built):
    return api.load_library()


def _write(tmp_path, name, recs, final_newline=True):
    txt = b"".join(b"@" + h + b"\n" + s + b"\n+\n" + q + b"\n" for h, s, q in recs)
    if not final_newline:
        txt = txt[:-1]
    p = tmp_path / name
    p.write_bytes(txt)
    return str(p)


def _rule(b1, b2, maxsiz):
    """independent restatement of src/parallel_aux.c:1520-1546 / 1056-1095"""
    starts, cnt, open_ = [], 0, False
    for i in range(len(b1)):
        if not open_:
            starts.append(i); open_ = True
        cnt += int(b1[i]) + (int(b2[i]) if b2 is not None else 0)
        if cnt > maxsiz:
            cnt, open_ = 0, False
    return starts + [len(b1)]


def test_chunk_rule_known_structure(lib):
    # 20 000 pairs of 150 bp with -K 1000000 -> 6 chunks of 6 668 / 6 660 reads (reference run)
    b = np.full(20000, 150, dtype=np.int32)
    st = fastq.chunk_starts(lib, b, None, 1_000_000 // 2)
    assert (np.diff(st) * 2).tolist() == [6668] * 5 + [6660]


def test_chunk_rule_strict_greater_and_random(lib):
    # a chunk closes when the count EXCEEDS maxsiz: reads summing exactly to it stay open for one more read
    b = np.array([100, 100, 100, 100, 100], dtype=np.int32)
    assert fastq.chunk_starts(lib, b, None, 200).tolist() == [0, 3, 5]
    assert fastq.chunk_starts(lib, b, None, 199).tolist() == [0, 2, 4, 5]
    assert fastq.chunk_starts(lib, b, None, 10 ** 9).tolist() == [0, 5]
    assert fastq.chunk_starts(lib, b[:0], None, 100).tolist() == [0]
    rng = np.random.default_rng(5)
    for _ in range(50):
        n = int(rng.integers(1, 400))
        b1 = rng.integers(30, 301, size=n).astype(np.int32)
        b2 = rng.integers(30, 301, size=n).astype(np.int32)
        m = int(rng.integers(100, 5000))
        assert fastq.chunk_starts(lib, b1, None, m).tolist() == _rule(b1, None, m)
        assert fastq.chunk_starts(lib, b1, b2, m).tolist() == _rule(b1, b2, m)


def _strings(rec, n):
    out = []
    for i in range(n):
        r = rec[i]
        out.append((C.string_at(int(r["name"])), C.string_at(int(r["comment"])) if r["comment"] else None,
                    C.string_at(int(r["seq"])), C.string_at(int(r["qual"])), int(r["l_seq"])))
    return out


def test_scan_and_fill_like_mpibwa_main(lib, tmp_path):
    r1 = [(b"r1/1 1:N:0:1", b"ACGTN", b"IIIII"), (b"r2 x y", b"ACG", b"#I#"), (b"r3", b"GATTACA", b"1234567")]
    r2 = [(b"r1/2 2:N:0:1", b"TTTTT", b"HHHHH"), (b"r2 x z", b"CCC", b"ABC"), (b"r3", b"TGTAATC", b"7654321")]
    p1, p2 = _write(tmp_path, "a_1.fq", r1), _write(tmp_path, "a_2.fq", r2, final_newline=False)
    src = fastq.FastqSource(lib, p1, p2, K=10 ** 7, copy_comment=True, mode="pe_trim")
    assert src.f1.n == 3 and src.f1.bases.tolist() == [5, 3, 7] and src.f2.bases.tolist() == [5, 3, 7]
    assert src.n_chunks == 1
    rec, n = src.chunk(0)
    assert n == 6
    got = _strings(rec, n)
    assert got[0] == (b"r1", b"1:N:0:1", b"ACGTN", b"IIIII", 5)     # "/1" dropped, comment = rest of the line
    assert got[1] == (b"r1", b"2:N:0:1", b"TTTTT", b"HHHHH", 5)
    assert got[2] == (b"r2", b"x y", b"ACG", b"#I#", 3)
    assert got[3] == (b"r2", b"x z", b"CCC", b"ABC", 3)
    assert got[4] == (b"r3", b"", b"GATTACA", b"1234567", 7)        # no white space: empty comment
    assert got[5] == (b"r3", b"", b"TGTAATC", b"7654321", 7)        # last record without a final newline
    # equal-size mode walks R2's header with R1's pointer (src/mainParallel.c:1274-1276): same cut positions
    src = fastq.FastqSource(lib, p1, _write(tmp_path, "a_2b.fq", r2), K=10 ** 7, copy_comment=False)
    assert src.mode == "pe"
    rec, n = src.chunk(0)
    got = _strings(rec, n)
    assert [g[0] for g in got] == [b"r1", b"r1", b"r2", b"r2", b"r3", b"r3"] and all(g[1] is None for g in got)
    # single end, several chunks: K counts the bases of the one file
    src = fastq.FastqSource(lib, p1, None, K=5)
    assert src.mode == "se" and np.diff(src.starts).tolist() == [2, 1]
    rec, n = src.chunk(1)
    assert _strings(rec, n) == [(b"r3", None, b"GATTACA", b"1234567", 7)]


def test_malformed_records_are_reported(lib, tmp_path):
    p = tmp_path / "bad.fq"
    p.write_bytes(b"@a\nACGT\n+\nIIII\nXb\nAC\n+\nII\n")
    with pytest.raises(ValueError):
        fastq.FastqFile(lib, str(p))
    p.write_bytes(b"@a\nACGT\n-\nIIII\n")
    with pytest.raises(ValueError):
        fastq.FastqFile(lib, str(p))


@pytest.mark.skipif(not __import__("oracle.pyoracle", fromlist=["x"]).ref_available(), reason="oracle/_ref/libbwaref.so not built")
def test_example_fastq_through_reference_library_gives_the_binary_md5(lib, tmp_path):
    """Pins the caller-side row on the CPU: our record scan / chunking / bseq1_t filling in front of the compiled reference's
    mem_process_seqs reproduces the SAM body the real `mpiBWA mem` binary wrote for its example data."""
    import hashlib
    import os
    import tarfile
    from oracle import pyoracle as po
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpibwa_examples")
    with tarfile.open(os.path.join(here, "hg19.small.tar.gz")) as t:
        t.extractall(tmp_path)
    ref = po.RefIndex(str(tmp_path / "hg19.small.fa"))
    C.c_int.in_dll(ref.lib, "bwa_verbose").value = 1
    src = fastq.FastqSource(lib, os.path.join(here, "HCC1187C_R1_10K.fastq.gz"), os.path.join(here, "HCC1187C_R2_10K.fastq.gz"))
    assert src.mode == "pe" and src.n_chunks == 1
    rec, n = src.chunk(0)
    ref.lib.mem_process_seqs(ref.opt(flag=abi.MEM_F_PE, n_threads=8), ref.bwt, ref.bns, ref.pac, 0, n,
                             C.cast(rec.ctypes.data, C.POINTER(abi.bseq1_t)), None)
    h = hashlib.md5()
    for i in range(n):
        p = int(rec[i]["sam"])
        h.update(C.string_at(p))
        po.libc.free(C.c_void_p(p))
    assert h.hexdigest() == "51ce7ba0592d4a199eac49526b6c9d8c"


def test_threaded_fill_matches_record_by_record(lib, tmp_path):
    """Chunks of 65 536 records and more are filled by several threads over contiguous ranges: same strings as the
    per-record walk, same base count, and a malformed record is reported by its index whichever range it falls into."""
    rng = np.random.default_rng(3)
    n = 70_000
    lens = rng.integers(30, 80, size=n)
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    recs1, recs2 = [], []
    for i in range(n):
        s = lut[rng.integers(0, 5, size=int(lens[i]))].tobytes()
        q = bytes(rng.integers(33, 74, size=int(lens[i]), dtype=np.uint8))
        q = q.replace(b"@", b"A")
        hdr = b"read%d" % i + (b"/1 c%d" % i if i % 3 == 0 else b"")
        recs1.append((hdr, s, q))
        recs2.append((hdr.replace(b"/1", b"/2"), s[::-1], q[::-1]))
    p1, p2 = _write(tmp_path, "big_1.fq", recs1), _write(tmp_path, "big_2.fq", recs2, final_newline=False)
    src = fastq.FastqSource(lib, p1, p2, K=10 ** 9, copy_comment=True, mode="pe_trim")
    assert src.n_chunks == 1 and src.f1.n == n
    rec, cnt = src.chunk(0)
    assert cnt == 2 * n
    got = _strings(rec, cnt)
    for i in range(0, n, 997):
        h, s, q = recs1[i]
        name = h.split(b"/")[0].split(b" ")[0]
        com = h.split(b" ", 1)[1] if b" " in h else b""
        assert got[2 * i] == (name, com, s, q, len(s)), i
        assert got[2 * i + 1] == (name, com, s[::-1], q[::-1], len(s)), i
    assert sum(g[4] for g in got) == 2 * int(lens.sum())
    # a record without its '+' line, deep inside the file
    bad = list(recs1)
    k = 54_321
    txt = b"".join(b"@" + h + b"\n" + s + b"\n" + (b"-" if i == k else b"+") + b"\n" + q + b"\n" for i, (h, s, q) in enumerate(bad))
    pb = tmp_path / "bad_big.fq"
    pb.write_bytes(txt)
    with pytest.raises(ValueError):
        fastq.FastqFile(lib, str(pb))


def _stub_engine(lib, seen, delay=True):
    """An engine whose C ABI file functions are the real ones and whose mem_process_seqs is a stand-in that writes
    "<name>\t<n_processed + i>" per read and finishes every other call late."""
    import threading
    import time
    libc = C.CDLL(None if os.environ.get("MPIBWA_SANITIZER_LIB") else "libc.so.6")   # (under tools/san_host.sh: the sanitizer's malloc / free)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    lock = threading.Lock()

    class StubLib:
        def __getattr__(self, name):
            return getattr(lib, name)

        def mem_process_seqs(self, opt, bwt, bns, pac, n_processed, n, seqs, pes0):
            with lock:
                seen.append((int(n_processed), int(n)))
                k = len(seen)
            if delay:
                time.sleep(0.05 if k % 2 else 0.0)        # odd calls finish late
            for i in range(n):
                line = b"%s\t%d\n" % (C.string_at(seqs[i].name), int(n_processed) + i)
                p = libc.malloc(len(line) + 1)
                C.memmove(p, line, len(line) + 1)
                seqs[i].sam = p

    class StubEngine:
        bwt = bns = pac = None
    StubEngine.lib = StubLib()
    return StubEngine()


def test_align_files_writes_chunks_in_order_whatever_the_completion_order(lib, tmp_path):
    """align_files with a stand-in engine that finishes chunks out of order: the SAM body is in chunk order, every record
    once, n_processed is passed as the trimmed branch does."""
    recs = [(b"r%d" % i, b"ACGT" * 5, b"I" * 20) for i in range(400)]
    p1 = _write(tmp_path, "o_1.fq", recs)
    p2 = _write(tmp_path, "o_2.fq", [(h, s[:-1], q[:-1]) for h, s, q in recs])    # different size -> trimmed branch
    seen = []
    body, counts = fastq.align_files(_stub_engine(lib, seen), None, p1, p2, out=None, K=2000, in_flight=3)
    lines = body.split(b"\n")[:-1]
    assert len(lines) == 800 and sum(counts) == 800 and len(counts) > 5
    assert [l.split(b"\t")[0] for l in lines] == [b"r%d" % (i // 2) for i in range(800)]
    assert [int(l.split(b"\t")[1]) for l in lines] == list(range(800))           # running n_processed of the trimmed branch
    assert sorted(seen) == sorted((sum(counts[:k]), counts[k]) for k in range(len(counts)))


def test_align_files_rank_striping_covers_every_chunk_once(lib, tmp_path):
    """world = 3: rank r takes chunks r, r + 3, ... of the same chunk list; together the ranks emit every record once and a
    rank's records are those of its chunks, in order.  Equal-size pairs: n_processed is 0 for every chunk (src/mainParallel.c:1314)."""
    recs = [(b"q%d" % i, b"ACGTA" * 6, b"J" * 30) for i in range(500)]
    p1, p2 = _write(tmp_path, "s_1.fq", recs), _write(tmp_path, "s_2.fq", recs)
    seen = []
    whole, counts = fastq.align_files(_stub_engine(lib, seen, delay=False), None, p1, p2, out=None, K=3000, in_flight=2)
    assert all(npz == 0 for npz, _ in seen)
    names = [l.split(b"\t")[0] for l in whole.split(b"\n")[:-1]]
    assert names == [b"q%d" % (i // 2) for i in range(1000)] and len(counts) >= 7
    starts = np.concatenate([[0], np.cumsum(counts)])
    for world in (2, 3):
        union = {}
        for rank in range(world):
            body, cr = fastq.align_files(_stub_engine(lib, [], delay=False), None, p1, p2, out=None, K=3000, rank=rank, world=world)
            mine = list(range(rank, len(counts), world))
            assert cr == [counts[c] for c in mine]
            got = [l.split(b"\t")[0] for l in body.split(b"\n")[:-1]]
            want = [n for c in mine for n in names[starts[c]:starts[c + 1]]]
            assert got == want
            for c in mine:
                assert c not in union
                union[c] = True
        assert sorted(union) == list(range(len(counts)))
