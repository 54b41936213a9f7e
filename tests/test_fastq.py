"""Caller-side row, host logic only: mpiBWA's FASTQ record scan, chunk rule and bseq1_t filling."""
import ctypes as C

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
