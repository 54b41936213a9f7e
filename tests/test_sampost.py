"""The passes mpiBWA runs over a chunk's SAM text behind mem_process_seqs (csrc/sampost.cpp; SURVEY §8f row 4, the caller's side):

  * -f fixmate against the reference's own fixmate() (src/fixmate.c compiled in place: oracle/_ref/libfixmateref.so) on the SAM the
    reference's mem_process_seqs writes for pairs of every kind: both ends mapped on one contig, on two contigs, one end or both
    unmapped, chimeric reads with supplementary lines, secondary lines (-a, -M), soft clips (-Y), qualities below and above the
    ms threshold;
  * -g / -b BGZF output: every block is a gzip member with the 'BC' field and the right BSIZE, blocks hold whole records, the
    decompressed stream is the text, the bytes do not depend on the thread count;
  * mpiBWAByChr's routing against a statement of the rule in Python, for pairs (with the discordant file) and single ends.

CPU only (the text is the hot path's product; these passes never touch the GPU)."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from oracle import pyoracle as po

HERE = os.path.dirname(os.path.abspath(__file__))
FIXREF = os.path.join(os.path.dirname(HERE), "oracle", "_ref", "libfixmateref.so")


@pytest.fixture(scope="module")
def lib(built):
    from mpibwa_amd import abi, api
    L = api.load_library()
    L.mi355x_fixmate_pair.restype = C.c_int
    L.mi355x_fixmate_pair.argtypes = [C.POINTER(abi.bseq1_t), C.POINTER(abi.bseq1_t), C.POINTER(abi.bntseq_t)]
    L.mi355x_fixmate.restype = C.c_int64
    L.mi355x_fixmate.argtypes = [C.POINTER(abi.bseq1_t), C.c_int, C.POINTER(abi.bntseq_t)]
    L.mi355x_bgzf_bound.restype = C.c_size_t
    L.mi355x_bgzf_bound.argtypes = [C.c_size_t]
    L.mi355x_bgzf_compress.restype = C.c_size_t
    L.mi355x_bgzf_compress.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
    L.mi355x_bgzf_eof.restype = C.c_size_t
    L.mi355x_bgzf_eof.argtypes = [C.c_void_p]
    L.mi355x_route_by_chr.restype = C.c_int64
    L.mi355x_route_by_chr.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(abi.bntseq_t), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    return L


def _pairs_of_every_kind(genome, n=260, seed=5):
    """(name, mate 1, mate 2) as code arrays: ordinary pairs, mates from two contigs, random (unmappable) mates, chimeric mates."""
    from mpibwa_amd import simulate
    rng = np.random.default_rng(seed)
    base = simulate.simulate_reads(genome["seqs"], n, 150, paired=True, seed=seed, frac_random=0.0)
    seqs = genome["seqs"]

    def piece(length):
        while True:
            c = int(rng.integers(0, len(seqs)))
            p = int(rng.integers(0, len(seqs[c]) - length))
            s = seqs[c][p:p + length]
            if (s < 4).all():
                return s.copy()
    out = []
    for k, (name, a, b) in enumerate(base):
        kind = k % 8
        if kind == 1:   # the mates come from two places (often two contigs): a pair on other contigs
            b = piece(150)
        elif kind == 2:   # one end does not map
            b = rng.integers(0, 4, 150).astype(np.uint8)
        elif kind == 3:   # neither does
            a = rng.integers(0, 4, 150).astype(np.uint8)
            b = rng.integers(0, 4, 150).astype(np.uint8)
        elif kind == 4:   # a chimeric mate 1: supplementary lines
            a = np.concatenate([a[:80], piece(70)])
        elif kind == 5:   # a chimeric mate 2 whose partner does not map
            a = rng.integers(0, 4, 150).astype(np.uint8)
            b = np.concatenate([piece(75), piece(75)])
        elif kind == 6:   # both chimeric
            a = np.concatenate([a[:70], piece(80)])
            b = np.concatenate([piece(90), b[90:]])
        out.append((name, a, b))
    return out


def _with_qualities(sam, rng):
    """The record's quality field redrawn (Phred 2..41: values on both sides of the ms threshold of 15), the same for every line of a read
    as bwa prints it — hard-clipped lines keep a quality string of their own length."""
    lines = sam.split(b"\n")[:-1]
    out = []
    for ln in lines:
        f = ln.split(b"\t")
        if f[10] != b"*":
            f[10] = bytes((rng.integers(2, 42, len(f[10])) + 33).astype(np.uint8).tolist())
        out.append(b"\t".join(f))
    return b"\n".join(out) + b"\n"


class _Pair:
    """two bseq1_t with malloc'ed .sam strings, as mem_process_seqs leaves them"""
    def __init__(self, name, sam1, sam2):
        from mpibwa_amd import abi, api
        self.libc = api.libc
        self.libc.strdup.restype = C.c_void_p
        self.libc.strdup.argtypes = [C.c_char_p]
        self.name = C.create_string_buffer(name)
        self.arr = (abi.bseq1_t * 2)()
        for k, s in enumerate((sam1, sam2)):
            self.arr[k].l_seq = 150
            self.arr[k].name = C.addressof(self.name)
            self.arr[k].sam = self.libc.strdup(s)

    def take(self):
        out = []
        for k in range(2):
            out.append(C.string_at(self.arr[k].sam))
            self.libc.free(C.c_void_p(self.arr[k].sam))
            self.arr[k].sam = None
        return out


@pytest.mark.skipif(not (po.ref_available() and os.path.exists(FIXREF)), reason="oracle/_ref/libfixmateref.so not built")
def test_fixmate_matches_the_reference_fixmate(genome, lib):
    from mpibwa_amd import abi, simulate
    ref = po.RefIndex(genome["prefix"])
    fx = C.CDLL(FIXREF)
    fx.fixmate.restype = C.c_int
    fx.fixmate.argtypes = [C.c_int, C.POINTER(abi.bseq1_t), C.POINTER(abi.bseq1_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(abi.bwaidx_t)]
    reads = simulate.reads_to_ascii(_pairs_of_every_kind(genome))
    rng = np.random.default_rng(9)
    seen = {"pair": 0, "other_contig": 0, "one_unmapped": 0, "both_unmapped": 0, "supp": 0, "secondary": 0, "lines": 0}
    variants = [dict(), dict(flag=abi.MEM_F_PE | abi.MEM_F_NO_MULTI), dict(flag=abi.MEM_F_PE | abi.MEM_F_ALL), dict(flag=abi.MEM_F_PE | abi.MEM_F_SOFTCLIP)]
    for kw in variants:
        kw.setdefault("flag", abi.MEM_F_PE)
        sams = ref.process(ref.opt(**kw), reads)
        assert len(sams) == 2 * len(reads)
        for p, (name, _, _) in enumerate(reads):
            s1, s2 = _with_qualities(sams[2 * p], rng), _with_qualities(sams[2 * p + 1], rng)
            a, b = _Pair(name.encode(), s1, s2), _Pair(name.encode(), s1, s2)
            n1, n2 = C.c_int(0), C.c_int(0)
            assert fx.fixmate(0, C.byref(a.arr[0]), C.byref(a.arr[1]), C.byref(n1), C.byref(n2), ref.idx) == 0
            got_n = lib.mi355x_fixmate_pair(C.byref(b.arr[0]), C.byref(b.arr[1]), ref.bns)
            want, got = a.take(), b.take()
            assert got_n == n1.value + n2.value == s1.count(b"\n") + s2.count(b"\n")
            assert got == want, (name, kw, s1, s2)
            for ln in (s1 + s2).split(b"\n")[:-1]:
                f = ln.split(b"\t")
                fl = int(f[1])
                seen["lines"] += 1
                if fl & 0x800:
                    seen["supp"] += 1
                elif fl & 0x100:
                    seen["secondary"] += 1
                elif (fl & 0xc) == 0xc:
                    seen["both_unmapped"] += 1
                elif fl & 0xc:
                    seen["one_unmapped"] += 1
                elif f[6] != b"=":
                    seen["other_contig"] += 1
                else:
                    seen["pair"] += 1
            # what -f is for: every primary line of a pair with a mapped mate now carries the mate's MAPQ
            for k in range(2):
                for ln in got[k].split(b"\n")[:-1]:
                    f = ln.split(b"\t")
                    if not int(f[1]) & 0x908:
                        assert any(t.startswith(b"MQ:i:") for t in f[11:]) and any(t.startswith(b"ms:i:") for t in f[11:])
    assert all(v > 20 for v in seen.values()), seen


def test_fixmate_over_a_chunk_and_text_that_is_not_a_pairs(genome, lib):
    """mi355x_fixmate (the loop over a chunk, on threads) gives what the pair-wise call gives; text that is not a pair's is reported
    and left alone."""
    from mpibwa_amd import abi, simulate
    if not po.ref_available():
        pytest.skip("oracle/_ref/libbwaref.so not built")
    ref = po.RefIndex(genome["prefix"])
    reads = simulate.reads_to_ascii(_pairs_of_every_kind(genome, n=120, seed=6))
    sams = ref.process(ref.opt(flag=abi.MEM_F_PE), reads)
    pairs = [_Pair(name.encode(), sams[2 * p], sams[2 * p + 1]) for p, (name, _, _) in enumerate(reads)]
    one = []
    for pr in pairs:
        q = _Pair(pr.name.value, C.string_at(pr.arr[0].sam), C.string_at(pr.arr[1].sam))
        assert lib.mi355x_fixmate_pair(C.byref(q.arr[0]), C.byref(q.arr[1]), ref.bns) > 0
        one += q.take()
    arr = (abi.bseq1_t * (2 * len(pairs)))()
    for p, pr in enumerate(pairs):
        for k in range(2):
            arr[2 * p + k] = pr.arr[k]
    n_lines = lib.mi355x_fixmate(arr, 2 * len(pairs), ref.bns)
    assert n_lines == sum(s.count(b"\n") for s in sams)
    got = [C.string_at(arr[i].sam) for i in range(2 * len(pairs))]
    assert got == one
    # a second pass over fixed text still parses (the tags are just longer); a single-end record (flag 0 / 16) is not a pair's
    bad = _Pair(b"r", b"r\t0\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII\tAS:i:0\n", b"r\t141\t*\t0\t0\t*\t*\t0\t0\tACGT\tIIII\tAS:i:0\n")
    before = [C.string_at(bad.arr[k].sam) for k in range(2)]
    assert lib.mi355x_fixmate_pair(C.byref(bad.arr[0]), C.byref(bad.arr[1]), ref.bns) == -1
    assert bad.take() == before
    for i in range(2 * len(pairs)):
        pairs[0].libc.free(C.c_void_p(arr[i].sam))


def _bgzf_blocks(data):
    """[(block bytes, payload)] of a BGZF stream, every header field checked (SAM spec 4.1; src/bgzf.c:245-330)"""
    out, at = [], 0
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04" and data[at + 10:at + 12] == b"\x06\x00" and data[at + 12:at + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", data, at + 16)[0] + 1
        blk = data[at:at + bsize]
        payload = zlib.decompress(blk[18:-8], -15)
        crc, isize = struct.unpack("<II", blk[-8:])
        assert crc == zlib.crc32(payload) and isize == len(payload) and bsize <= 65536
        out.append((blk, payload))
        at += bsize
    assert at == len(data)
    return out


def test_bgzf_blocks_hold_whole_records_and_decompress_to_the_text(lib):
    rng = np.random.default_rng(4)
    def record(n):
        return b"r%d\t99\tchr1\t%d\t60\t%dM\t=\t%d\t400\t" % (n, n * 7, 150, n * 7 + 250) + bytes(rng.choice(list(b"ACGT"), 150).tolist()) + b"\t" + \
            bytes((rng.integers(2, 42, 150) + 33).astype(np.uint8).tolist()) + b"\tNM:i:0\tMD:Z:150\tAS:i:150\tXS:i:0\n"
    text = b"".join(record(n) for n in range(4000))                       # 1.5 MB of ordinary records
    long_line = b"x\t4\t*\t0\t0\t*\t*\t0\t0\t" + b"A" * 200000 + b"\t" + b"I" * 200000 + b"\n"   # a record longer than a block
    cases = [text, b"", record(1), text[:70000] + long_line + text[70000:140000], bytes(rng.integers(0, 256, 300000).astype(np.uint8).tolist())]
    for level in (-1, 1, 9, 0):
        for t in cases:
            cap = lib.mi355x_bgzf_bound(len(t))
            out = C.create_string_buffer(cap)
            n = lib.mi355x_bgzf_compress(t, len(t), level, out, cap)
            data = out.raw[:n]
            blocks = _bgzf_blocks(data)
            assert b"".join(p for _, p in blocks) == t
            if t:
                assert gzip.decompress(data) == t                        # a BGZF stream is a multi-member gzip file
            if t is text:
                assert all(p.endswith(b"\n") for _, p in blocks) and all(len(p) > 60000 for _, p in blocks[:-1])
                assert len(data) < 0.7 * len(t) or level == 0   # (random bases and qualities: 2 + 5.3 of 8 bits)
    # the same bytes whatever the number of threads
    t = text
    cap = lib.mi355x_bgzf_bound(len(t))
    outs = []
    for cpus in ("1", "3", None):
        if cpus is None:
            os.environ.pop("MPIBWA_SAMPOST_THREADS", None)
        else:
            os.environ["MPIBWA_SAMPOST_THREADS"] = cpus
        out = C.create_string_buffer(cap)
        outs.append(out.raw[:lib.mi355x_bgzf_compress(t, len(t), 6, out, cap)])
    assert outs[0] == outs[1] == outs[2]
    assert lib.mi355x_bgzf_compress(t, len(t), 6, out, 65536) == 0       # room for one block only
    eof = C.create_string_buffer(28)
    assert lib.mi355x_bgzf_eof(eof) == 28
    assert eof.raw == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    assert _bgzf_blocks(eof.raw) == [(eof.raw, b"")]


def _route_python(text, names, discordant):
    """the routing rule of src/mainParallelByChromosome.c:1340-1455 / :3437-3486, stated on lists of lines"""
    idx = {}
    for i, n in enumerate(names):
        idx.setdefault(n, i)
    n_dest = len(names) + 1 + (1 if discordant else 0)
    dest = [[] for _ in range(n_dest)]
    for ln in text.split(b"\n")[:-1]:
        f = ln.split(b"\t")
        if f[2] == b"*" or f[2] not in idx:
            dest[n_dest - 1].append(ln)
            continue
        c = idx[f[2]]
        dest[c].append(ln)
        if discordant:
            m = c if f[6] == b"=" else idx.get(f[6], -1)
            if m >= 0 and m != c:
                dest[len(names)].append(ln)
    return [b"".join(l + b"\n" for l in d) for d in dest]


def test_route_by_chr_matches_the_rule(genome, lib):
    from mpibwa_amd import abi, simulate
    if not po.ref_available():
        pytest.skip("oracle/_ref/libbwaref.so not built")
    ref = po.RefIndex(genome["prefix"])
    names = [n.encode() for n in genome["names"]]
    reads = simulate.reads_to_ascii(_pairs_of_every_kind(genome, n=200, seed=8))
    pe = b"".join(ref.process(ref.opt(flag=abi.MEM_F_PE), reads))
    se = b"".join(ref.process(ref.opt(), [(n, a, None) for n, a, _ in reads]))
    for text, disc in ((pe, 1), (pe, 0), (se, 0), (b"", 1)):
        n_dest = len(names) + 1 + disc
        outs, lens = (C.c_void_p * n_dest)(), (C.c_size_t * n_dest)()
        n = lib.mi355x_route_by_chr(text, len(text), ref.bns, disc, outs, lens)
        assert n == text.count(b"\n")
        got = []
        for d in range(n_dest):
            got.append(C.string_at(outs[d], lens[d]) if outs[d] else b"")
            assert (outs[d] is None) == (lens[d] == 0)
            if outs[d]:
                po.libc.free(C.c_void_p(outs[d]))
        want = _route_python(text, names, disc)
        assert got == want
        if text is pe and disc:
            assert all(len(g) > 0 for g in got), [len(g) for g in got]   # every contig, discordant and unmapped are exercised
            assert sum(len(g) for g in got) == len(text) + len(got[len(names)])
    assert lib.mi355x_route_by_chr(b"only\tone\n", 9, ref.bns, 0, (C.c_void_p * (len(names) + 1))(), (C.c_size_t * (len(names) + 1))()) == -1


def test_fixmate_on_the_committed_vectors_of_the_reference(lib, tmp_path):
    """tests/golden/fixmate_cases.json.gz: 288 pairs (default, -M, -a) as the reference's mem_process_seqs wrote them on the golden genome
    and what the reference's fixmate() made of each (tools/make_golden_fixmate.py) — needs neither /root/reference nor oracle/_ref."""
    import json
    from golden_util import G, golden_index
    from mpibwa_amd import api
    eng = api.Engine(golden_index(tmp_path), upload=False)
    gold = json.load(gzip.open(os.path.join(G, "fixmate_cases.json.gz"), "rt"))
    assert gold["contigs"] == [eng.bns.contents.anns[i].name.decode() for i in range(eng.bns.contents.n_seqs)]
    kinds = set()
    for c in gold["cases"]:
        pr = _Pair(c["name"].encode(), c["in"][0].encode(), c["in"][1].encode())
        n = lib.mi355x_fixmate_pair(C.byref(pr.arr[0]), C.byref(pr.arr[1]), eng.bns)
        assert n == c["in"][0].count("\n") + c["in"][1].count("\n")
        assert [t.decode() for t in pr.take()] == c["out"], (c["opt"], c["name"])
        for ln in (c["in"][0] + c["in"][1]).splitlines():
            fl = int(ln.split("\t")[1])
            kinds.add("supp" if fl & 0x800 else "sec" if fl & 0x100 else "both_un" if (fl & 12) == 12 else "one_un" if fl & 12 else "other" if ln.split("\t")[6] != "=" else "pair")
    assert kinds == {"supp", "sec", "both_un", "one_un", "other", "pair"}
