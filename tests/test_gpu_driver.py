"""mpibwa_amd/mpibwa_gpu end to end on the GPU box: `mpiexec -n N mpibwa_gpu mem ...` on the reference's example data.
One chunk (default -K with -t 8): the SAM body hashes to what the real mpiBWA wrote.  Several chunks handed out by the
fetch-and-add counter to 1, 2 and 3 ranks (sharing the box's GPU): the sorted body is the single-process loop's
(fastq.align_files, itself compared chunk by chunk with the compiled reference in test_gpu_examples.py)."""
import gzip
import hashlib
import os
import subprocess
import sys
import tarfile

import pytest

from test_driver import EXE, EX, mpiexec

pytestmark = pytest.mark.gpu
KNOWN_MD5 = "51ce7ba0592d4a199eac49526b6c9d8c"


@pytest.fixture(scope="module")
def example(tmp_path_factory, built):
    d = tmp_path_factory.mktemp("drv")
    with tarfile.open(os.path.join(EX, "hg19.small.tar.gz")) as t:
        t.extractall(d)
    fq = []
    for k in (1, 2):
        dst = str(d / ("R%d.fastq" % k))
        with gzip.open(os.path.join(EX, "HCC1187C_R%d_10K.fastq.gz" % k), "rb") as g, open(dst, "wb") as f:
            f.write(g.read())
        fq.append(dst)
    return str(d), os.path.join(str(d), "hg19.small.fa"), fq


_EXTRA_ENV = {}   # (tests/test_driver.py runs the output-option checks of this file on the CPU: its preload goes here)


def _run(ranks, args, cwd):
    env = dict(os.environ)
    env.pop("LD_LIBRARY_PATH", None)
    env.update(_EXTRA_ENV)
    r = subprocess.run([mpiexec(), "-n", str(ranks), EXE, "mem"] + args, capture_output=True, text=True, timeout=900, env=env, cwd=cwd)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + "\n" + r.stderr[-4000:])
    assert r.returncode == 0


def _body(path):
    return [ln for ln in open(path, "rb").read().splitlines(keepends=True) if not ln.startswith(b"@")]


@pytest.mark.skipif(mpiexec() is None or not os.path.exists(EXE), reason="mpibwa_gpu or mpiexec not present")
def test_driver_reproduces_the_mpibwa_binary_and_is_rank_count_independent(example):
    from mpibwa_amd import abi, api, fastq
    d, prefix, fq = example
    api.load_library().mi355x_finalize()
    out = os.path.join(d, "one.sam")
    _run(2, ["-t", "8", "-o", out, prefix] + fq, d)
    body = _body(out)
    assert len(body) == 20036 and hashlib.md5(b"".join(body)).hexdigest() == KNOWN_MD5
    # several chunks: -K 1000000 -> 9902 + 9902 + 196 reads; any number of ranks gives the same records
    eng = api.Engine(prefix, device=0)
    want, counts = fastq.align_files(eng, eng.opt(flag=abi.MEM_F_PE), fq[0], fq[1], K=1_000_000)
    assert counts == [9902, 9902, 196]
    api.load_library().mi355x_finalize()
    for ranks in (1, 3):
        out = os.path.join(d, "k%d.sam" % ranks)
        _run(ranks, ["-K", "1000000", "-o", out, prefix] + fq, d)
        assert sorted(_body(out)) == sorted(want.splitlines(keepends=True)), ranks
    # single end
    se, _ = fastq.align_files(api.Engine(prefix, device=0), eng.opt(flag=0), fq[0], None, K=400_000)
    api.load_library().mi355x_finalize()
    out = os.path.join(d, "se.sam")
    _run(2, ["-K", "400000", "-o", out, prefix, fq[0]], d)
    assert sorted(_body(out)) == sorted(se.splitlines(keepends=True))


@pytest.mark.skipif(mpiexec() is None or not os.path.exists(EXE), reason="mpibwa_gpu or mpiexec not present")
def test_driver_takes_the_reference_mem_options(example):
    """-R / -M / -T / -k / -H through the driver's own option parser (the reference's letters, src/mainParallel.c:311-398), three
    chunks in flight: same records as the single-process loop with the same mem_opt_t and read group; the header carries the
    @SQ lines, the -H line, the read group and a @PG line."""
    import ctypes as C
    from mpibwa_amd import abi, api, fastq
    d, prefix, fq = example
    lib = api.load_library()
    lib.mi355x_finalize()
    lib.bwa_set_rg.restype = C.c_void_p
    lib.bwa_set_rg.argtypes = [C.c_char_p]
    eng = api.Engine(prefix, device=0)
    try:
        p = lib.bwa_set_rg(b"@RG\\tID:lane7\\tSM:s1")
        assert p
        api.libc.free(C.c_void_p(p))
        want, counts = fastq.align_files(eng, eng.opt(flag=abi.MEM_F_PE | abi.MEM_F_NO_MULTI, T=20, min_seed_len=17), fq[0], fq[1], K=1_000_000)
    finally:
        C.memset((C.c_char * 256).in_dll(lib, "bwa_rg_id"), 0, 256)
    lib.mi355x_finalize()
    out = os.path.join(d, "opts.sam")
    _run(2, ["-K", "1000000", "-M", "-T", "20", "-k", "17", "-R", "@RG\\tID:lane7\\tSM:s1", "-H", "@CO\tmade by the test", "--in-flight", "3", "-o", out,
             prefix] + fq, d)
    assert sorted(_body(out)) == sorted(want.splitlines(keepends=True))
    head = [ln for ln in open(out, "rb").read().splitlines() if ln.startswith(b"@")]
    assert head[0].startswith(b"@SQ\tSN:") and b"@CO\tmade by the test" in head and b"@RG\tID:lane7\tSM:s1" in head
    assert head[-1].startswith(b"@PG\tID:mpibwa_gpu")
    assert all(b"\tRG:Z:lane7" in ln for ln in _body(out))
    # with PREFIX.map next to the index the ranks attach that image (one host copy per node) instead of the five files
    assert lib.mi355x_write_map(prefix.encode(), (prefix + ".map").encode()) == 0
    out2 = os.path.join(d, "opts_map.sam")
    env = dict(os.environ); env.pop("LD_LIBRARY_PATH", None)
    r = subprocess.run([mpiexec(), "-n", "2", EXE, "mem", "-K", "1000000", "-M", "-T", "20", "-k", "17", "-R", "@RG\\tID:lane7\\tSM:s1", "-o", out2, prefix] + fq,
                       capture_output=True, text=True, timeout=900, env=env, cwd=d)
    os.remove(prefix + ".map")
    assert r.returncode == 0 and "index attached from" in r.stderr, r.stderr[-2000:]
    assert sorted(_body(out2)) == sorted(want.splitlines(keepends=True))


@pytest.mark.skipif(mpiexec() is None or not os.path.exists(EXE), reason="mpibwa_gpu or mpiexec not present")
def test_driver_warms_its_call_contexts_beside_the_fastq_scan(example):
    """One rank, the 20 036 reads of the example as one chunk: the driver loads the index first and warms a call context on as many
    sampled reads while it reads the FASTQ offsets (mi355x_prewarm; it says so on stderr; inputs under 20 000 reads per chunk are
    not worth it); --no-prewarm does not; the records are the same either way, and the real mpiBWA's."""
    d, prefix, fq = example
    env = dict(os.environ); env.pop("LD_LIBRARY_PATH", None)
    outs = []
    for extra in ([], ["--no-prewarm"]):
        out = os.path.join(d, "warm%d.sam" % len(extra))
        r = subprocess.run([mpiexec(), "-n", "1", EXE, "mem", "-t", "8", "--in-flight", "8"] + extra + ["-o", out, prefix] + fq,
                           capture_output=True, text=True, timeout=900, env=env, cwd=d)
        assert r.returncode == 0, r.stderr[-3000:]
        assert ("call contexts warmed" in r.stderr) == (not extra), r.stderr[-3000:]
        outs.append(_body(out))
    assert outs[0] == outs[1] and hashlib.md5(b"".join(outs[0])).hexdigest() == KNOWN_MD5


def _bgzf_payload(data):
    """the text of a BGZF stream, block by block, every block's header and trailer checked; returns (text, number of empty blocks at the end)"""
    import struct
    import zlib
    out, at, empty_tail = [], 0, 0
    while at < len(data):
        assert data[at:at + 4] == b"\x1f\x8b\x08\x04" and data[at + 12:at + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", data, at + 16)[0] + 1
        blk = data[at:at + bsize]
        payload = zlib.decompress(blk[18:-8], -15)
        crc, isize = struct.unpack("<II", blk[-8:])
        assert crc == zlib.crc32(payload) and isize == len(payload)
        empty_tail = empty_tail + 1 if not payload else 0
        out.append(payload)
        at += bsize
    assert at == len(data)
    return b"".join(out), empty_tail


def _fixmate_expected(body, idx, fix):
    """The -f pass over a SAM body, pair by pair (the lines of a read stay in file order), by `fix` — the reference's fixmate()
    (oracle/_ref/libfixmateref.so) where that build travelled, else the library's own (pinned to the reference's in test_sampost.py)."""
    import ctypes as C
    from mpibwa_amd import abi, api
    libc = api.libc
    libc.strdup.restype = C.c_void_p
    libc.strdup.argtypes = [C.c_char_p]
    by_name = {}
    for ln in body:
        f = ln.split(b"\t", 2)
        by_name.setdefault(f[0], ([], []))[0 if int(f[1]) & 64 else 1].append(ln)
    out = []
    for name, (l1, l2) in by_name.items():
        nm = C.create_string_buffer(name)
        arr = (abi.bseq1_t * 2)()
        for k, ls in enumerate((l1, l2)):
            arr[k].l_seq = 101
            arr[k].name = C.addressof(nm)
            arr[k].sam = libc.strdup(b"".join(ls))
        fix(arr, idx)
        for k in range(2):
            out += C.string_at(arr[k].sam).splitlines(keepends=True)
            libc.free(C.c_void_p(arr[k].sam))
    return out


@pytest.mark.skipif(mpiexec() is None or not os.path.exists(EXE), reason="mpibwa_gpu or mpiexec not present")
def test_driver_output_options_fixmate_bgzf_and_by_chromosome(example, genome, tmp_path):
    check_output_options(example, genome, tmp_path)


def check_output_options(example, genome, tmp_path):
    """The reference's output options on the example data (src/mainParallel.c:298-299, 395; mpiBWAByChr): -f gives the records the
    reference's own fixmate() makes of the plain run's, -g / -b decompress to the plain file (header included; -b ends with the
    empty block), --by-chr puts every record into the file of its contig, pairs on two contigs into discordant as well and RNAME
    '*' into unmapped."""
    import ctypes as C
    from mpibwa_amd import abi, api
    d, prefix, fq = example
    lib = api.load_library()
    if not _EXTRA_ENV:
        lib.mi355x_finalize()
    plain = os.path.join(d, "plain.sam")
    _run(2, ["-K", "1000000", "-o", plain, prefix] + fq, d)
    body = _body(plain)
    assert len(body) >= 20000   # (three chunks: a supplementary line more or less than the one-chunk run's 20 036)
    head = [ln for ln in open(plain, "rb").read().splitlines(keepends=True) if ln.startswith(b"@")]
    eng = api.Engine(prefix, device=0, upload=False)
    # -f
    ref_fix = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "libfixmateref.so")
    if os.path.exists(ref_fix):
        fx = C.CDLL(ref_fix)
        fx.fixmate.argtypes = [C.c_int, C.POINTER(abi.bseq1_t), C.POINTER(abi.bseq1_t), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(abi.bwaidx_t)]
        def fix(arr, idx):
            a, b = C.c_int(0), C.c_int(0)
            assert fx.fixmate(0, C.byref(arr[0]), C.byref(arr[1]), C.byref(a), C.byref(b), idx) == 0
    else:
        lib.mi355x_fixmate_pair.argtypes = [C.POINTER(abi.bseq1_t), C.POINTER(abi.bseq1_t), C.POINTER(abi.bntseq_t)]
        def fix(arr, idx):
            assert lib.mi355x_fixmate_pair(C.byref(arr[0]), C.byref(arr[1]), idx.contents.bns) > 0
    want_f = _fixmate_expected(body, eng.idx, fix)
    out = os.path.join(d, "fixed.sam")
    _run(2, ["-K", "1000000", "-f", "--in-flight", "3", "-o", out, prefix] + fq, d)
    got_f = _body(out)
    assert sorted(got_f) == sorted(want_f) and got_f != body
    assert sum(b"\tMQ:i:" in ln for ln in got_f) > 1000 and all(b"\tms:i:" in ln for ln in got_f if not int(ln.split(b"\t")[1]) & 0x900)
    # -g and -b: the same text in BGZF blocks
    for flag, ends in (("-g", 0), ("-b", 1)):
        out = os.path.join(d, "z%s.bin" % flag[1])
        _run(2, ["-K", "1000000", flag, "-o", out, prefix] + fq, d)
        data = open(out, "rb").read()
        text, empty_tail = _bgzf_payload(data)
        assert empty_tail == ends and len(data) < 0.6 * os.path.getsize(plain)
        lines = text.splitlines(keepends=True)
        zhead = [ln for ln in lines if ln.startswith(b"@")]
        assert zhead[:-1] == head[:-1] and zhead[-1].startswith(b"@PG\tID:mpibwa_gpu")
        assert sorted(ln for ln in lines if not ln.startswith(b"@")) == sorted(body)
        assert gzip.decompress(data) == text
    # --by-chr, as text and compressed with -f, on the three contigs of the session genome with pairs of every kind (mates on two
    # contigs, unmapped ends, chimeric reads)
    from mpibwa_amd import simulate
    from test_sampost import _pairs_of_every_kind
    reads = _pairs_of_every_kind(genome, n=4000, seed=12)
    d, prefix = str(tmp_path), genome["prefix"]
    fq = [os.path.join(d, "g1.fastq"), os.path.join(d, "g2.fastq")]
    simulate.write_fastq(fq[0], reads, 0)
    simulate.write_fastq(fq[1], reads, 1)
    plain = os.path.join(d, "gplain.sam")
    _run(2, ["-K", "300000", "-o", plain, prefix] + fq, d)
    body = _body(plain)
    head = [ln for ln in open(plain, "rb").read().splitlines(keepends=True) if ln.startswith(b"@")]
    eng = api.Engine(prefix, device=0, upload=False)
    want_f = _fixmate_expected(body, eng.idx, fix)
    names = [eng.bns.contents.anns[i].name for i in range(eng.bns.contents.n_seqs)]
    assert len(names) == 3 and len(body) > 8000
    for extra, ext, disc in ((["-o", os.path.join(d, "bychr", "x.sam")], "sam", True), (["-f", "-b", "-o", os.path.join(d, "bychr_fb", "x.bam")], "bam", False)):
        os.makedirs(os.path.dirname(extra[-1]), exist_ok=True)
        _run(2, ["-K", "300000", "--by-chr"] + extra + [prefix] + fq, d)
        dd = os.path.dirname(extra[-1])
        src = body if disc else want_f
        files = sorted(os.listdir(dd))
        assert files == sorted([n.decode() + "." + ext for n in names] + ["unmapped." + ext] + (["discordant." + ext] if disc else []))
        total = 0
        for fn in files:
            data = open(os.path.join(dd, fn), "rb").read()
            if ext == "bam":
                data, empty_tail = _bgzf_payload(data)
                assert empty_tail == 1
            lines = data.splitlines(keepends=True)
            recs = [ln for ln in lines if not ln.startswith(b"@")]
            who = fn[:-len(ext) - 1].encode()
            if who == b"unmapped":
                assert sorted(recs) == sorted(ln for ln in src if ln.split(b"\t")[2] == b"*")
            elif who == b"discordant":
                assert sorted(recs) == sorted(ln for ln in src if ln.split(b"\t")[2] != b"*" and ln.split(b"\t")[6] not in (b"=", b"*"))
                assert len(recs) > 0
            else:
                assert sorted(recs) == sorted(ln for ln in src if ln.split(b"\t")[2] == who)
            if who not in (b"unmapped", b"discordant") or ext != "sam":
                assert [ln for ln in lines if ln.startswith(b"@SQ")] == [ln for ln in head if ln.startswith(b"@SQ")]
            if who != b"discordant":
                total += len(recs)
        assert total == len(src)
