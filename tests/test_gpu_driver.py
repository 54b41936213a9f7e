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


def _run(ranks, args, cwd):
    env = dict(os.environ)
    env.pop("LD_LIBRARY_PATH", None)
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
