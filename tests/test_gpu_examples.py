"""The reference's own example data (examples/data, kept as fixtures under tests/golden/mpibwa_examples) through the caller-side
row and the hot path: the SAM body must hash to what the real `mpiBWA mem` binary wrote (md5 of all non-@ lines for
HCC1187C_R{1,2}_10K vs hg19.small with default options, identical for -n 1/2 and -t 1/8 because everything fits one chunk)."""
import hashlib
import os
import tarfile

import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mpibwa_examples")
KNOWN_MD5 = "51ce7ba0592d4a199eac49526b6c9d8c"   # 20 036 records


@pytest.fixture(scope="module")
def hg19_small(tmp_path_factory):
    d = tmp_path_factory.mktemp("hg19small")
    with tarfile.open(os.path.join(HERE, "hg19.small.tar.gz")) as t:
        t.extractall(d)
    return os.path.join(d, "hg19.small.fa")   # the reference's own bwa index files sit next to it


@pytest.fixture(scope="module")
def engine(hg19_small):
    from mpibwa_amd import api
    return api.Engine(hg19_small, device=0)


def test_example_run_matches_mpibwa_binary_output(engine):
    from mpibwa_amd import abi, fastq
    r1, r2 = (os.path.join(HERE, "HCC1187C_R%d_10K.fastq.gz" % k) for k in (1, 2))
    body, counts = fastq.align_files(engine, engine.opt(flag=abi.MEM_F_PE), r1, r2)
    assert counts == [20000]
    assert body.count(b"\n") == 20036
    assert hashlib.md5(body).hexdigest() == KNOWN_MD5


@pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libbwaref.so not built")
def test_example_multi_chunk_matches_reference_per_chunk(engine, hg19_small):
    """-K 1000000: 101-bp reads -> chunks close after 4 951 pairs (500 051 bases per file > K/2); every chunk has its own
    mem_pestat, so each one is compared with the compiled reference run on exactly that chunk."""
    import ctypes as C
    import numpy as np
    from mpibwa_amd import abi, api, fastq
    r1, r2 = (os.path.join(HERE, "HCC1187C_R%d_10K.fastq.gz" % k) for k in (1, 2))
    body, counts = fastq.align_files(engine, engine.opt(flag=abi.MEM_F_PE), r1, r2, K=1_000_000)
    assert counts == [9902, 9902, 196]
    ref = po.RefIndex(hg19_small)
    src = fastq.FastqSource(api.load_library(), r1, r2, K=1_000_000)
    want = []
    for c in range(src.n_chunks):
        rec, n = src.chunk(c)
        ref.lib.mem_process_seqs(ref.opt(flag=abi.MEM_F_PE, n_threads=8), ref.bwt, ref.bns, ref.pac, 0, n,
                                 C.cast(rec.ctypes.data, C.POINTER(abi.bseq1_t)), None)
        for i in range(n):
            p = int(rec[i]["sam"])
            want.append(C.string_at(p))
            po.libc.free(C.c_void_p(p))
    assert body == b"".join(want)
    assert hashlib.md5(body).hexdigest() != KNOWN_MD5   # per-chunk insert-size statistics: -K changes the output


def test_example_chunks_spread_over_ranks_give_the_same_records(engine):
    """Chunks handed to 3 ranks (one after the other on this GPU): the union of their SAM records is the single-rank output —
    chunk boundaries, hence mem_pestat and every record, do not depend on the number of ranks."""
    from mpibwa_amd import abi, fastq
    r1, r2 = (os.path.join(HERE, "HCC1187C_R%d_10K.fastq.gz" % k) for k in (1, 2))
    opt = engine.opt(flag=abi.MEM_F_PE)
    whole, counts = fastq.align_files(engine, opt, r1, r2, K=200_000)
    assert len(counts) >= 8
    parts = [fastq.align_files(engine, opt, r1, r2, K=200_000, rank=r, world=3) for r in range(3)]
    assert sum(len(c) for _, c in parts) == len(counts)
    assert sorted(whole.splitlines()) == sorted(b"".join(b for b, _ in parts).splitlines())
