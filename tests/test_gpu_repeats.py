"""End-to-end parity on a repeat-rich reference (SURVEY §8d's generator, half of the genome in families of up to thousands of copies):
the reads that take the library's heavy paths — hundreds of seeds and chains per read (chaining beyond one B-tree node and beyond
255 seeds, src/bwamem.c:251-385), dozens of regions per end, mem_sam_pe's rescue loop with its redundancy pass per attempted
alignment (src/bwamem_pair.c:250-276, :176), pairing tables of hundreds of entries (:182-243), XA (src/bwamem_extra.c:98-140) —
against the compiled reference, byte for byte."""
import ctypes as C

import numpy as np
import pytest

from mpibwa_amd import abi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu
needs_ref = pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libbwaref.so not present")


@pytest.fixture(scope="module")
def rep(tmp_path_factory, built):
    from mpibwa_amd import api, bigindex
    lib = api.load_library()
    lib.mi355x_finalize()
    lib.mi355x_index_build_gpu.restype = C.c_int
    lib.mi355x_index_build_gpu.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_char_p, C.POINTER(C.c_double)]
    pac, lens = bigindex.synth_packed_genome_grch38like(6e6, seed=17, n_contigs=3, repeat_frac=0.5)
    prefix = str(tmp_path_factory.mktemp("rep") / "rep.fa")
    bigindex.write_meta_files(prefix, pac, lens)
    secs = C.c_double(0)
    assert lib.mi355x_index_build_gpu(0, pac.ctypes.data, int(lens.sum()), prefix.encode(), C.byref(secs)) == 0
    eng = api.Engine(prefix, device=0)
    return bigindex.BigIndex(prefix, pac, lens, eng), po.RefIndex(prefix)


def _cmp(eng, ref, reads, kw, **pk):
    want = ref.process(ref.opt(**kw), reads, **pk)
    got = eng.process(eng.opt(**kw), reads, **pk)
    assert len(got) == len(want)
    bad = [i for i, (a, b) in enumerate(zip(got, want)) if a != b]
    assert not bad, (len(bad), bad[:5], got[bad[0]][:400], want[bad[0]][:400])
    return got


@needs_ref
@pytest.mark.parametrize("seed,n_pairs,kw", [
    (101, 3000, dict(flag=abi.MEM_F_PE)),
    (102, 1500, dict(flag=abi.MEM_F_PE, max_occ=100, max_matesw=10)),
    (103, 1500, dict(flag=abi.MEM_F_PE | abi.MEM_F_NO_MULTI, pen_unpaired=9, XA_drop_ratio=0.5)),
    (104, 1200, dict(flag=abi.MEM_F_PE | abi.MEM_F_ALL)),
])
def test_repeat_rich_pairs(rep, seed, n_pairs, kw):
    idx, ref = rep
    reads = idx.simulate_pairs(n_pairs, seed=seed, read_len=150)
    got = _cmp(idx.engine, ref, reads, kw)
    st = idx.engine.stats()
    # the workload is what it claims to be: many seeds per read, many rescue alignments, XA tags
    assert st["n_seeds"] > 40 * st["n_reads"], st
    if not (kw["flag"] & abi.MEM_F_ALL):
        assert sum(b"\tXA:Z:" in s for s in got) > 20
    assert st["n_msw"] > 0.2 * st["n_reads"], st


@needs_ref
def test_repeat_rich_single_end_and_host_paths(rep, monkeypatch):
    idx, ref = rep
    pairs = idx.simulate_pairs(1200, seed=105, read_len=150)
    se = [(n, a, None) for n, a, _ in pairs]
    _cmp(idx.engine, ref, se, dict(flag=0))
    # the same pairs with the pairing decisions, the chaining and the rescue alignments on the library's host paths
    for var in ("MPIBWA_HOST_PAIR", "MPIBWA_HOST_CHAIN", "MPIBWA_HOST_MATESW"):
        monkeypatch.setenv(var, "1")
        _cmp(idx.engine, ref, pairs[:600], dict(flag=abi.MEM_F_PE))
        monkeypatch.delenv(var)
