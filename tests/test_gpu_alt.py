"""ALT contigs (`.alt` file next to the index, src/bntseq.c:179-204), read groups (-R, src/bwa.c:431-476) and -j / -C on the
device path, against the compiled reference: the two-round primary marking (src/bwamem.c:521-569), alnreg_hlt2's order, the
chain filter's ALT exception (:351), ALT in the redundancy count (:397-400), mem_approx_mapq_se / pairing with alt_sc
(:983-987, src/bwamem_pair.c:347), XA with max_XA_hits_alt (src/bwamem_extra.c:110) and the pa:f tag (src/bwamem.c:1083)."""
import ctypes as C

import numpy as np
import pytest

from mpibwa_amd import abi, simulate
from oracle import pyoracle as po

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libbwaref.so not present")]


@pytest.fixture(scope="module")
def both_alt(genome_alt):
    from mpibwa_amd import api
    api.load_library().mi355x_finalize()
    eng, ref = api.Engine(genome_alt["prefix"], device=0), po.RefIndex(genome_alt["prefix"])
    flags = [[int(x.bns.contents.anns[i].is_alt) for i in range(x.bns.contents.n_seqs)] for x in (eng, ref)]
    assert flags[0] == flags[1] and sum(flags[0]) == len(genome_alt["alt"])
    return eng, ref


@pytest.fixture(scope="module")
def reads_alt(genome_alt):
    """Pairs drawn from the whole genome plus a dense set from the ALT contigs and the primary regions they copy."""
    seqs = genome_alt["seqs"]
    n_pri = len(seqs) - len(genome_alt["alt"])
    rd = simulate.simulate_reads(seqs, 500, 150, paired=True, seed=31)
    rd += [("a" + n, a, b) for n, a, b in simulate.simulate_reads(seqs[n_pri:], 900, 150, paired=True, seed=32, frac_random=0.0)]
    return simulate.reads_to_ascii(rd)


def _cmp(eng, ref, reads, kw, **pk):
    want = ref.process(ref.opt(**kw), reads, **pk)
    got = eng.process(eng.opt(**kw), reads, **pk)
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert a == b, (i, a[:400], b[:400])
    return want


def test_alt_pe_default_has_pa_and_xa(both_alt, reads_alt):
    eng, ref = both_alt
    want = b"".join(_cmp(eng, ref, reads_alt, dict(flag=abi.MEM_F_PE)))
    assert b"\tpa:f:" in want and b"\tXA:Z:" in want and b"_alt" in want     # the compared SAM does exercise the ALT paths
    assert eng.stats()["n_sam_dev"] > 0


@pytest.mark.parametrize("kw", [
    dict(flag=abi.MEM_F_PE | abi.MEM_F_ALL),
    dict(flag=abi.MEM_F_PE | abi.MEM_F_NO_MULTI),
    dict(flag=abi.MEM_F_PE | abi.MEM_F_PRIMARY5 | abi.MEM_F_KEEP_SUPP_MAPQ),
    dict(flag=abi.MEM_F_PE | abi.MEM_F_SOFTCLIP, max_XA_hits=1, max_XA_hits_alt=3),
    dict(flag=abi.MEM_F_PE | abi.MEM_F_NO_RESCUE, T=20, drop_ratio=0.3),
    dict(flag=abi.MEM_F_PE | abi.MEM_F_NOPAIRING, XA_drop_ratio=0.5),
])
def test_alt_pe_option_variants(both_alt, reads_alt, kw):
    eng, ref = both_alt
    _cmp(eng, ref, reads_alt[:900], kw)


def test_alt_single_end(both_alt, reads_alt):
    eng, ref = both_alt
    se = [(n, a, None) for n, a, _ in reads_alt] + [(n + "m", b, None) for n, a, b in reads_alt[500:900]]
    want = b"".join(_cmp(eng, ref, se, dict(flag=0)))
    assert b"\tpa:f:" in want
    _cmp(eng, ref, se[:600], dict(flag=abi.MEM_F_ALL))


def test_ignore_alt_like_option_j(both_alt, reads_alt):
    """-j (src/mainParallel.c:316, src/parallel_aux.c:1831): the caller clears is_alt in its bntseq_t before aligning."""
    eng, ref = both_alt
    saved = []
    for x in (eng, ref):
        anns = x.bns.contents.anns
        saved.append([int(anns[i].is_alt) for i in range(x.bns.contents.n_seqs)])
        for i in range(x.bns.contents.n_seqs):
            anns[i].is_alt = 0
    try:
        want = b"".join(_cmp(eng, ref, reads_alt, dict(flag=abi.MEM_F_PE)))
        assert b"\tpa:f:" not in want
    finally:
        for x, s in zip((eng, ref), saved):
            for i, v in enumerate(s):
                x.bns.contents.anns[i].is_alt = v
    _cmp(eng, ref, reads_alt[:300], dict(flag=abi.MEM_F_PE))      # and back


def test_read_group_and_comment_from_the_device(both_alt, reads_alt):
    """bwa_set_rg on both libraries: RG:Z: reaches every record, also the ones sam_emit_kernel writes; -C appends the comment."""
    eng, ref = both_alt
    libs = (eng.lib, ref.lib)
    for lib in libs:
        lib.bwa_set_rg.restype = C.c_void_p
        lib.bwa_set_rg.argtypes = [C.c_char_p]
    try:
        for lib in libs:
            p = lib.bwa_set_rg(b"@RG\\tID:grp.7\\tSM:y\\tPL:illumina")
            assert p
            po.libc.free(C.c_void_p(p))
        want = _cmp(eng, ref, reads_alt, dict(flag=abi.MEM_F_PE))
        assert all(b"\tRG:Z:grp.7" in line for s in want for line in s.splitlines())
        assert eng.stats()["n_sam_dev"] > len(want) // 5          # the device formatter wrote a good share of what was compared (reads with ALT hits carry XA / extra lines: host)
        _cmp(eng, ref, reads_alt[:400], dict(flag=abi.MEM_F_PE), comment="BC:Z:ACGT+TTAG")
        _cmp(eng, ref, [(n, a, None) for n, a, _ in reads_alt[:400]], dict(flag=0))
    finally:
        for lib in libs:
            C.memset((C.c_char * 256).in_dll(lib, "bwa_rg_id"), 0, 256)
    _cmp(eng, ref, reads_alt[:200], dict(flag=abi.MEM_F_PE))


def test_alt_through_the_gpu_index_builder_and_map(both_alt, genome_alt, reads_alt, tmp_path):
    """Same genome indexed by the GPU builder and attached from a `.map` image: the ALT flags travel (src/bwa.c:347-386)."""
    from mpibwa_amd import api
    eng, ref = both_alt
    want = ref.process(ref.opt(flag=abi.MEM_F_PE), reads_alt[:500])
    import shutil
    pre = str(tmp_path / "gb.fa")
    shutil.copy(genome_alt["prefix"], pre)
    shutil.copy(genome_alt["prefix"] + ".alt", pre + ".alt")
    lib = api.load_library()
    lib.mi355x_finalize()
    # CPU builder writes all five files; the GPU builder then rebuilds .bwt / .sa from the same packed text
    api.build_index(pre, pre)
    cpu = {e: open(pre + "." + e, "rb").read() for e in ("bwt", "sa")}
    l_pac = int(sum(len(s) for s in genome_alt["seqs"]))
    pac = np.fromfile(pre + ".pac", dtype=np.uint8)[:(l_pac + 3) // 4].copy()
    secs = C.c_double(0)
    assert lib.mi355x_index_build_gpu(0, pac.ctypes.data, l_pac, pre.encode(), C.byref(secs)) == 0
    assert all(open(pre + "." + e, "rb").read() == cpu[e] for e in cpu)
    mp = str(tmp_path / "gb.map")
    assert lib.mi355x_write_map(pre.encode(), mp.encode()) == 0
    e2 = api.Engine(pre, device=0, map_path=mp)
    assert [int(e2.bns.contents.anns[i].is_alt) for i in range(e2.bns.contents.n_seqs)][-len(genome_alt["alt"]):] == [1] * len(genome_alt["alt"])
    assert e2.process(e2.opt(flag=abi.MEM_F_PE), reads_alt[:500]) == want
    lib.mi355x_finalize()
