"""End-to-end parity of mem_process_seqs() (through the C ABI, on the GPU) with
 (a) the committed golden SAM produced by the compiled reference, and
 (b) the compiled reference itself (oracle/_ref/libbwaref.so) on fresh seeded inputs and option variants."""
import ctypes as C

import numpy as np
import pytest

from mpibwa_amd import abi, simulate
from oracle import pyoracle as po
from golden_util import golden_index, load_reads, load_sam, sam_cases, kernel_vectors, ragged

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold_engine(tmp_path_factory, built):
    from mpibwa_amd import api
    prefix = golden_index(tmp_path_factory.mktemp("gold"))
    return api.Engine(prefix, device=0)


def test_golden_sam_all_cases(gold_engine):
    pe, se = load_reads("reads_pe150.tsv.gz"), load_reads("reads_se_var.tsv.gz")
    for case, kw in sam_cases().items():
        reads = pe if case.startswith("pe") else se
        got = b"".join(gold_engine.process(gold_engine.opt(**kw), reads))
        assert got == load_sam(case), case


def test_golden_kernel_vectors_on_gpu(gold_engine):
    kv = kernel_vectors()
    opt = gold_engine.opt()
    reads, outs = ragged(kv, "intv_reads"), ragged(kv, "intv_out")
    got, _, _ = gold_engine.smem(opt, reads, cap=512)
    for g, o in zip(got, outs):
        assert (g.reshape(-1) == o).all()
    sa, _, _ = gold_engine.sa(kv["sa_k"])
    assert (sa == kv["sa_v"]).all()
    qs, ts, p = ragged(kv, "ext_q"), ragged(kv, "ext_t"), kv["ext_p"]
    # zdrop is a batch-wide option: run one batch per distinct value
    for zd in np.unique(p[:, 3]):
        sel = np.nonzero(p[:, 3] == zd)[0]
        o = gold_engine.opt(zdrop=int(zd))
        out, _, _ = gold_engine.extend(o, [qs[i] for i in sel], [ts[i] for i in sel], p[sel, 0], p[sel, 1], p[sel, 2])
        assert (out == kv["ext_o"][sel]).all()


needs_ref = pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libbwaref.so not present")


@pytest.fixture(scope="module")
def both(genome):
    from mpibwa_amd import api
    # the device index is a process-wide singleton: (re)upload this module's genome
    api.load_library().mi355x_finalize()
    return api.Engine(genome["prefix"], device=0), po.RefIndex(genome["prefix"])


def _cmp(eng, ref, reads, kw, **pk):
    want = ref.process(ref.opt(**kw), reads, **pk)
    got = eng.process(eng.opt(**kw), reads, **pk)
    assert len(got) == len(want)
    for i, (a, b) in enumerate(zip(got, want)):
        assert a == b, (i, a[:300], b[:300])


@needs_ref
def test_pe_default_and_chunking(both, reads_pe):
    eng, ref = both
    ra = simulate.reads_to_ascii(reads_pe)
    _cmp(eng, ref, ra, dict(flag=abi.MEM_F_PE))
    # a different chunking changes mem_pestat and therefore the SAM: both sides must agree chunk by chunk
    _cmp(eng, ref, ra[:217], dict(flag=abi.MEM_F_PE))
    _cmp(eng, ref, ra[217:], dict(flag=abi.MEM_F_PE), n_processed=434)


@needs_ref
@pytest.mark.parametrize("kw", [
    dict(flag=abi.MEM_F_PE | abi.MEM_F_NO_MULTI),
    dict(flag=abi.MEM_F_PE | abi.MEM_F_SOFTCLIP),
    dict(flag=abi.MEM_F_PE | abi.MEM_F_ALL),
    dict(flag=abi.MEM_F_PE | abi.MEM_F_PRIMARY5),
    dict(flag=abi.MEM_F_PE | abi.MEM_F_NO_RESCUE),
    dict(flag=abi.MEM_F_PE | abi.MEM_F_NOPAIRING),
    dict(flag=abi.MEM_F_PE | abi.MEM_F_REF_HDR, T=15, min_seed_len=15),
    dict(flag=abi.MEM_F_PE, w=20, zdrop=30, max_occ=50, split_width=3),
    dict(flag=abi.MEM_F_PE, pen_unpaired=5, pen_clip5=0, pen_clip3=9, max_mem_intv=0),
    dict(flag=abi.MEM_F_PE, o_del=4, e_del=2, o_ins=8, e_ins=1, drop_ratio=0.3, mask_level=0.7),
])
def test_pe_option_variants(both, reads_pe, kw):
    eng, ref = both
    _cmp(eng, ref, simulate.reads_to_ascii(reads_pe[:250]), kw)


@needs_ref
def test_se_variable_length(both, reads_var):
    eng, ref = both
    ra = simulate.reads_to_ascii(reads_var)
    _cmp(eng, ref, ra, dict(flag=0))
    _cmp(eng, ref, ra, dict(flag=abi.MEM_F_ALL | abi.MEM_F_SOFTCLIP), n_processed=1000)


@needs_ref
@pytest.mark.parametrize("lens", [(700, 900), (2500, 3500), (5000, 7000)])
def test_long_single_end_reads(both, genome, lens):
    """Reads far beyond the lengths the kernels are sized for: long enough for mem_flt_chained_seeds to act (host chaining), for
    the extension kernel's LDS to need the opt-in above 64 KB, and (beyond ~1 700 bp) for the CIGAR stage to go to the host."""
    eng, ref = both
    reads = simulate.simulate_reads(genome["seqs"], 60, 150, paired=False, seed=5, var_len=lens)
    _cmp(eng, ref, simulate.reads_to_ascii(reads), dict(flag=0))


@needs_ref
def test_scoring_matrix_and_misc_inputs(both, reads_pe):
    eng, ref = both
    ra = simulate.reads_to_ascii(reads_pe[:200])

    def scaled(e, a, b):
        o = e.opt(flag=abi.MEM_F_PE, a=a, b=b)
        e.lib.bwa_fill_scmat(a, b, o.contents.mat)
        return o
    want = ref.process(scaled(ref, 2, 5), ra)
    got = eng.process(scaled(eng, 2, 5), ra)
    assert got == want
    # no qualities, with a comment column (-C), user-supplied insert size (-I)
    _cmp(eng, ref, ra, dict(flag=abi.MEM_F_PE), with_qual=False, comment="BC:Z:ACGT")
    pes = (abi.mem_pestat_t * 4)()
    for d in range(4):
        pes[d].failed = 1
    pes[1].failed = 0
    pes[1].avg, pes[1].std, pes[1].low, pes[1].high = 400.0, 50.0, 100, 700
    _cmp(eng, ref, ra, dict(flag=abi.MEM_F_PE), pes0=pes)
    assert eng.stats()["n_pair_dev"] > 0
    # a degenerate user distribution (-I 400,0): the pair score is infinite / NaN for some distances; the pairs go to the host's arithmetic
    pes[1].std = 0.0
    pes[1].low, pes[1].high = 330, 470
    _cmp(eng, ref, ra, dict(flag=abi.MEM_F_PE), pes0=pes)
    assert eng.stats()["n_pair_dev"] == 0


@needs_ref
def test_edge_reads(both, genome):
    eng, ref = both
    g = genome["seqs"][0]
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)

    def asc(a):
        return lut[np.minimum(a, 4)].tobytes()
    reads = [
        ("short", b"ACGTACGTAC", b"TTGACCA"),                       # shorter than a seed
        ("allN", b"N" * 80, b"N" * 80),
        ("homo", b"A" * 150, b"T" * 150),
        ("edge", asc(g[:150]), asc(simulate._COMP[np.minimum(g[200:350], 3)][::-1])),   # at a contig start
        ("one_unmappable", asc(g[5000:5150]), bytes(np.random.default_rng(4).choice(list(b"ACGT"), 150).tolist())),
        ("tail", asc(g[-150:]), asc(simulate._COMP[np.minimum(g[-400:-250], 3)][::-1])),
    ]
    pes = (abi.mem_pestat_t * 4)()
    for d in range(4):
        pes[d].failed = 1
    pes[1].failed = 0
    pes[1].avg, pes[1].std, pes[1].low, pes[1].high = 300.0, 60.0, 50, 800
    _cmp(eng, ref, reads, dict(flag=abi.MEM_F_PE), pes0=pes)
    _cmp(eng, ref, [(n, a, None) for n, a, b in reads], dict(flag=0))
    # empty batch: returns without touching anything
    assert eng.process(eng.opt(flag=0), []) == []


@needs_ref
def test_overlapped_sub_batches_and_host_cigar_paths(both, reads_pe, monkeypatch):
    """The two-sub-batch overlap (normally only used for big chunks) and the host-side CIGAR path give the same bytes."""
    eng, ref = both
    ra = simulate.reads_to_ascii(reads_pe)
    want = ref.process(ref.opt(flag=abi.MEM_F_PE), ra)
    monkeypatch.setenv("MPIBWA_SUBBATCH_MIN", "100")
    assert eng.process(eng.opt(flag=abi.MEM_F_PE), ra) == want
    monkeypatch.setenv("MPIBWA_SUBBATCH", "1")
    assert eng.process(eng.opt(flag=abi.MEM_F_PE), ra) == want
    assert eng.stats()["n_sam_dev"] > len(want) // 2   # most records of this chunk were written by sam_kernel
    assert eng.stats()["n_pair_dev"] > len(ra) // 2   # most pairs of this chunk were decided by pair_kernel (one plain hit per end)
    monkeypatch.setenv("MPIBWA_HOST_PAIR", "1")     # every pair decided by the host's mem_sam_pe instead
    assert eng.process(eng.opt(flag=abi.MEM_F_PE), ra) == want
    assert eng.stats()["n_pair_dev"] == 0
    monkeypatch.delenv("MPIBWA_HOST_PAIR")
    monkeypatch.setenv("MPIBWA_HOST_SAM", "1")      # every record formatted by the host instead
    assert eng.process(eng.opt(flag=abi.MEM_F_PE), ra) == want
    assert eng.stats()["n_sam_dev"] == 0
    monkeypatch.delenv("MPIBWA_HOST_SAM")
    monkeypatch.setenv("MPIBWA_HOST_CIGAR", "1")
    assert eng.process(eng.opt(flag=abi.MEM_F_PE), ra) == want
    monkeypatch.setenv("MPIBWA_HOST_CHAIN", "1")    # chaining of every read on the host instead of chain_kernel
    assert eng.process(eng.opt(flag=abi.MEM_F_PE), ra) == want
    monkeypatch.delenv("MPIBWA_HOST_CHAIN")
    monkeypatch.setenv("MPIBWA_CHAIN_BIG", "0")     # chain_kernel for reads up to 64 seeds / 9 chains only (the rest goes to the host)
    assert eng.process(eng.opt(flag=abi.MEM_F_PE), ra) == want
    monkeypatch.delenv("MPIBWA_CHAIN_BIG")
    monkeypatch.setenv("MPIBWA_C2A_EARLY", "2")     # every extension also computed row by row like the reference: aborts on a difference
    assert eng.process(eng.opt(flag=abi.MEM_F_PE), ra) == want
    monkeypatch.setenv("MPIBWA_C2A_EARLY", "0")     # no closed form, no early stop
    assert eng.process(eng.opt(flag=abi.MEM_F_PE), ra) == want
    monkeypatch.delenv("MPIBWA_C2A_EARLY")
    monkeypatch.setenv("MPIBWA_HOST_MATESW", "1")   # mate rescue computed by the host's striped SW instead of msw_kernel
    assert eng.process(eng.opt(flag=abi.MEM_F_PE), ra) == want
    assert eng.stats()["n_msw"] == 0
    monkeypatch.delenv("MPIBWA_HOST_MATESW")
    monkeypatch.delenv("MPIBWA_HOST_CIGAR")
    assert eng.process(eng.opt(flag=abi.MEM_F_PE), ra) == want
    assert eng.stats()["n_msw"] > 0


@needs_ref
def test_calls_in_flight_from_several_threads(both, reads_pe, monkeypatch):
    """Several caller threads may be inside mem_process_seqs at once (eight run, further ones wait): different chunks,
    different options, many rounds — every call returns the bytes the reference gives for its chunk."""
    import threading
    eng, ref = both
    ra = simulate.reads_to_ascii(reads_pe)
    monkeypatch.setenv("MPIBWA_SUBBATCH_MIN", "100")   # the sub-batch lanes and the two-part SAM stage as for big chunks
    jobs = [(ra, dict(flag=abi.MEM_F_PE)), (ra[:300], dict(flag=abi.MEM_F_PE)), (ra[100:], dict(flag=abi.MEM_F_PE, T=20)),
            ([(n, a, None) for n, a, _ in ra], dict(flag=0))]
    want = [ref.process(ref.opt(**kw), r) for r, kw in jobs]
    bad = []

    def caller(t):
        for rnd in range(6):
            j = (t + rnd) % len(jobs)
            r, kw = jobs[j]
            got = eng.process(eng.opt(**kw), r)
            st = eng.stats()
            if got != want[j] or st["n_reads"] != len(got):
                bad.append((t, rnd, j))

    th = [threading.Thread(target=caller, args=(t,)) for t in range(10)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not bad, bad


@needs_ref
def test_work_buffers_stand_still_once_a_caller_is_warm(both, reads_pe, monkeypatch):
    """A lone caller runs its chunk in sub-batches, a caller with several calls in flight in one piece, through the same
    grow-only work buffers: once a context has seen a chunk in the bigger of its modes, the same chunk again allocates nothing —
    a reallocation stalls every stream of the device (mi355x_buffer_growths counts them)."""
    import threading
    import time
    eng, ref = both
    ra = simulate.reads_to_ascii(reads_pe)
    monkeypatch.setenv("MPIBWA_SUBBATCH_MIN", "100")   # sub-batches and the two-part SAM stage as for big chunks
    opt = eng.opt(flag=abi.MEM_F_PE)
    want = ref.process(ref.opt(flag=abi.MEM_F_PE), ra)
    time.sleep(2.1)                                    # (an earlier test's burst of calls is forgotten after two seconds)
    for _ in range(2):
        assert eng.process(opt, ra) == want            # lone caller: warm
    g0 = int(eng.lib.mi355x_buffer_growths())
    assert eng.process(opt, ra) == want
    assert int(eng.lib.mi355x_buffer_growths()) == g0
    gate, bad = threading.Barrier(4), []

    def caller(rounds):
        for _ in range(rounds):
            gate.wait(timeout=600)
            if eng.process(opt, ra) != want:
                bad.append(1)

    def burst(rounds):
        th = [threading.Thread(target=caller, args=(rounds,)) for _ in range(4)]
        for x in th:
            x.start()
        for x in th:
            x.join()
    burst(3)                                           # four calls in flight, started together: every context warm in the busy mode
    g1 = int(eng.lib.mi355x_buffer_growths())
    burst(3)
    assert not bad
    assert int(eng.lib.mi355x_buffer_growths()) == g1


@needs_ref
def test_prewarm_is_an_ordinary_caller(both, reads_pe):
    """mi355x_prewarm runs calls on reads sampled from the resident reference and throws their text away: it takes time, leaves
    the per-thread statistics of a real call behind, changes nothing about what later calls return, and single-end / odd counts
    / nothing to do are handled."""
    eng, ref = both
    ra = simulate.reads_to_ascii(reads_pe)
    want = ref.process(ref.opt(flag=abi.MEM_F_PE), ra)
    opt = eng.opt(flag=abi.MEM_F_PE)
    secs = eng.prewarm(opt, 4001, read_len=150, n_calls=3)       # 4000 reads (whole pairs) x 3 calls side by side
    assert secs > 0
    assert eng.stats()["n_reads"] == 4000
    assert eng.process(opt, ra) == want
    assert eng.prewarm(eng.opt(), 1500, read_len=101, n_calls=1) > 0   # single end
    assert eng.stats()["n_reads"] == 1500
    assert eng.prewarm(opt, 0, read_len=150, n_calls=2) == 0
    assert eng.process(opt, ra) == want


@needs_ref
def test_long_indels_take_the_wide_band_paths(both, genome):
    """Reads with insertions / deletions of 8 to 40 bp: their CIGARs need bands beyond the narrow direction matrix (the
    full-size variant of the CIGAR kernel, and the host for the few that outgrow that too).  Same bytes as the reference."""
    eng, ref = both
    g = genome["seqs"][0]
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    rng = np.random.default_rng(77)
    reads = []
    for k in range(240):
        L = int(rng.integers(8, 41))
        p = int(rng.integers(1000, len(g) - 2000))
        cut = int(rng.integers(40, 110))
        if k % 2:      # deletion in the read: skip L reference bases
            r1 = np.concatenate([g[p:p + cut], g[p + cut + L:p + 150 + L]])
        else:          # insertion in the read: L random bases
            r1 = np.concatenate([g[p:p + cut], rng.integers(0, 4, L).astype(g.dtype), g[p + cut:p + 150 - L]])
        m = g[p + 300:p + 450]
        r2 = simulate._COMP[np.minimum(m, 3)][::-1]
        reads.append(("indel%d" % k, lut[np.minimum(r1, 4)].tobytes(), lut[np.minimum(r2, 4)].tobytes()))
    _cmp(eng, ref, reads, dict(flag=abi.MEM_F_PE))
    _cmp(eng, ref, reads, dict(flag=abi.MEM_F_PE, w=200))          # wider extension band: longer gaps survive into the CIGAR
    _cmp(eng, ref, [(n, a, None) for n, a, b in reads], dict(flag=0, o_del=3, o_ins=3))
    assert eng.stats()["n_aln"] > 0


def test_overlapping_seqs_in_flight_abort_with_a_message(genome, tmp_path):
    """Two calls in flight on the same seqs[] would both write its seq[] / sam: the library refuses loudly (own process)."""
    import subprocess
    import sys
    import textwrap
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, threading
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        from mpibwa_amd import abi, api, simulate
        eng = api.Engine(%r, device=0)
        _, seqs = simulate.make_genome(50000, 1, seed=5, n_runs=0)
        import numpy as np
        rd = simulate.reads_to_ascii(simulate.simulate_reads(%s, 3000, 150, paired=True, seed=3))
        batch = abi.SeqBatch(api.libc, rd)
        opt = eng.opt(flag=abi.MEM_F_PE)
        th = [threading.Thread(target=lambda: [eng.process_batch(opt, batch) for _ in range(20)]) for _ in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        print("SURVIVED")
    """) % (root, os.path.join(root, "tests"), genome["prefix"], "__import__('pickle').load(open(%r, 'rb'))" % str(tmp_path / "g.pkl"))
    import pickle
    pickle.dump(genome["seqs"], open(tmp_path / "g.pkl", "wb"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "SURVIVED" not in r.stdout
    assert "another call in flight" in r.stderr


@needs_ref
def test_pe_250bp_config4_shape(both, genome):
    """BASELINE config 4's shape (2x250 bp PE): mate rescue takes its 16-bit flavour (250 * a >= 250, src/bwamem_pair.c:152),
    the CIGAR kernel's direction-matrix budget scales with the read length, the seeding kernel its long-read LDS footprint.
    2 400 pairs, default options and one variant, byte-identical to the reference."""
    eng, ref = both
    rd = simulate.reads_to_ascii(simulate.simulate_reads(genome["seqs"], 2400, 250, paired=True, seed=44, frag_mean=600.0, frag_sd=80.0,
                                                         sub=0.012, indel=0.002))
    _cmp(eng, ref, rd, dict(flag=abi.MEM_F_PE))
    st = eng.stats()
    assert st["n_msw"] > 0 and st["n_aln"] > 0
    _cmp(eng, ref, rd[:1200], dict(flag=abi.MEM_F_PE | abi.MEM_F_NO_MULTI, w=60, T=40, pen_unpaired=9))
    # config 3's shape next to it: single-end reads of 50..300 bp
    se = simulate.reads_to_ascii(simulate.simulate_reads(genome["seqs"], 1500, 150, paired=False, seed=45, var_len=(50, 300)))
    _cmp(eng, ref, se, dict(flag=0))


@needs_ref
def test_seedless_chunk_after_a_mapped_one(both, reads_pe):
    """A chunk in which no read has a seed (all-N reads) right after an ordinary chunk of the same size, from the same caller
    thread (hence the same call context and work buffers): the pairing kernel must see region counts of zero for it — not the
    previous chunk's — and every record must be the reference's unmapped one (flags 77 / 141)."""
    eng, ref = both
    ra = simulate.reads_to_ascii(reads_pe[:300])
    _cmp(eng, ref, ra, dict(flag=abi.MEM_F_PE))
    junk = [(n, b"N" * len(a), b"N" * len(b)) for n, a, b in ra]
    want = ref.process(ref.opt(flag=abi.MEM_F_PE), junk)
    got = eng.process(eng.opt(flag=abi.MEM_F_PE), junk)
    assert got == want
    for rec in got:
        assert rec.split(b"\t")[1] in (b"77", b"141")
    _cmp(eng, ref, ra, dict(flag=abi.MEM_F_PE))
