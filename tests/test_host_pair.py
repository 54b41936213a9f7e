"""The library's HOST path of mem_sam_pe (csrc/host_pair.cpp, host_regs.cpp, host_ksw.cpp: the path of every pair the pairing kernel leaves
to the host — mate rescue, more than eight hits per end, XA, supplementary and secondary lines, -a / -5 / -P runs) against the reference's
own mem_sam_pe, pair by pair, on the CPU: the regions of both ends come from the reference's mem_align1_core, the insert-size statistics
from its mem_pestat, and both sides turn the same regions into SAM text — the reference through mem_sam_pe (src/bwamem_pair.c:250-393),
the library through mi355x_host_sam_pe.  The genome is a third repeats, so that reads carry many regions and the rescue loop runs with
several candidates per end; the reads include pairs on two contigs, unmapped and chimeric mates.  No GPU involved."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libbwaref.so not built")


def _ref_handle():
    """a ctypes handle of our own on the reference's library: the prototypes set here must not change those of oracle/pyoracle.py's handle"""
    import os
    po.ref_lib()
    h = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(po.__file__)), "_ref", "libbwaref.so"))
    h.bwa_fill_scmat.argtypes = [C.c_int, C.c_int, C.c_void_p]
    return h


class _alnreg_v(C.Structure):   # mem_alnreg_v (src/bwamem.h:79)
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


@pytest.fixture(scope="module")
def repeat_genome(tmp_path_factory, built):
    from mpibwa_amd import api, simulate
    d = tmp_path_factory.mktemp("repeat_genome")
    names, seqs = simulate.make_genome(240_000, 3, seed=17, repeat_frac=0.35)
    fa = str(d / "r.fa")
    simulate.write_fasta(fa, names, seqs)
    api.build_index(fa, fa)
    return {"prefix": fa, "names": names, "seqs": seqs}


def _batch(ref, lib_ref, opt, reads):
    """phase 1 of the reference for every read: regions (mem_align1_core), reads as nt4 codes, the chunk's mem_pestat"""
    from mpibwa_amd import abi
    lib_ref.mem_align1_core.restype = _alnreg_v
    lib_ref.mem_align1_core.argtypes = [C.POINTER(abi.mem_opt_t), C.POINTER(abi.bwt_t), C.POINTER(abi.bntseq_t), C.POINTER(C.c_uint8), C.c_int, C.c_char_p, C.c_void_p]
    lib_ref.mem_pestat.restype = None
    lib_ref.mem_pestat.argtypes = [C.POINTER(abi.mem_opt_t), C.c_int64, C.c_int, C.POINTER(_alnreg_v), C.POINTER(abi.mem_pestat_t)]
    n = 2 * len(reads)
    regs = (_alnreg_v * n)()
    seqs = []
    for p, (name, s1, s2) in enumerate(reads):
        for k, sq in enumerate((s1, s2)):
            buf = C.create_string_buffer(sq, len(sq) + 1)          # ASCII in, nt4 codes out (src/bwamem.c:1057-1058)
            regs[2 * p + k] = lib_ref.mem_align1_core(opt, ref.bwt, ref.bns, ref.pac, len(sq), buf, None)
            seqs.append(buf)
    pes = (abi.mem_pestat_t * 4)()
    lib_ref.mem_pestat(opt, ref.bns.contents.l_pac, n, regs, pes)
    return regs, seqs, pes


def _regs_copy(v):
    if not v.n:
        return np.zeros(0, dtype=po.ALNREG_DT)
    return np.ctypeslib.as_array(C.cast(v.a, C.POINTER(C.c_uint8)), shape=(v.n * 88,)).view(po.ALNREG_DT).copy()


CASES = [
    dict(),
    dict(flag_add="MEM_F_NO_MULTI"),
    dict(flag_add="MEM_F_ALL"),
    dict(flag_add="MEM_F_SOFTCLIP"),
    dict(flag_add="MEM_F_PRIMARY5|MEM_F_KEEP_SUPP_MAPQ"),
    dict(flag_add="MEM_F_NO_RESCUE"),
    dict(flag_add="MEM_F_NOPAIRING"),
    dict(max_matesw=3, pen_unpaired=9, max_XA_hits=2, max_XA_hits_alt=4, XA_drop_ratio=0.6),
    dict(a=2, b=5, o_del=8, e_del=2, o_ins=7, e_ins=3, T=50, pen_clip5=8, pen_clip3=6, zdrop=150),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_host_sam_pe_matches_the_reference_mem_sam_pe(repeat_genome, case):
    from mpibwa_amd import abi, api, simulate
    from test_sampost import _pairs_of_every_kind
    lib = api.load_library()
    lib_ref = _ref_handle()
    lib_ref.mem_sam_pe.restype = C.c_int
    lib_ref.mem_sam_pe.argtypes = [C.POINTER(abi.mem_opt_t), C.POINTER(abi.bntseq_t), C.POINTER(C.c_uint8), C.POINTER(abi.mem_pestat_t), C.c_uint64,
                                   C.POINTER(abi.bseq1_t), C.POINTER(_alnreg_v)]
    ref = po.RefIndex(repeat_genome["prefix"])
    kw = dict(CASES[case])
    flag = abi.MEM_F_PE
    for f in kw.pop("flag_add", "").split("|"):
        if f:
            flag |= getattr(abi, f)
    opt = ref.opt(flag=flag, **kw)
    if "a" in kw:
        lib_ref.bwa_fill_scmat(kw["a"], kw["b"], opt.contents.mat)
    reads = simulate.reads_to_ascii(_pairs_of_every_kind(repeat_genome, n=360, seed=40 + case))
    n_rescued, n_many, n_xa, n_supp, n_lines = _compare_pairs(lib, lib_ref, ref, opt, reads, case)
    # the cases are the ones the kernel leaves to the host
    if not flag & (abi.MEM_F_NO_RESCUE | abi.MEM_F_NOPAIRING):
        assert n_rescued > 20, n_rescued
    assert n_many > 20 and n_lines >= 2 * len(reads) and n_xa + n_supp > 40, (n_rescued, n_many, n_xa, n_supp, n_lines)


def _compare_pairs(lib, lib_ref, ref, opt, reads, tag):
    """every pair of `reads` through the reference's mem_sam_pe and the library's host path on the regions of the reference's phase 1; -> counts"""
    from mpibwa_amd import abi, api
    regs, seqs, pes = _batch(ref, lib_ref, opt, reads)
    libc = api.libc
    n_rescued = n_many = n_xa = n_supp = n_lines = 0
    _compare_pairs.n_pa = 0
    for p, (name, _, _) in enumerate(reads):
        nm = C.create_string_buffer(name.encode())
        qual = [C.create_string_buffer(bytes((33 + (7 * i + p) % 40 for i in range(len(seqs[2 * p + k]) - 1)))) for k in range(2)]
        copies = [_regs_copy(regs[2 * p + k]) for k in range(2)]
        texts = []
        for who in ("ref", "own"):
            s = (abi.bseq1_t * 2)()
            for k in range(2):
                s[k].l_seq = len(seqs[2 * p + k]) - 1
                s[k].name = C.addressof(nm)
                s[k].seq = C.addressof(seqs[2 * p + k])
                s[k].qual = C.addressof(qual[k])
            if who == "ref":
                a = (_alnreg_v * 2)(regs[2 * p], regs[2 * p + 1])
                r = lib_ref.mem_sam_pe(opt, ref.bns, ref.pac, pes, p, s, a)
                for k in range(2):   # (mem_sam_pe may have moved the arrays: they are freed through what it left)
                    regs[2 * p + k] = a[k]
            else:
                r = lib.mi355x_host_sam_pe(opt, ref.bns, C.cast(ref.pac, C.c_void_p), C.cast(pes, C.c_void_p), p, s, copies[0].ctypes.data, len(copies[0]),
                                           copies[1].ctypes.data, len(copies[1]))
            t = []
            for k in range(2):
                t.append(C.string_at(s[k].sam))
                libc.free(C.c_void_p(s[k].sam))
            texts.append((r, t))
        assert texts[0] == texts[1], (tag, name, [len(c) for c in copies])
        n_rescued += texts[0][0]
        n_many += len(copies[0]) > 8 or len(copies[1]) > 8
        both = texts[0][1][0] + texts[0][1][1]
        n_xa += both.count(b"\tXA:Z:")
        n_supp += sum(int(ln.split(b"\t")[1]) & 0x900 != 0 for ln in both.splitlines())
        n_lines += both.count(b"\n")
        _compare_pairs.n_pa += both.count(b"\tpa:f:")
    for v in regs:
        libc.free(C.c_void_p(v.a))
    return n_rescued, n_many, n_xa, n_supp, n_lines


@pytest.mark.parametrize("kw", [dict(), dict(flag_add="MEM_F_ALL"), dict(max_XA_hits=2, max_XA_hits_alt=20, XA_drop_ratio=0.5)])
def test_host_sam_pe_with_alt_contigs(genome_alt, kw):
    """The same on a reference with four ALT contigs (diverged copies of primary regions, named in an .alt file as bwa's GRCh38 kit does): reads
    from those regions hit the primary assembly and the ALT contig — the two rounds of mem_mark_primary_se, alt_sc, the pa:f tag, XA entries of
    ALT hits (max_XA_hits_alt), MAPQ caps — text for text."""
    from mpibwa_amd import abi, api, simulate
    lib = api.load_library()
    lib_ref = _ref_handle()
    lib_ref.mem_sam_pe.restype = C.c_int
    lib_ref.mem_sam_pe.argtypes = [C.POINTER(abi.mem_opt_t), C.POINTER(abi.bntseq_t), C.POINTER(C.c_uint8), C.POINTER(abi.mem_pestat_t), C.c_uint64,
                                   C.POINTER(abi.bseq1_t), C.POINTER(_alnreg_v)]
    ref = po.RefIndex(genome_alt["prefix"])
    assert sum(ref.bns.contents.anns[i].is_alt for i in range(ref.bns.contents.n_seqs)) == len(genome_alt["alt"]) > 0
    kw = dict(kw)
    flag = abi.MEM_F_PE
    for f in kw.pop("flag_add", "").split("|"):
        if f:
            flag |= getattr(abi, f)
    opt = ref.opt(flag=flag, **kw)
    # pairs from everywhere, and as many again from the ALT contigs alone (their twins on the primary assembly come along as hits)
    alt_seqs = [s for n, s in zip(genome_alt["names"], genome_alt["seqs"]) if n in genome_alt["alt"]]
    reads = simulate.reads_to_ascii(simulate.simulate_reads(genome_alt["seqs"], 250, 150, paired=True, seed=61) +
                                    [("a" + n, a, b) for n, a, b in simulate.simulate_reads(alt_seqs, 250, 150, paired=True, seed=62, frag_mean=300.0, frag_sd=30.0)])
    n_rescued, n_many, n_xa, n_supp, n_lines = _compare_pairs(lib, lib_ref, ref, opt, reads, str(kw))
    assert n_lines >= 2 * len(reads) and _compare_pairs.n_pa > 50, (n_lines, _compare_pairs.n_pa, n_xa)   # (pa:f: = a primary hit that has an ALT twin)


class _chain_v(C.Structure):   # mem_chain_v (src/bwamem.c:180)
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


CHAIN_T_BYTES = 40   # mem_chain_t (src/bwamem.c:174-179): n, m, first, rid, w:29|kept:2|is_alt:1, frac_rep, pos, seeds*


@pytest.mark.parametrize("kw", [dict(), dict(a=2, b=5, o_del=8, e_del=2, o_ins=7, e_ins=3, T=50, zdrop=150), dict(mask_level_redun=0.8, w=40),
                                dict(flag_add="MEM_F_PRIMARY5|MEM_F_KEEP_SUPP_MAPQ", T=20), dict(flag_add="MEM_F_ALL", max_XA_hits=3)])
def test_host_dedup_patch_and_single_end_records_match_the_reference(repeat_genome, kw):
    """mem_sort_dedup_patch with its patch alignments (src/bwamem.c:406-489) and the single-end half of worker2 (mem_mark_primary_se,
    mem_reorder_primary5, mem_reg2sam with XA / SA / supplementary lines: src/bwamem.c:521-569, 978-1048) of the library's host path against
    the reference's own functions: the regions come from the reference's mem_chain -> mem_chain_flt -> mem_flt_chained_seeds ->
    mem_chain2aln, are copied, and go through both.  Reads of 150-400 bp built from two or three pieces a few bases apart on the genome,
    from chimeric pieces, and — every fifth — from two long pieces a deletion apart that is wider than the band, so that two co-linear regions
    are there for mem_patch_reg to join."""
    from mpibwa_amd import abi, api
    lib = api.load_library()
    R = _ref_handle()
    ref = po.RefIndex(repeat_genome["prefix"])
    kw = dict(kw)
    flag = 0
    for f in kw.pop("flag_add", "").split("|"):
        if f:
            flag |= getattr(abi, f)
    opt = ref.opt(flag=flag, **kw)
    if "a" in kw:
        R.bwa_fill_scmat(kw["a"], kw["b"], opt.contents.mat)
    P_opt, P_bwt, P_bns, P_u8 = C.POINTER(abi.mem_opt_t), C.POINTER(abi.bwt_t), C.POINTER(abi.bntseq_t), C.POINTER(C.c_uint8)
    R.mem_chain.restype = _chain_v
    R.mem_chain.argtypes = [P_opt, P_bwt, P_bns, C.c_int, C.c_char_p, C.c_void_p]
    R.mem_chain_flt.restype = C.c_int
    R.mem_chain_flt.argtypes = [P_opt, C.c_int, C.c_void_p]
    R.mem_flt_chained_seeds.restype = None
    R.mem_flt_chained_seeds.argtypes = [P_opt, P_bns, P_u8, C.c_int, C.c_char_p, C.c_int, C.c_void_p]
    R.mem_chain2aln.restype = None
    R.mem_chain2aln.argtypes = [P_opt, P_bns, P_u8, C.c_int, C.c_char_p, C.c_void_p, C.POINTER(_alnreg_v)]
    R.mem_sort_dedup_patch.restype = C.c_int
    R.mem_sort_dedup_patch.argtypes = [P_opt, P_bns, P_u8, C.c_char_p, C.c_int, C.c_void_p]
    R.mem_mark_primary_se.restype = C.c_int
    R.mem_mark_primary_se.argtypes = [P_opt, C.c_int, C.c_void_p, C.c_int64]
    R.mem_reorder_primary5.restype = None
    R.mem_reorder_primary5.argtypes = [C.c_int, C.POINTER(_alnreg_v)]
    R.mem_reg2sam.restype = None
    R.mem_reg2sam.argtypes = [P_opt, P_bns, P_u8, C.POINTER(abi.bseq1_t), C.POINTER(_alnreg_v), C.c_int, C.c_void_p]
    rng = np.random.default_rng(77 + len(kw))
    seqs = repeat_genome["seqs"]
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    n_patched = n_dropped = n_multi = n_xa = n_sa = 0
    for r in range(400):
        c = int(rng.integers(0, len(seqs)))
        kind = r % 5
        w_opt = int(opt.contents.w)
        if kind == 0:   # two long pieces a deletion apart that neither the chain's band nor an extension crosses, but mem_patch_reg's limits admit
            gap = w_opt + 1 + int(rng.integers(0, w_opt // 2))
            ln = int(gap * (11 + 3 * rng.random()))
            pos = int(rng.integers(0, len(seqs[c]) - 2 * ln - gap - 10))
            parts = [seqs[c][pos:pos + ln].copy(), seqs[c][pos + ln + gap:pos + 2 * ln + gap].copy()]
        elif kind == 1:   # a chimeric read: pieces from anywhere, either strand (supplementary lines, SA tags)
            parts = []
            for _ in range(int(rng.integers(2, 4))):
                cc = int(rng.integers(0, len(seqs)))
                ln = int(rng.integers(60, 140))
                pp = int(rng.integers(0, len(seqs[cc]) - ln))
                piece = seqs[cc][pp:pp + ln].copy()
                parts.append(comp[np.minimum(piece, 4)[::-1]] if rng.random() < 0.5 else piece)
        else:   # two or three pieces of one contig with small gaps between them (deletions of 0-40 bases; sometimes an insertion)
            pos = int(rng.integers(0, len(seqs[c]) - 1500))
            parts = []
            for _ in range(int(rng.integers(1, 4))):
                ln = int(rng.integers(60, 160))
                parts.append(seqs[c][pos:pos + ln].copy())
                pos += ln + int(rng.integers(0, 41))
                if rng.random() < 0.3:
                    parts.append(rng.integers(0, 4, int(rng.integers(1, 12))).astype(np.uint8))
        read = np.concatenate(parts)
        read = np.where(read > 3, 0, read).astype(np.uint8)
        m = rng.random(len(read)) < (0.003 if kind == 0 else 0.01)
        read[m] = (read[m] + 1) & 3
        if rng.random() < 0.5:
            read = comp[read[::-1]]
        ascii_read = lut[read].tobytes()
        buf = C.create_string_buffer(ascii_read, len(read) + 1)
        for i in range(len(read)):   # nt4 codes, as mem_align1_core makes them (src/bwamem.c:1057-1058)
            buf[i] = bytes([int(read[i])])
        chn = R.mem_chain(opt, ref.bwt, ref.bns, len(read), buf, None)
        chn.n = R.mem_chain_flt(opt, chn.n, chn.a)
        R.mem_flt_chained_seeds(opt, ref.bns, ref.pac, len(read), buf, chn.n, chn.a)
        regs = _alnreg_v()
        for i in range(chn.n):
            R.mem_chain2aln(opt, ref.bns, ref.pac, len(read), buf, chn.a + i * CHAIN_T_BYTES, C.byref(regs))
        raw = _regs_copy(regs)
        # mem_sort_dedup_patch
        n_ref = R.mem_sort_dedup_patch(opt, ref.bns, ref.pac, buf, regs.n, regs.a)
        regs.n = n_ref
        want = _regs_copy(regs)
        mine = raw.copy()
        n_own = lib.mi355x_host_sort_dedup_patch(opt, ref.bns, C.cast(ref.pac, C.c_void_p), C.cast(buf, C.c_void_p), mine.ctypes.data, len(mine))
        assert n_own == n_ref, (r, len(raw), n_own, n_ref)
        got = mine[:n_own]
        for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep"):
            assert (got[f] == want[f]).all(), (r, f, got[f], want[f])
        n_dropped += len(raw) - n_ref
        n_patched += int(sum(1 for x in want if not ((raw["rb"] == x["rb"]) & (raw["re"] == x["re"])).any()))
        n_multi += n_ref > 1
        # single end: primary marking with the read's id, -5, the records
        qual = C.create_string_buffer(bytes((33 + (3 * i + r) % 41 for i in range(len(read)))))
        nm = C.create_string_buffer(b"read%d" % r)
        texts = []
        for who in ("ref", "own"):
            s = abi.bseq1_t()
            s.l_seq = len(read); s.name = C.addressof(nm); s.seq = C.addressof(buf); s.qual = C.addressof(qual)
            if who == "ref":
                R.mem_mark_primary_se(opt, regs.n, regs.a, 1000 + r)
                if flag & abi.MEM_F_PRIMARY5:
                    R.mem_reorder_primary5(opt.contents.T, C.byref(regs))
                R.mem_reg2sam(opt, ref.bns, ref.pac, C.byref(s), C.byref(regs), 0, None)
            else:
                lib.mi355x_host_reg2sam_se(opt, ref.bns, C.cast(ref.pac, C.c_void_p), C.byref(s), want.ctypes.data, len(want), 1000 + r)
            texts.append(C.string_at(s.sam))
            api.libc.free(C.c_void_p(s.sam))
        assert texts[0] == texts[1], (r, kw)
        n_xa += texts[0].count(b"\tXA:Z:")
        n_sa += texts[0].count(b"\tSA:Z:")
        api.libc.free(C.c_void_p(regs.a))
    assert n_patched > 15 and n_dropped > 5 and n_multi > 100 and n_sa > 50, (n_patched, n_dropped, n_multi, n_xa, n_sa)
    assert n_xa > 20 or flag & abi.MEM_F_ALL, n_xa   # (-a prints the secondary hits as lines of their own instead of XA)


def test_host_pestat_matches_the_reference_mem_pestat(repeat_genome, capfd):
    """mem_pestat (src/bwamem_pair.c:46-109) of the library's host path — the counting form (max_ins up to 2^20) and the sorting form, on one
    and on several threads — against the reference's own on the regions of its mem_align1_core: libraries with one, two and four live
    orientations (mates reverse-complemented to make FF / RF / RR pairs), too few pairs for any, a wide and a narrow insert-size
    distribution; the four (low, high, failed, avg, std) records bit for bit and the lines on stderr."""
    from mpibwa_amd import abi, api, simulate
    lib = api.load_library()
    R = _ref_handle()
    ref = po.RefIndex(repeat_genome["prefix"])
    comp = {ord("A"): "T", ord("C"): "G", ord("G"): "C", ord("T"): "A", ord("N"): "N"}
    rc = lambda s: s.decode().translate(comp)[::-1].encode()
    l_pac = int(ref.bns.contents.l_pac)
    n_live = []
    verbose = [C.c_int.in_dll(x, "bwa_verbose") for x in (lib, R)]   # (each library has its own; other tests leave them at other levels)
    saved = [v.value for v in verbose]
    for v in verbose:
        v.value = 3
    for case, (n_pairs, sd, mix, kw) in enumerate([(1200, 50.0, (1, 0, 0, 0), {}), (1500, 120.0, (6, 2, 1, 1), {}), (1600, 30.0, (1, 1, 1, 1), dict(max_ins=3_000_000)),
                                                   (12, 50.0, (1, 0, 0, 0), {}), (900, 200.0, (3, 0, 1, 0), dict(max_ins=700))]):
        opt = ref.opt(flag=abi.MEM_F_PE, **kw)
        rng = np.random.default_rng(90 + case)
        base = simulate.reads_to_ascii(simulate.simulate_reads(repeat_genome["seqs"], n_pairs, 150, paired=True, seed=70 + case, frag_sd=sd))
        kinds = rng.choice(4, size=n_pairs, p=np.array(mix) / sum(mix))
        reads = [(n, a, b) if k == 0 else (n, a, rc(b)) if k == 1 else (n, rc(a), rc(b)) if k == 2 else (n, rc(a), b) for (n, a, b), k in zip(base, kinds)]
        capfd.readouterr()
        regs, seqs, want = _batch(ref, R, opt, reads)
        ref_err = capfd.readouterr().err
        flat = [_regs_copy(v) for v in regs]
        n_regs = np.array([len(f) for f in flat], dtype=np.int32)
        allregs = np.concatenate(flat) if len(flat) else np.zeros(0, dtype=po.ALNREG_DT)
        for threads in (1, 4):
            got = (abi.mem_pestat_t * 4)()
            lib.mi355x_host_pestat(opt, l_pac, len(flat), allregs.ctypes.data, n_regs.ctypes.data, C.cast(got, C.c_void_p), threads)
            own_err = capfd.readouterr().err
            assert bytes(got) == bytes(want), (case, threads, [(p.low, p.high, p.failed, p.avg, p.std) for p in got], [(p.low, p.high, p.failed, p.avg, p.std) for p in want])
            assert [l for l in own_err.splitlines() if l.startswith("[M::mem_pestat]")] == [l for l in ref_err.splitlines() if l.startswith("[M::mem_pestat]")]
        n_live.append(sum(1 for p in want if not p.failed))
        for v in regs:
            api.libc.free(C.c_void_p(v.a))
    for v, x in zip(verbose, saved):
        v.value = x
    assert n_live[0] == 1 and n_live[1] >= 3 and n_live[2] == 4 and n_live[3] == 0, n_live


SEED_DT = np.dtype([("rbeg", "<i8"), ("qbeg", "<i4"), ("len", "<i4"), ("score", "<i4"), ("pad", "<i4")])   # mem_seed_t (src/bwamem.c:168-172), 24 bytes
CHAIN_DT = np.dtype([("n", "<i4"), ("m", "<i4"), ("first", "<i4"), ("rid", "<i4"), ("bits", "<u4"), ("frac_rep", "<f4"), ("pos", "<i8"), ("seeds", "<u8")])   # mem_chain_t, 40 bytes


@pytest.mark.parametrize("kw", [dict(), dict(min_chain_weight=30), dict(a=2, b=5, o_del=8, e_del=2, o_ins=7, e_ins=3)])
def test_host_seed_rescoring_of_long_reads_matches_the_reference(repeat_genome, kw):
    """mem_flt_chained_seeds / mem_seed_sw (src/bwamem.c:571-617): for reads of ~700 bp and more every short seed of a chain is rescored by a
    local alignment of its neighbourhood and dropped when the score stays under min_HSP_score.  The chains of the reference's mem_chain +
    mem_chain_flt are copied, go through the reference's function and through the library's host one (mi355x_host_flt_chained_seeds);
    reads of 800-2 000 bp with 4 % errors, so that short seeds in noisy stretches are there to be dropped."""
    from mpibwa_amd import abi, api
    assert SEED_DT.itemsize == 24 and CHAIN_DT.itemsize == CHAIN_T_BYTES
    lib = api.load_library()
    R = _ref_handle()
    ref = po.RefIndex(repeat_genome["prefix"])
    opt = ref.opt(**kw)
    if "a" in kw:
        R.bwa_fill_scmat(kw["a"], kw["b"], opt.contents.mat)
    P_opt, P_bwt, P_bns, P_u8 = C.POINTER(abi.mem_opt_t), C.POINTER(abi.bwt_t), C.POINTER(abi.bntseq_t), C.POINTER(C.c_uint8)
    R.mem_chain.restype = _chain_v
    R.mem_chain.argtypes = [P_opt, P_bwt, P_bns, C.c_int, C.c_char_p, C.c_void_p]
    R.mem_chain_flt.restype = C.c_int
    R.mem_chain_flt.argtypes = [P_opt, C.c_int, C.c_void_p]
    R.mem_flt_chained_seeds.restype = None
    R.mem_flt_chained_seeds.argtypes = [P_opt, P_bns, P_u8, C.c_int, C.c_char_p, C.c_int, C.c_void_p]
    rng = np.random.default_rng(123)
    seqs = repeat_genome["seqs"]
    comp = np.array([3, 2, 1, 0, 4], dtype=np.uint8)
    n_dropped = n_rescored = n_seeds_all = 0

    def chains_of(chn):
        ch = np.ctypeslib.as_array(C.cast(chn.a, C.POINTER(C.c_uint8)), shape=(chn.n * CHAIN_T_BYTES,)).view(CHAIN_DT)
        out = []
        for c in ch:
            sd = np.ctypeslib.as_array(C.cast(int(c["seeds"]), C.POINTER(C.c_uint8)), shape=(int(c["n"]) * 24,)).view(SEED_DT).copy()
            out.append(sd)
        return out
    for r in range(120):
        c = int(rng.integers(0, len(seqs)))
        ln = int(rng.integers(800, 2001))
        pos = int(rng.integers(0, len(seqs[c]) - ln))
        read = np.where(seqs[c][pos:pos + ln] > 3, 0, seqs[c][pos:pos + ln]).astype(np.uint8)
        m = rng.random(ln) < 0.04
        read[m] = (read[m] + 1 + rng.integers(0, 3, int(m.sum()))) & 3
        # behind it: random bases with a few exact 21-32-mers of the genome planted in them — seeds of chains of their own whose
        # neighbourhood does not align: the ones the rescoring is there to drop
        tail = []
        for _ in range(int(rng.integers(1, 5))):
            cc = int(rng.integers(0, len(seqs)))
            k = int(rng.integers(21, 33))
            pp = int(rng.integers(0, len(seqs[cc]) - k))
            tail += [rng.integers(0, 4, int(rng.integers(80, 200))).astype(np.uint8), np.minimum(seqs[cc][pp:pp + k], 3).astype(np.uint8)]
        read = np.concatenate([read] + tail + [rng.integers(0, 4, 100).astype(np.uint8)])
        ln = len(read)
        if r % 2:
            read = comp[read[::-1]]
        buf = C.create_string_buffer(bytes(read.tolist()), ln + 1)
        chn = R.mem_chain(opt, ref.bwt, ref.bns, ln, buf, None)
        chn.n = R.mem_chain_flt(opt, chn.n, chn.a)
        if chn.n == 0:
            continue
        before = chains_of(chn)
        R.mem_flt_chained_seeds(opt, ref.bns, ref.pac, ln, buf, chn.n, chn.a)
        after = chains_of(chn)
        flat = np.concatenate(before)
        n_seeds = np.array([len(b) for b in before], dtype=np.int32)
        lib.mi355x_host_flt_chained_seeds(opt, ref.bns, C.cast(ref.pac, C.c_void_p), ln, C.cast(buf, C.c_void_p), len(before), flat.ctypes.data, n_seeds.ctypes.data)
        at = 0
        for k, (b, a) in enumerate(zip(before, after)):
            got = flat[at:at + n_seeds[k]]
            assert len(got) == len(a), (r, k, len(got), len(a))
            for f in ("rbeg", "qbeg", "len", "score"):
                assert (got[f] == a[f]).all(), (r, k, f)
            at += len(b)
            n_dropped += len(b) - len(a)
            n_rescored += int((a["score"] != a["len"] * opt.contents.a).sum())
            n_seeds_all += len(b)
    assert n_seeds_all > 2000 and n_dropped > 20 and n_rescored > 200, (n_seeds_all, n_dropped, n_rescored)


@pytest.mark.parametrize("which_pes,kw", [(0, {}), (1, {}), (2, {}), (0, dict(pen_unpaired=5, mask_level=0.3, XA_drop_ratio=0.5, max_matesw=4))])
def test_host_sam_pe_on_adversarial_region_lists(repeat_genome, which_pes, kw):
    """The region lists of tests/pair_cases.py (1-9 hits per end: equal scores, contained / shifted / overlapping hits, other strands and contigs, mates
    at every distance; FR, RF and all-four-alive libraries) with random reads under them, so that every record is a low-identity alignment and the
    rescue loop runs on windows that hold nothing: through the reference's mem_sort_dedup_patch, then the reference's mem_sam_pe and the library's host
    path — the same text (the pairing kernel's stage test, tests/test_pair_stage.py, feeds the same lists to the device path)."""
    from mpibwa_amd import abi, api
    from pair_cases import adversarial_pairs
    from test_pair_stage import _pes, PES_SETS
    lib = api.load_library()
    R = _ref_handle()
    ref = po.RefIndex(repeat_genome["prefix"])
    P = C.POINTER
    R.mem_sam_pe.restype = C.c_int
    R.mem_sam_pe.argtypes = [P(abi.mem_opt_t), P(abi.bntseq_t), P(C.c_uint8), P(abi.mem_pestat_t), C.c_uint64, P(abi.bseq1_t), P(_alnreg_v)]
    R.mem_sort_dedup_patch.restype = C.c_int
    R.mem_sort_dedup_patch.argtypes = [P(abi.mem_opt_t), P(abi.bntseq_t), P(C.c_uint8), C.c_char_p, C.c_int, C.c_void_p]
    libc = api.libc
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    l_pac = int(ref.bns.contents.l_pac)
    n_seqs = int(ref.bns.contents.n_seqs)
    offs = np.array([int(ref.bns.contents.anns[k].offset) for k in range(n_seqs)] + [l_pac])
    rng = np.random.default_rng(800 + which_pes + len(kw))
    opt = ref.opt(flag=abi.MEM_F_PE, **kw)
    pes = _pes(PES_SETS[which_pes][0])
    pairs = adversarial_pairs(rng, 1200, l_pac, offs, orient=PES_SETS[which_pes][1])
    n_rescued = n_many = n_dedup = 0
    for k, ends in enumerate(pairs):
        reads = [C.create_string_buffer(bytes(rng.integers(0, 4, 150).astype(np.uint8).tolist()), 151) for _ in range(2)]
        quals = [C.create_string_buffer(b"I" * 150) for _ in range(2)]
        nm = C.create_string_buffer(b"p%d" % k)
        vs = []
        for e in range(2):
            a = np.ascontiguousarray(ends[e], dtype=po.ALNREG_DT)
            p = libc.malloc(max(1, a.nbytes))
            C.memmove(p, a.ctypes.data, a.nbytes)
            v = _alnreg_v(len(a), len(a), p)
            v.n = R.mem_sort_dedup_patch(opt, ref.bns, ref.pac, reads[e], v.n, v.a)
            vs.append(v)
            # (the library's own pass over the same raw list: the same regions in the same order)
            mine = a.copy()
            m = lib.mi355x_host_sort_dedup_patch(opt, ref.bns, C.cast(ref.pac, C.c_void_p), C.cast(reads[e], C.c_void_p), mine.ctypes.data, len(mine))
            want = _regs_copy(v)
            assert m == v.n, (k, e, m, v.n)
            for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov"):
                assert (mine[:m][f] == want[f]).all(), (k, e, f)
            n_dedup += len(a) - m
        copies = [_regs_copy(v) for v in vs]
        texts = []
        for who in ("ref", "own"):
            s = (abi.bseq1_t * 2)()
            for e in range(2):
                s[e].l_seq = 150; s[e].name = C.addressof(nm); s[e].seq = C.addressof(reads[e]); s[e].qual = C.addressof(quals[e])
            if who == "ref":
                a = (_alnreg_v * 2)(vs[0], vs[1])
                r = R.mem_sam_pe(opt, ref.bns, ref.pac, pes, 1000 + k, s, a)
                for e in range(2):
                    libc.free(C.c_void_p(a[e].a))
            else:
                r = lib.mi355x_host_sam_pe(opt, ref.bns, C.cast(ref.pac, C.c_void_p), C.cast(pes, C.c_void_p), 1000 + k, s, copies[0].ctypes.data, len(copies[0]),
                                           copies[1].ctypes.data, len(copies[1]))
            t = [C.string_at(s[e].sam) for e in range(2)]
            for e in range(2):
                libc.free(C.c_void_p(s[e].sam))
            texts.append((r, t))
        assert texts[0] == texts[1], (which_pes, kw, k, [len(c) for c in copies])
        n_rescued += texts[0][0]
        n_many += len(copies[0]) > 4 or len(copies[1]) > 4
    assert n_rescued > 500 and n_many > 100 and n_dedup > 300, (n_rescued, n_many, n_dedup)


@pytest.mark.parametrize("kw", [dict(), dict(flag_add="MEM_F_ALL"), dict(flag_add="MEM_F_PRIMARY5|MEM_F_KEEP_SUPP_MAPQ|MEM_F_SOFTCLIP", T=25),
                                dict(flag_add="MEM_F_NO_MULTI", mask_level=0.3, drop_ratio=0.8, max_XA_hits=2, XA_drop_ratio=0.5)])
def test_host_single_end_records_on_adversarial_region_lists(repeat_genome, kw):
    """The same lists, every end as a single-end read with random bases under it: mem_sort_dedup_patch, mem_mark_primary_se (hash tie-breaks on the
    read's id), -5, mem_reg2sam (XA / SA / supplementary / secondary lines, MAPQ) of the reference and of the library's host path — the same text."""
    from mpibwa_amd import abi, api
    from pair_cases import adversarial_pairs
    lib = api.load_library()
    R = _ref_handle()
    ref = po.RefIndex(repeat_genome["prefix"])
    kw = dict(kw)
    flag = 0
    for f in kw.pop("flag_add", "").split("|"):
        if f:
            flag |= getattr(abi, f)
    opt = ref.opt(flag=flag, **kw)
    P_opt, P_bns, P_u8 = C.POINTER(abi.mem_opt_t), C.POINTER(abi.bntseq_t), C.POINTER(C.c_uint8)
    R.mem_sort_dedup_patch.restype = C.c_int
    R.mem_sort_dedup_patch.argtypes = [P_opt, P_bns, P_u8, C.c_char_p, C.c_int, C.c_void_p]
    R.mem_mark_primary_se.restype = C.c_int
    R.mem_mark_primary_se.argtypes = [P_opt, C.c_int, C.c_void_p, C.c_int64]
    R.mem_reorder_primary5.restype = None
    R.mem_reorder_primary5.argtypes = [C.c_int, C.POINTER(_alnreg_v)]
    R.mem_reg2sam.restype = None
    R.mem_reg2sam.argtypes = [P_opt, P_bns, P_u8, C.POINTER(abi.bseq1_t), C.POINTER(_alnreg_v), C.c_int, C.c_void_p]
    libc = api.libc
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    l_pac = int(ref.bns.contents.l_pac)
    n_seqs = int(ref.bns.contents.n_seqs)
    offs = np.array([int(ref.bns.contents.anns[k].offset) for k in range(n_seqs)] + [l_pac])
    rng = np.random.default_rng(900 + len(kw) + flag)
    n_lines = n_tags = 0
    for k, ends in enumerate(adversarial_pairs(rng, 700, l_pac, offs)):
        for e in range(2):
            read = C.create_string_buffer(bytes(rng.integers(0, 4, 150).astype(np.uint8).tolist()), 151)
            qual = C.create_string_buffer(bytes((33 + (5 * i + k) % 41 for i in range(150))))
            nm = C.create_string_buffer(b"s%d_%d" % (k, e))
            a = np.ascontiguousarray(ends[e], dtype=po.ALNREG_DT)
            p = libc.malloc(max(1, a.nbytes))
            C.memmove(p, a.ctypes.data, a.nbytes)
            v = _alnreg_v(len(a), len(a), p)
            v.n = R.mem_sort_dedup_patch(opt, ref.bns, ref.pac, read, v.n, v.a)
            want = _regs_copy(v)
            texts = []
            for who in ("ref", "own"):
                s = abi.bseq1_t()
                s.l_seq = 150; s.name = C.addressof(nm); s.seq = C.addressof(read); s.qual = C.addressof(qual)
                if who == "ref":
                    R.mem_mark_primary_se(opt, v.n, v.a, 5000 + 2 * k + e)
                    if flag & abi.MEM_F_PRIMARY5:
                        R.mem_reorder_primary5(opt.contents.T, C.byref(v))
                    R.mem_reg2sam(opt, ref.bns, ref.pac, C.byref(s), C.byref(v), 0, None)
                else:
                    lib.mi355x_host_reg2sam_se(opt, ref.bns, C.cast(ref.pac, C.c_void_p), C.byref(s), want.ctypes.data, len(want), 5000 + 2 * k + e)
                texts.append(C.string_at(s.sam))
                libc.free(C.c_void_p(s.sam))
            libc.free(C.c_void_p(v.a))
            assert texts[0] == texts[1], (kw, k, e, len(want))
            n_lines += texts[0].count(b"\n")
            n_tags += texts[0].count(b"\tXA:Z:") + texts[0].count(b"\tSA:Z:")
    assert n_lines >= 1400 and n_tags > 40, (n_lines, n_tags)   # (-a: the secondary hits are lines, not XA entries)
