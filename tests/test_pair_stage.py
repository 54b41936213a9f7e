"""pair_simple_kernel (mpibwa_amd/csrc/pair_kernel.hip) against the reference's OWN mem_sam_pe on adversarial region lists: the
reference's bwamem_pair.c compiled in place with its record writers redirected to recorders (oracle/pair_inject.c), so that what it
decided — which hit of each end is reported, flags, MAPQ, whether mem_matesw would align, whether a hit gets an XA entry — can be
compared with the kernel's line descriptors and CIGAR requests.  A pair the kernel takes (status 1) must be one the reference reports
through its paired branch without rescue alignment and without XA, with the same region, flag, MAPQ and sub-optimal score for each
end; pairs the kernel leaves to the host are not judged here (the end-to-end tests cover them), only counted."""
import ctypes as C

import numpy as np
import pytest

from mpibwa_amd import abi
from oracle import pyoracle as po


def _pes(specs):
    pes = (abi.mem_pestat_t * 4)()
    for d in range(4):
        pes[d].failed = 1
    for d, (low, high, avg, std) in specs.items():
        pes[d].failed = 0; pes[d].low = low; pes[d].high = high; pes[d].avg = avg; pes[d].std = std
    return pes


def _infer_bw(l1, l2, score, a, q, r):   # src/bwamem.c:792-800
    if l1 == l2 and l1 * a - score < (q + r - a) << 1:
        return 0
    w = int(float((l1 if l1 < l2 else l2) * a - score - q) / r + 2.)
    return max(w, abs(l1 - l2))


# (with more than one orientation alive mem_matesw always finds one that no mate hit explains and aligns: src/bwamem_pair.c:118-128 —
# the kernel then leaves every pair to the host, which the second case checks; only one-orientation libraries have "plain" pairs)
PES_SETS = [
    ({1: (160, 640, 400.0, 50.0)}, "FR"),                                           # FR only (the usual library)
    ({2: (120, 800, 420.0, 80.0)}, "RF"),                                           # RF only (mate-pair libraries)
    ({0: (50, 900, 400.0, 120.0), 1: (160, 640, 400.0, 50.0), 2: (1, 700, 300.0, 90.0), 3: (100, 800, 420.0, 60.0)}, "FR"),   # all four alive
]


@pytest.mark.skipif(not po.pair_inject_available(), reason="oracle/_ref/libpairinj.so not present")
def test_reference_pairing_through_the_injector(genome):
    """CPU: the injector itself — a plain proper pair comes out of the reference's paired branch with flag 0x43 / 0x83 and MAPQ 60,
    a lone far mate makes mem_matesw align, two hits of one end at one place leave one."""
    from pair_cases import _reg
    ref = po.RefIndex(genome["prefix"])
    opt = ref.opt()
    l_pac = int(ref.bns.contents.l_pac)
    pes = _pes(PES_SETS[0][0])
    r0 = np.array([_reg(1000, 1150, 0, 150, 0, 150)], dtype=po.ALNREG_DT)
    rb = 2 * l_pac - (1000 + 400)
    r1 = np.array([_reg(rb, rb + 150, 0, 150, 0, 150)], dtype=po.ALNREG_DT)
    got = po.ref_pair(opt, ref.bns, ref.pac, pes, 7, 150, r0, r1)
    assert got["paired"] and got["n_align"] == 0 and got["n_xa"] == (0, 0)
    assert [l["flag"] for l in got["lines"]] == [0x43, 0x83] and [l["mapq"] for l in got["lines"]] == [60, 60]
    far = np.array([_reg(50000, 50150, 0, 150, 0, 150)], dtype=po.ALNREG_DT)
    got = po.ref_pair(opt, ref.bns, ref.pac, pes, 7, 150, r0, far)
    assert got["n_align"] > 0
    dup = np.array([_reg(1000, 1150, 0, 150, 0, 150), _reg(1000, 1150, 0, 150, 0, 140)], dtype=po.ALNREG_DT)
    got = po.ref_pair(opt, ref.bns, ref.pac, pes, 7, 150, dup, r1)
    assert got["paired"] and got["lines"][0]["score"] == 150


@pytest.mark.gpu
@pytest.mark.skipif(not po.pair_inject_available(), reason="oracle/_ref/libpairinj.so not present")
@pytest.mark.parametrize("which_pes,kw", [(0, {}), (1, {}), (2, {}), (0, dict(pen_unpaired=5, mask_level=0.3, XA_drop_ratio=0.5)),
                                          (0, dict(a=2, b=3, T=40, o_del=4, e_del=2, mapQ_coef_len=70))])
def test_pair_kernel_matches_the_reference_mem_sam_pe(genome, which_pes, kw):
    from mpibwa_amd import api
    from pair_cases import adversarial_pairs
    eng = api.Engine(genome["prefix"], device=0)
    ref = po.RefIndex(genome["prefix"])
    opt, ropt = eng.opt(**kw), ref.opt(**kw)
    l_pac = int(eng.bns.contents.l_pac)
    n_seqs = int(eng.bns.contents.n_seqs)
    offs = np.array([int(eng.bns.contents.anns[k].offset) for k in range(n_seqs)] + [l_pac])
    rng = np.random.default_rng(500 + which_pes + len(kw))
    pairs = adversarial_pairs(rng, 3000, l_pac, offs, orient=PES_SETS[which_pes][1])
    pes = _pes(PES_SETS[which_pes][0])
    maxreg = int(eng.lib.mi355x_pair_maxreg())
    regs = np.zeros((2 * len(pairs), maxreg), dtype=api.Engine.REG_DT)
    n_regs = np.zeros(2 * len(pairs), dtype=np.int32)
    for k, ends in enumerate(pairs):
        for e in range(2):
            n_regs[2 * k + e] = len(ends[e])
            if len(ends[e]) > maxreg:
                continue      # (the kernel leaves such a pair to the host: only the count matters)
            for f in ("rb", "re", "qb", "qe", "rid", "score", "truesc", "w", "seedcov", "seedlen0", "frac_rep"):
                regs[2 * k + e, :len(ends[e])][f] = ends[e][f]
    id0 = 1234
    status, desc, req = eng.pairs(opt, pes, regs, n_regs, max_len=150, n_processed=2 * id0)
    a, q_del, r_del, q_ins, r_ins, w_opt = opt.contents.a, opt.contents.o_del, opt.contents.e_del, opt.contents.o_ins, opt.contents.e_ins, opt.contents.w
    n_taken = n_ref_plain = n_multi = n_tie = n_many = 0
    for k, ends in enumerate(pairs):
        want = po.ref_pair(ropt, ref.bns, ref.pac, pes, id0 + k, 150, ends[0], ends[1])
        plain = want["paired"] and want["n_align"] == 0 and want["n_xa"] == (0, 0) and want["n_lines"] == 2
        n_ref_plain += plain
        if status[k] != 1:
            continue
        n_taken += 1
        assert plain, (k, "the kernel decided a pair the reference treats otherwise", want, ends)
        n_multi += len(ends[0]) + len(ends[1]) > 2
        n_many += len(ends[0]) > 4 or len(ends[1]) > 4
        n_tie += len(set(ends[0]["score"])) < len(ends[0]) or len(set(ends[1]["score"])) < len(ends[1])
        for e in range(2):
            d, rq, L = desc[2 * k + e], req[2 * k + e], want["lines"][e]
            got = dict(rb=int(d["rb"]), re=int(d["re"]), qb=int(d["qb"]), qe=int(d["qe"]), score=int(d["score"]), sub=int(d["sub"]), flag=int(d["flag"]), mapq=int(d["mapq"]))
            exp = {f: L[f] for f in got}
            assert got == exp, (k, e, got, exp, ends)
            assert d["req"] == e and (int(rq["rb"]), int(rq["re"]), int(rq["qb"]), int(rq["qe"]), int(rq["truesc"])) == (L["rb"], L["re"], L["qb"], L["qe"], L["truesc"])
            # the band mem_reg2aln starts its global alignment with (src/bwamem.c:1094-1099)
            l1, l2 = L["qe"] - L["qb"], L["re"] - L["rb"]
            w2 = max(_infer_bw(l1, l2, L["truesc"], a, q_del, r_del), _infer_bw(l1, l2, L["truesc"], a, q_ins, r_ins))
            if w2 > w_opt:
                w2 = min(w2, L["w"])
            assert int(rq["w2"]) == w2, (k, e, int(rq["w2"]), w2)
    # the kernel takes most of what it could take, and the cases are not the easy ones only
    if which_pes == 2:
        assert n_taken == 0 and n_ref_plain == 0
    else:
        assert n_taken > 0.5 * n_ref_plain and n_taken > 100 and n_multi > 60 and n_tie > 20 and n_many > 10, (n_taken, n_ref_plain, n_multi, n_tie, n_many)
