"""The library's host chaining (general B-tree path, host_chain.cpp: used for reads with more than 9 chains or 64 seeds and
under MPIBWA_HOST_CHAIN=1) against the reference's own mem_chain + mem_chain_flt (src/bwamem.c:251-385) on adversarial seed
sets; no GPU involved (mi355x_chain_batch with which = 1 is host code of the product)."""
import numpy as np
import pytest

from oracle import pyoracle as po


@pytest.mark.skipif(not po.chain_inject_available(), reason="oracle/_ref/libchaininj.so not present")
def test_host_chaining_matches_the_reference_mem_chain(genome):
    from mpibwa_amd import api
    from chain_cases import adversarial_interval_sets, reference_chains
    eng = api.Engine(genome["prefix"], upload=False)
    ref = po.RefIndex(genome["prefix"])
    l_pac = int(eng.bns.contents.l_pac)
    n_seqs = int(eng.bns.contents.n_seqs)
    offs = [int(eng.bns.contents.anns[k].offset) for k in range(n_seqs)] + [l_pac]
    for kw in (dict(), dict(mask_level=0.3, drop_ratio=0.8, max_chain_gap=200), dict(min_chain_weight=40, w=10)):
        rng = np.random.default_rng(5)
        lens, seedsets, want = reference_chains(ref, ref.opt(**kw), adversarial_interval_sets(rng, 1500, l_pac, offs, n_seqs))
        host = eng.chains(eng.opt(**kw), lens, [0] * len(lens), seedsets, 1)
        for h, w, sd in zip(host, want, seedsets):
            assert [(c[0], c[5], c[6]) for c in h] == w, (kw, sd)
        assert sum(len(w) > 9 for w in want) > 5 and sum(len(s) > 64 for s in seedsets) > 20


@pytest.mark.skipif(not po.chain_inject_available(), reason="oracle/_ref/libchaininj.so not present")
def test_host_chain_filter_on_reads_of_high_copy_repeats(genome):
    """Hundreds of equal-weight chains that overlap completely: mem_chain_flt's kept list grows with every chain (the library
    scans it as columns of ints, host_chain.cpp: chain_filter); max_occ raised so that every hit is a seed, as a user's -c does."""
    from mpibwa_amd import api
    from chain_cases import repeat_like_interval_sets, reference_chains
    eng = api.Engine(genome["prefix"], upload=False)
    ref = po.RefIndex(genome["prefix"])
    l_pac = int(eng.bns.contents.l_pac)
    n_seqs = int(eng.bns.contents.n_seqs)
    offs = [int(eng.bns.contents.anns[k].offset) for k in range(n_seqs)] + [l_pac]
    for kw in (dict(max_occ=1000), dict(max_occ=1000, mask_level=0.3, drop_ratio=0.8), dict(max_occ=1000, min_chain_weight=25, max_chain_extend=50)):
        rng = np.random.default_rng(11)
        lens, seedsets, want = reference_chains(ref, ref.opt(**kw), repeat_like_interval_sets(rng, 40, l_pac, offs, n_seqs))
        host = eng.chains(eng.opt(**kw), lens, [0] * len(lens), seedsets, 1)
        for h, w, sd in zip(host, want, seedsets):
            assert [(c[0], c[5], c[6]) for c in h] == w, kw
        assert max(len(sd) for sd in seedsets) > 500 and max(len(w) for w in want) >= 30   # hundreds of chains in, the cap of max_chain_extend out
