"""Adversarial seed sets for the chaining stage and what the reference's own mem_chain + mem_chain_flt make of them
(oracle/chain_inject.c).  Shared by the CPU test of the library's host chaining and the GPU test of chain_kernel."""
import numpy as np

from oracle import pyoracle as po


def adversarial_interval_sets(rng, n_cases, l_pac, offs, n_seqs):
    """Seed sets as mem_chain sees them: intervals with distinct (qbeg, qend), each with its suffix-array hits."""
    cases = []
    for it in range(n_cases):
        lq = int(rng.choice([100, 150, 151, 250]))
        mode = rng.random()
        n_anchor = int(rng.integers(1, 4)) if mode < 0.6 else int(rng.integers(3, 14))
        if mode > 0.96:
            n_anchor = int(rng.integers(15, 90))            # dozens of chains: the ordered map grows past one node, then past two levels
        anchors = []
        for _ in range(n_anchor):
            k = int(rng.integers(0, n_seqs))
            p = int(rng.integers(offs[k], max(offs[k] + 1, offs[k + 1] - lq - 1)))
            if rng.random() < 0.5:
                p = max(2 * l_pac - 1 - p - lq, l_pac)      # reverse strand
            anchors.append(p)
        if rng.random() < 0.15 and len(anchors) > 1:
            anchors[1] = anchors[0]                          # two chains anchored at the same position
        if rng.random() < 0.05 and n_seqs > 1:
            anchors[0] = offs[int(rng.integers(1, n_seqs))] - 10   # seeds bridging two contigs
        n_iv = int(rng.integers(1, 16)) if mode < 0.9 else int(rng.integers(30, 50))
        if mode > 0.96:
            n_iv = int(rng.integers(60, 110))
        ivs = {}
        for _ in range(n_iv):
            qb = int(rng.integers(0, lq - 19))
            ln = int(rng.integers(19, min(lq - qb, 80) + 1))
            hits = []
            for _ in range(int(rng.choice([1, 1, 1, 2, 3]))):
                a = anchors[int(rng.integers(0, len(anchors)))]
                shift = int(rng.choice([0, 0, 0, 1, -1, 3, 120, 20000]))
                hits.append(min(max(a + qb + shift, 0), 2 * l_pac - ln - 1))
            ivs[(qb, qb + ln)] = hits
        if rng.random() < 0.3 and mode <= 0.96:              # equal weights: disjoint seeds of the same length
            ivs = {((30 * j) % (lq - 25), (30 * j) % (lq - 25) + 25): [min(max(anchors[j % len(anchors)] + 30 * j + (0 if j % 2 else 7), 0), 2 * l_pac - 26)]
                   for j in range(int(rng.integers(3, 9)))}
        cases.append((lq, [(qb, qe, h) for (qb, qe), h in ivs.items()]))
    return cases




def repeat_like_interval_sets(rng, n_cases, l_pac, offs, n_seqs, n_copies=(100, 400), n_ivs=(2, 4)):
    """Reads of a high-copy repeat: two or three intervals with 100 to 400 hits each at unrelated positions, so mem_chain_flt
    sees hundreds of chains of (nearly) equal weight that overlap completely on the query and mostly all survive — the quadratic
    case of its kept-list scan."""
    cases = []
    for it in range(n_cases):
        lq = int(rng.choice([100, 150, 250]))
        n_copy = int(rng.integers(n_copies[0], n_copies[1]))
        copies = []
        for _ in range(n_copy):
            k = int(rng.integers(0, n_seqs))
            p = int(rng.integers(offs[k], max(offs[k] + 1, offs[k + 1] - lq - 1)))
            if rng.random() < 0.5:
                p = max(2 * l_pac - 1 - p - lq, l_pac)
            copies.append(p)
        ivs = {}
        for j in range(int(rng.integers(n_ivs[0], n_ivs[1]))):
            qb = int(rng.integers(0, lq - 40))
            ln = int(rng.integers(19, min(lq - qb, 60) + 1))
            # most copies carry the interval at its place, some a little off (a gap in the copy), some not at all
            hits = sorted(min(max(c + qb + int(rng.choice([0, 0, 0, 0, 2, -3])), 0), 2 * l_pac - ln - 1) for c in copies if rng.random() < 0.9)
            ivs[(qb, qb + ln)] = hits
        cases.append((lq, [(qb, qe, h) for (qb, qe), h in ivs.items()]))
    return cases


def reference_chains(ref, ropt, cases):
    """-> (read lengths, seeds per read in mem_chain's visiting order, expected chains [(rid, frac_rep bits, seeds)])"""
    lens, seedsets, want = [], [], []
    for lq, ivs in cases:
        # mem_chain visits the intervals in info order (mem_collect_intv sorts them, src/bwamem.c:161) and the hits of one
        # interval in suffix-array order (:273-283); every interval here has at most max_occ hits, so l_rep = 0
        ivs = sorted(ivs, key=lambda t: (t[0] << 32) | t[1])
        seedsets.append([(rb, qb, qe - qb) for qb, qe, hits in ivs for rb in hits])
        lens.append(lq)
        exp = []
        for rid, w, kept, is_alt, fb, sd in po.ref_chains(ropt, ref.bns, lq, ivs):
            # mem_chain2aln sorts the seeds of a chain by (score << 32 | index) and walks that array from its end
            # (src/bwamem.c:663-668); the stage hook returns the sorted array
            order = sorted(range(len(sd)), key=lambda i: (sd[i][2] << 32) | i)
            exp.append((rid, fb, [sd[i] for i in order]))
        want.append(exp)
    return lens, seedsets, want
