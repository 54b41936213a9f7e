"""Adversarial region lists for the pairing stage: what mem_chain2aln could leave for the two ends of a pair (1-4 regions per end),
built around a true fragment (1-9 regions per end) and bent in the ways that steer mem_sam_pe's decisions — overlapping and contained hits (the redundancy
pass), equal scores (hash tie-breaks of the primary marking and of mem_pair), hits on the other strand / another contig / an ALT contig,
mates at every distance and orientation, short and clipped hits."""
import numpy as np

from oracle import pyoracle as po


def _reg(rb, re, qb, qe, rid, score, truesc=None, w=100, seedcov=None, seedlen0=19, frac_rep=0.0):
    r = np.zeros(1, dtype=po.ALNREG_DT)[0]
    r["rb"], r["re"], r["qb"], r["qe"], r["rid"], r["score"] = rb, re, qb, qe, rid, score
    r["truesc"] = score if truesc is None else truesc
    r["w"] = w
    r["seedcov"] = (qe - qb) // 2 if seedcov is None else seedcov
    r["seedlen0"] = seedlen0
    r["frac_rep"] = frac_rep
    r["secondary"] = r["secondary_all"] = -1
    return r


def adversarial_pairs(rng, n_pairs, l_pac, offs, lq=150, orient="FR"):
    """-> list of (regs0, regs1): arrays of ALNREG_DT"""
    n_seqs = len(offs) - 1

    def place(p_f, length, rev):
        """[rb, re) of `length` reference bases whose forward-strand start is p_f, on the strand asked for (doubled coordinate)"""
        return (2 * l_pac - (p_f + length), 2 * l_pac - p_f) if rev else (p_f, p_f + length)

    def rid_of(p_f):
        return int(np.searchsorted(offs, p_f, side="right") - 1)

    out = []
    for it in range(n_pairs):
        k = int(rng.integers(0, n_seqs))
        frag = int(rng.choice([int(rng.normal(400, 50)), int(rng.integers(150, 3000)), int(rng.integers(160, 640))]))
        frag = max(frag, lq + 5)
        p = int(rng.integers(offs[k] + 10, max(offs[k] + 11, offs[k + 1] - frag - lq - 10)))
        flip = rng.random() < 0.5                 # which read is the forward one
        ends = []
        for e in range(2):
            regs = []
            n = int(rng.choice([1, 1, 2, 2, 3, 4, 5, 6, 8, 9]))
            fwd_read = (e == 0) != flip
            base_p = p if fwd_read == (orient == "FR") else p + frag - lq   # FR: the forward read upstream; RF: downstream
            base_rev = not fwd_read
            mode = rng.random()
            if mode < 0.08:                        # mate in the same orientation (FF / RR) or swapped (RF)
                base_rev = bool(rng.integers(0, 2))
            sc0 = int(rng.choice([150, 145, 140, 120, 100, 80, 60, 45, 31, 30]))
            for j in range(n):
                kind = rng.random() if j else 0.0
                if kind < 0.35:                    # the hit itself, possibly clipped
                    qb = int(rng.choice([0, 0, 0, 5, 30])); qe = lq - int(rng.choice([0, 0, 0, 7, 40]))
                    pf, rev, sc = base_p + (qb if not base_rev else lq - qe), base_rev, sc0 - 0 * j
                    ln = qe - qb + int(rng.choice([0, 0, 0, 1, -2]))
                elif kind < 0.6:                   # overlaps the first hit: contained / shifted (redundancy, patching tests)
                    qb = int(rng.integers(0, 60)); qe = int(rng.integers(90, lq + 1))
                    sh = int(rng.choice([0, 0, 1, -1, 3, 10, 60]))
                    pf, rev = base_p + (qb if not base_rev else lq - qe) + sh, base_rev
                    sc = int(rng.choice([sc0, sc0, sc0 - 5, sc0 - 20, 40, 25]))
                    ln = qe - qb
                elif kind < 0.8:                   # a chance hit elsewhere (same or other contig, any strand), short
                    kk = int(rng.integers(0, n_seqs))
                    pf = int(rng.integers(offs[kk] + 10, offs[kk + 1] - lq - 10))
                    rev = bool(rng.integers(0, 2))
                    qb = int(rng.integers(0, lq - 30)); qe = min(lq, qb + int(rng.integers(19, 40)))
                    sc = int(rng.choice([qe - qb, qe - qb - 5, 20, 19, sc0]))
                    ln = qe - qb
                else:                              # a second full-length copy (repeat): near or far, equal or close score
                    d = int(rng.choice([300, 1000, 20000, -500]))
                    pf = min(max(base_p + d, offs[k] + 10), offs[k + 1] - lq - 10)
                    rev = base_rev if rng.random() < 0.7 else not base_rev
                    qb, qe = 0, lq
                    sc = int(rng.choice([sc0, sc0, sc0 - 1, sc0 - 4, sc0 - 6, sc0 - 30]))
                    ln = lq
                sc = max(sc, 19)
                ln = max(ln, 19)
                rb, re = place(int(pf), int(ln), rev)
                tsc = sc if rng.random() < 0.8 else max(sc - int(rng.integers(1, 20)), 1)
                regs.append(_reg(rb, re, qb, qe, rid_of(int(pf)), sc, truesc=tsc, w=int(rng.choice([100, 100, 200, 37])),
                                 seedlen0=int(rng.integers(19, 60)), frac_rep=float(rng.choice([0.0, 0.0, 0.2, 0.5]))))
            ends.append(np.array(regs, dtype=po.ALNREG_DT))
        out.append((ends[0], ends[1]))
    return out
