"""Seeded synthetic genomes and reads (SURVEY.md §8d generator).

The GPU box has no reference data and no network, so every test / bench input
is generated here from a seed: a multi-contig genome with an order-3 Markov
base composition, planted repeat families and optional N runs, and single- or
paired-end reads with substitutions, small indels and a share of random
(unmappable) reads.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def make_genome(total_len: int, n_contigs: int = 4, seed: int = 38, repeat_frac: float = 0.05,
                n_runs: int = 2, n_run_len: int = 200):
    """Return (names, seqs) with seqs a list of uint8 arrays of codes 0..4 (4 = N)."""
    rng = np.random.default_rng(seed)
    # order-3 Markov chain with GC ~ 41 %: per-context categorical drawn once from a Dirichlet
    base_p = np.array([0.295, 0.205, 0.205, 0.295])
    trans = rng.dirichlet(base_p * 12.0, size=64)
    cum = np.cumsum(trans, axis=1)
    lens = np.full(n_contigs, total_len // n_contigs, dtype=np.int64)
    lens[-1] += total_len - lens.sum()
    seqs = []
    for ci in range(n_contigs):
        L = int(lens[ci])
        u = rng.random(L)
        s = np.empty(L, dtype=np.uint8)
        ctx = int(rng.integers(64))
        # vectorising a Markov chain exactly is awkward; generate in blocks with a python loop over
        # positions only for small genomes, else fall back to block-independent draws
        if L <= 400_000:
            for i in range(L):
                c = int(np.searchsorted(cum[ctx], u[i]))
                if c > 3:
                    c = 3
                s[i] = c
                ctx = ((ctx << 2) | c) & 63
        else:
            s[:] = np.minimum(np.searchsorted(np.cumsum(base_p), u), 3)
        seqs.append(s)
    # planted repeat families
    n_rep_bases = int(total_len * repeat_frac)
    placed = 0
    while placed < n_rep_bases:
        unit_len = int(rng.integers(200, 2000))
        copies = int(rng.integers(2, 40))
        div = float(rng.uniform(0.0, 0.05))
        src_c = int(rng.integers(n_contigs))
        if len(seqs[src_c]) <= unit_len + 1:
            break
        src_p = int(rng.integers(0, len(seqs[src_c]) - unit_len))
        unit = seqs[src_c][src_p:src_p + unit_len].copy()
        for _ in range(copies):
            c = int(rng.integers(n_contigs))
            if len(seqs[c]) <= unit_len + 1:
                continue
            p = int(rng.integers(0, len(seqs[c]) - unit_len))
            cp = unit.copy()
            if rng.random() < 0.5:
                cp = _COMP[cp[::-1]]
            mut = rng.random(unit_len) < div
            cp[mut] = rng.integers(0, 4, size=int(mut.sum()), dtype=np.uint8)
            seqs[c][p:p + unit_len] = cp
            placed += unit_len
    for _ in range(n_runs):
        c = int(rng.integers(n_contigs))
        if len(seqs[c]) > 4 * n_run_len:
            p = int(rng.integers(0, len(seqs[c]) - n_run_len))
            seqs[c][p:p + n_run_len] = 4
    names = [f"chrS{i + 1}" for i in range(n_contigs)]
    return names, seqs


def write_fasta(path: str, names, seqs, width: int = 60):
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    with open(path, "wb") as fp:
        for name, s in zip(names, seqs):
            fp.write(b">" + name.encode() + b"\n")
            asc = lut[s]
            for i in range(0, len(asc), width):
                fp.write(asc[i:i + width].tobytes())
                fp.write(b"\n")


def _mutate(frag: np.ndarray, rng, sub: float, indel: float) -> np.ndarray:
    out = frag.copy()
    m = rng.random(len(out)) < sub
    if m.any():
        out[m] = (out[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
    if indel > 0:
        k = int(rng.binomial(len(out), indel))
        for _ in range(k):
            p = int(rng.integers(1, max(2, len(out) - 1)))
            l = int(rng.geometric(0.5))
            if rng.random() < 0.5:
                out = np.concatenate([out[:p], out[p + l:]])
            else:
                out = np.concatenate([out[:p], rng.integers(0, 4, size=l, dtype=np.uint8), out[p:]])
    return out


def simulate_reads(seqs, n: int, read_len=150, paired: bool = True, seed: int = 1, frag_mean: float = 400.0,
                   frag_sd: float = 50.0, sub: float = 0.01, indel: float = 0.001, frac_random: float = 0.02,
                   var_len=None, n_frac: float = 0.0005):
    """Return a list of (name, seq1, seq2|None) with seqs as uint8 code arrays (0..4).

    var_len=(lo,hi) draws each read length uniformly (trimmed-read stress, config 3)."""
    rng = np.random.default_rng(seed)
    clen = np.array([len(s) for s in seqs], dtype=np.float64)
    cprob = clen / clen.sum()
    out = []
    for i in range(n):
        l1 = l2 = read_len
        if var_len is not None:
            l1 = int(rng.integers(var_len[0], var_len[1] + 1))
            l2 = int(rng.integers(var_len[0], var_len[1] + 1))
        name = f"r{i}"
        if rng.random() < frac_random:
            r1 = rng.integers(0, 4, size=l1, dtype=np.uint8)
            r2 = rng.integers(0, 4, size=l2, dtype=np.uint8) if paired else None
            out.append((name, r1, r2))
            continue
        c = int(rng.choice(len(seqs), p=cprob))
        fl = int(max(rng.normal(frag_mean, frag_sd), max(l1, l2) + 10)) if paired else l1 + 20
        fl = min(fl, len(seqs[c]) - 1)
        p = int(rng.integers(0, len(seqs[c]) - fl))
        frag = seqs[c][p:p + fl].copy()
        nmask = frag > 3
        if nmask.any():
            frag[nmask] = rng.integers(0, 4, size=int(nmask.sum()), dtype=np.uint8)
        if rng.random() < 0.5:
            frag = _COMP[frag[::-1]]
        frag = _mutate(frag, rng, sub, indel)
        r1 = frag[:l1].copy()
        r2 = _COMP[frag[::-1]][:l2].copy() if paired else None
        for r in (r1, r2):
            if r is not None and n_frac > 0:
                m = rng.random(len(r)) < n_frac
                r[m] = 4
        out.append((name, r1, r2))
    return out


def reads_to_ascii(reads):
    """[(name, bytes seq1, bytes|None seq2)] with ACGTN letters."""
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    return [(n, lut[a].tobytes(), None if b is None else lut[b].tobytes()) for n, a, b in reads]


def write_fastq(path: str, reads, which: int):
    with open(path, "wb") as fp:
        for name, s1, s2 in reads_to_ascii(reads):
            s = s1 if which == 0 else s2
            fp.write(b"@" + name.encode() + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
