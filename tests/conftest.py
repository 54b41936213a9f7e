import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Product library + oracle built in-tree (the reference lib only where /root/reference exists)."""
    from mpibwa_amd.build import build
    build()
    from oracle import pyoracle
    pyoracle.build()
    return True


@pytest.fixture(scope="session")
def genome(tmp_path_factory, built):
    """Seeded 3-contig synthetic genome (with repeats and N runs) and its bwa-format index built by the product."""
    from mpibwa_amd import simulate, api
    d = tmp_path_factory.mktemp("genome")
    names, seqs = simulate.make_genome(360_000, 3, seed=7)
    fa = str(d / "g.fa")
    simulate.write_fasta(fa, names, seqs)
    api.build_index(fa, fa)
    return {"prefix": fa, "names": names, "seqs": seqs}


@pytest.fixture(scope="session")
def reads_pe(genome):
    from mpibwa_amd import simulate
    return simulate.simulate_reads(genome["seqs"], 600, 150, paired=True, seed=11)


@pytest.fixture(scope="session")
def reads_var(genome):
    from mpibwa_amd import simulate
    return simulate.simulate_reads(genome["seqs"], 300, 150, paired=False, seed=12, var_len=(30, 300))


def make_alt_genome(seqs, names, seed=21, n_alt=4):
    """Append ALT contigs to a genome: each one is a diverged copy (substitutions and a few indels) of a primary region, named
    in a SAM-style `.alt` file next to the index like bwa's GRCh38 `.alt` (src/bntseq.c:179-204: the first column of every
    non-@ line names an ALT contig).  Returns (names, seqs, alt_names)."""
    from mpibwa_amd import simulate
    rng = np.random.default_rng(seed)
    names, seqs, alt = list(names), [s.copy() for s in seqs], []
    n_pri = len(seqs)
    for k in range(n_alt):
        c = int(rng.integers(n_pri))
        L = int(rng.integers(6000, 16000))
        p = int(rng.integers(1000, len(seqs[c]) - L - 1000))
        src = seqs[c][p:p + L].copy()
        src[src > 3] = rng.integers(0, 4, size=int((src > 3).sum()), dtype=np.uint8)
        cp = simulate._mutate(src, rng, sub=float(rng.uniform(0.004, 0.03)), indel=0.0008)
        if k % 2:
            cp = simulate._COMP[cp[::-1]]
        names.append("%s_alt%d" % (names[c], k + 1))
        seqs.append(cp.astype(np.uint8))
        alt.append(names[-1])
    return names, seqs, alt


def write_alt_file(prefix, alt_names, seqs_by_name):
    with open(prefix + ".alt", "w") as f:
        f.write("@HD\tVN:1.0\n@SQ\tSN:ignored_header_line\tLN:1\n")
        for n in alt_names:
            f.write("%s\t0\tchrX\t1\t60\t%dM\t*\t0\t0\t*\t*\n" % (n, len(seqs_by_name[n])))
        f.write("not_a_contig\t0\n")


@pytest.fixture(scope="session")
def genome_alt(tmp_path_factory, genome):
    """The session genome plus four ALT contigs, indexed with an `.alt` file."""
    from mpibwa_amd import simulate, api
    d = tmp_path_factory.mktemp("genome_alt")
    names, seqs, alt = make_alt_genome(genome["seqs"], genome["names"])
    fa = str(d / "ga.fa")
    simulate.write_fasta(fa, names, seqs)
    write_alt_file(fa, alt, dict(zip(names, seqs)))
    api.build_index(fa, fa)
    return {"prefix": fa, "names": names, "seqs": seqs, "alt": alt}
