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
