"""The GPU index builder must write the same .bwt / .sa bytes as the CPU builder (itself byte-identical to the
reference's own hg19.small index, tests/test_index.py)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mbp,seed,wide", [(0.3, 3, False), (2.0, 5, False), (5.0, 38, False), (3.0, 7, True)])
def test_gpu_builder_matches_cpu_builder(tmp_path, built, mbp, seed, wide, monkeypatch):
    """wide: a repeat-rich genome (SURVEY §8d's generator, half of it repeats) through the doubling rounds' two-pass sort, the
    path the builder takes by itself beyond 2^30 tied suffixes."""
    from mpibwa_amd import api, bigindex
    lib = api.load_library()
    if wide:
        monkeypatch.setenv("MPIBWA_IDX_WIDE", "1")
        pac, lens = bigindex.synth_packed_genome_grch38like(mbp * 1e6, seed=seed, n_contigs=3, repeat_frac=0.5)
    else:
        pac, lens = bigindex.synth_packed_genome(mbp * 1e6, seed=seed, n_contigs=3, repeat_frac=0.08)
    l_pac = int(lens.sum())
    # CPU path: FASTA -> mi355x_index_build
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    bases = bigindex.unpack_windows(pac, np.array([0], dtype=np.int64), l_pac)[0]
    fa = str(tmp_path / "c.fa")
    with open(fa, "wb") as f:
        off = 0
        for i, L in enumerate(lens):
            f.write(b">chrS%d\n" % (i + 1))
            f.write(lut[bases[off:off + int(L)]].tobytes() + b"\n")
            off += int(L)
    api.build_index(fa, fa)
    # GPU path
    g = str(tmp_path / "g.fa")
    bigindex.write_meta_files(g, pac, lens)
    secs = C.c_double(0)
    assert lib.mi355x_index_build_gpu(0, pac.ctypes.data, l_pac, g.encode(), C.byref(secs)) == 0
    for ext in ("pac", "ann", "amb", "bwt", "sa"):
        a, b = open(fa + "." + ext, "rb").read(), open(g + "." + ext, "rb").read()
        assert a == b, (ext, len(a), len(b))
