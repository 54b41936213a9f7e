"""Full-size index paths in the gated suite: a synthetic reference of 2.3 Gbp (4.6 G suffixes: more than 2^32, so the GPU
index builder's wide suffix numbers, the seeding kernel's 34-bit packed interval bounds, the dense suffix array and the
third-pass jump table are all in their production regime), 24 000 read pairs against the compiled reference."""
import os
import time

import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libbwaref.so not present")
def test_index_beyond_2_32_suffixes_end_to_end(tmp_path_factory, built):
    import ctypes as C
    from mpibwa_amd import abi, api, bigindex
    lib = api.load_library()
    lib.mi355x_finalize()
    d = os.environ.get("MPIBWA_BENCH_DIR", "/tmp/mpibwa_bench")     # shared with bench.py: a completed build is reused
    os.makedirs(d, exist_ok=True)
    t0 = time.time()
    idx = bigindex.make_or_get(d, genome_mbp=2300.0, seed=23)
    eng = idx.engine
    assert int(eng.bwt.contents.seq_len) > (1 << 32)
    dense_bytes = C.c_size_t(0)
    lib.mi355x_sa_dense_info(C.byref(dense_bytes))
    assert dense_bytes.value == (int(eng.bwt.contents.seq_len) + 1) * 8          # dense SA in HBM
    reads = idx.simulate_pairs(24000, seed=5, read_len=150)
    ref = po.RefIndex(idx.prefix)
    C.c_int.in_dll(ref.lib, "bwa_verbose").value = 1
    kw = dict(flag=abi.MEM_F_PE, n_threads=int(lib.mi355x_host_cpus()))
    want = ref.process(ref.opt(**kw), reads)
    got = eng.process(eng.opt(**kw), reads)
    assert got == want
    st = eng.stats()
    assert st["n_seeds"] > 24000 and st["n_aln"] > 0
    # kernel-level: intervals above 2^32 (rows of the reverse-strand half) and SA values above 2^32
    import numpy as np
    codes = [np.frombuffer(r[1].translate(bytes.maketrans(b"ACGTN", bytes(range(5)))), dtype=np.uint8) for r in reads[:200]]
    iv, _, _ = eng.smem(eng.opt(), codes, cap=512)
    big_rows = [int(a[:, 0].max()) for a in iv if len(a)]
    assert max(big_rows) > (1 << 32)
    for c, a in list(zip(codes, iv))[:40]:
        assert (a == ref.collect_intv(ref.opt(), c.copy())).all()
    rows = np.unique(np.concatenate([a[:, 0] for a in iv if len(a)]))[:4000]
    sa_d = eng.sa_dense(rows)
    sa_w, _, _ = eng.sa(rows)
    assert sa_d is not None and (sa_d[0] == sa_w).all() and int(sa_w.max()) > (1 << 32)
    for k, v in list(zip(rows, sa_w))[:300]:
        assert int(v) == ref.sa_lookup(int(k))
    lib.mi355x_finalize()
    C.c_int.in_dll(ref.lib, "bwa_verbose").value = 3
