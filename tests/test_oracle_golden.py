"""The oracle restatement against the COMMITTED golden vectors (generated from the compiled reference by
tools/make_golden.py).  Runs everywhere, with or without oracle/_ref."""
import numpy as np
import pytest

from oracle import pyoracle as po
from golden_util import kernel_vectors, ragged, golden_index

MAT = np.array([1, -4, -4, -4, -1, -4, 1, -4, -4, -1, -4, -4, 1, -4, -1, -4, -4, -4, 1, -1, -1, -1, -1, -1, -1], dtype=np.int8)


@pytest.fixture(scope="module")
def gold(tmp_path_factory, built):
    return golden_index(tmp_path_factory.mktemp("gold"))


def test_oracle_smem_vs_golden(gold):
    kv = kernel_vectors()
    fm = po.OracleFM(gold)
    reads, outs = ragged(kv, "intv_reads"), ragged(kv, "intv_out")
    assert len(reads) == len(outs) > 100
    for r, o in zip(reads, outs):
        got = fm.collect_intv(r)
        assert (got.reshape(-1) == o).all()


def test_oracle_sa_vs_golden(gold):
    kv = kernel_vectors()
    fm = po.OracleFM(gold)
    for k, v in zip(kv["sa_k"], kv["sa_v"]):
        assert fm.sa_lookup(int(k)) == int(v)


def test_oracle_extend_vs_golden(built):
    kv = kernel_vectors()
    qs, ts = ragged(kv, "ext_q"), ragged(kv, "ext_t")
    for q, t, p, o in zip(qs, ts, kv["ext_p"], kv["ext_o"]):
        got, _ = po.oracle_extend2(q, t, MAT, 6, 1, 6, 1, int(p[0]), int(p[2]), int(p[3]), int(p[1]))
        assert (got == o).all()


def test_oracle_global_vs_golden(built):
    kv = kernel_vectors()
    qs, ts, cs = ragged(kv, "glo_q"), ragged(kv, "glo_t"), ragged(kv, "glo_c")
    for q, t, w, s, c in zip(qs, ts, kv["glo_w"], kv["glo_s"], cs):
        sc, cg = po.oracle_global2(q, t, MAT, 6, 1, 6, 1, int(w))
        assert sc == int(s) and len(cg) == len(c) and (cg == c).all()
