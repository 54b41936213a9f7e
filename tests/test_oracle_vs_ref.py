"""Pin the plain-C oracle restatement against the REAL reference compiled from its own sources
(oracle/_ref/libbwaref.so).  Runs wherever that build exists (this container and, as a prebuilt .so, the GPU box)."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libbwaref.so not built")


def _sorted_rows(a):
    return np.array(sorted(map(tuple, a.tolist())), dtype=np.uint64).reshape(-1, 4)


def test_smem_intervals_match_reference(genome, reads_pe, reads_var):
    ref = po.RefIndex(genome["prefix"])
    fm = po.OracleFM(genome["prefix"])
    opt = ref.opt()
    n = 0
    for name, r1, r2 in reads_pe[:200] + reads_var[:150]:
        for r in (r1, r2):
            if r is None:
                continue
            a = ref.collect_intv(opt, r.copy())
            b = fm.collect_intv(r)
            assert a.shape == b.shape
            assert (_sorted_rows(a) == _sorted_rows(b)).all()
            assert (np.diff(b[:, 3].astype(np.int64)) >= 0).all() if len(b) > 1 else True
            n += len(b)
    assert n > 1000


def test_smem_edge_cases(genome):
    ref = po.RefIndex(genome["prefix"])
    fm = po.OracleFM(genome["prefix"])
    opt = ref.opt()
    g = genome["seqs"][0]
    cases = [
        np.zeros(0, np.uint8) + 0,                     # (empty is never passed by the caller; shortest real case next)
        np.array([0, 1, 2, 3] * 4, np.uint8),          # shorter than min_seed_len
        np.full(40, 4, np.uint8),                      # all N
        np.concatenate([g[1000:1060], [4], g[1061:1150]]).astype(np.uint8),   # N in the middle
        np.array([0] * 150, np.uint8),                 # homopolymer
        g[5000:5019].astype(np.uint8),                 # exactly min_seed_len
    ]
    for r in cases[1:]:
        r = np.where(r > 4, 0, r).astype(np.uint8)
        a = ref.collect_intv(opt, r.copy())
        b = fm.collect_intv(r)
        assert a.shape == b.shape and (_sorted_rows(a) == _sorted_rows(b)).all()


def test_sa_lookup_matches_reference(genome):
    ref = po.RefIndex(genome["prefix"])
    fm = po.OracleFM(genome["prefix"])
    rng = np.random.default_rng(3)
    ks = list(rng.integers(0, fm.fm.seq_len + 1, size=3000)) + [0, 1, int(fm.fm.primary), int(fm.fm.primary) + 1,
                                                                 int(fm.fm.seq_len)]
    for k in ks:
        assert ref.sa_lookup(k) == fm.sa_lookup(k)


def _rand_pair(rng, qlen, div):
    q = rng.integers(0, 4, size=qlen, dtype=np.uint8)
    t = list(q)
    out = []
    for b in t:
        u = rng.random()
        if u < div:
            out.append((b + 1 + rng.integers(0, 3)) & 3)
        elif u < div * 1.3:
            continue
        elif u < div * 1.6:
            out += [b, rng.integers(0, 4)]
        else:
            out.append(b)
    out += list(rng.integers(0, 4, size=rng.integers(0, 60)))
    return q, np.array(out, dtype=np.uint8)


def test_extend2_matches_reference(genome):
    ref = po.RefIndex(genome["prefix"])
    opt = ref.opt().contents
    mat = np.array(list(opt.mat), dtype=np.int8)
    rng = np.random.default_rng(5)
    for it in range(1500):
        qlen = int(rng.integers(1, 260))
        q, t = _rand_pair(rng, qlen, float(rng.choice([0.0, 0.02, 0.1, 0.3])))
        if rng.random() < 0.1:
            q[rng.integers(0, qlen)] = 4
        if len(t) == 0:
            continue
        w = int(rng.choice([100, 200, 5, 30]))
        h0 = int(rng.integers(1, 200))
        eb = int(rng.choice([5, 0]))
        zd = int(rng.choice([100, 0, 20]))
        a = ref.extend2(q, t, mat, 6, 1, 6, 1, w, eb, zd, h0)
        b, cells = po.oracle_extend2(q, t, mat, 6, 1, 6, 1, w, eb, zd, h0)
        assert (a == b).all(), (it, a, b)


def test_global2_matches_reference(genome):
    ref = po.RefIndex(genome["prefix"])
    opt = ref.opt().contents
    mat = np.array(list(opt.mat), dtype=np.int8)
    rng = np.random.default_rng(6)
    for it in range(600):
        qlen = int(rng.integers(1, 200))
        q, t = _rand_pair(rng, qlen, float(rng.choice([0.0, 0.02, 0.1])))
        t = t[:max(1, len(q) + int(rng.integers(-8, 9)))]
        w = int(abs(len(t) - len(q)) + rng.integers(3, 40))
        sa, ca = ref.global2(q, t, mat, 6, 1, 6, 1, w)
        sb, cb = po.oracle_global2(q, t, mat, 6, 1, 6, 1, w)
        assert sa == sb and len(ca) == len(cb) and (ca == cb).all()
