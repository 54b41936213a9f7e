"""Parity of the three HIP kernels against the oracle, through the C ABI (stage-level entry points)."""
import numpy as np
import pytest

from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(genome):
    from mpibwa_amd import api
    return api.Engine(genome["prefix"], device=0)


def _flatten(reads):
    out = []
    for _, r1, r2 in reads:
        out.append(r1)
        if r2 is not None:
            out.append(r2)
    return out


def test_smem_kernel_matches_oracle(engine, genome, reads_pe, reads_var):
    fm = po.OracleFM(genome["prefix"])
    seqs = _flatten(reads_pe) + _flatten(reads_var)
    g = genome["seqs"][0]
    seqs += [np.array([0, 1, 2, 3] * 4, np.uint8), np.full(40, 4, np.uint8), np.zeros(150, np.uint8),
             np.concatenate([g[1000:1060], [4], g[1061:1150]]).astype(np.uint8), g[5000:5019].astype(np.uint8),
             np.where(g[9000:9400] > 3, 0, g[9000:9400]).astype(np.uint8)]
    opt = engine.opt()
    fm.reset_counters()
    got, ms, nbytes = engine.smem(opt, seqs, cap=512)
    total_in = 0
    for s, a in zip(seqs, got):
        if len(s) < 19:   # src/bwamem.c:260: reads shorter than a seed never reach mem_collect_intv
            assert len(a) == 0
            total_in += len(s)
            continue
        b = fm.collect_intv(s)
        assert a.shape == b.shape, (len(s), a.shape, b.shape)
        assert (a == b).all()
        total_in += len(s)
    # the device's own count of occ blocks equals the oracle's instrumented count (SURVEY §8d definition)
    n_intv = sum(len(a) for a in got)
    assert nbytes == fm.fm.n_blocks * 64 + total_in + 32 * n_intv


def test_smem_kernel_overflow_is_reported(engine, reads_pe):
    opt = engine.opt()
    with pytest.raises(RuntimeError):
        engine.smem(opt, _flatten(reads_pe)[:64], cap=1)


def test_sa_kernel_matches_oracle(engine, genome):
    fm = po.OracleFM(genome["prefix"])
    rng = np.random.default_rng(9)
    ks = np.concatenate([rng.integers(0, fm.fm.seq_len + 1, size=20000).astype(np.uint64),
                         np.array([0, 1, fm.fm.primary, fm.fm.primary + 1, fm.fm.seq_len, 32, 64], dtype=np.uint64)])
    fm.reset_counters()
    got, ms, nbytes = engine.sa(ks)
    want = np.array([fm.sa_lookup(k) for k in ks], dtype=np.uint64)
    assert (got == want).all()
    assert nbytes == fm.fm.n_sa_steps * 64 + 8 * len(ks)


def test_dense_sa_matches_walk_and_oracle(engine, genome):
    """The SA table expanded in HBM must return exactly what the LF walk (and the oracle) returns, for every row class."""
    fm = po.OracleFM(genome["prefix"])
    n = int(fm.fm.seq_len)
    rng = np.random.default_rng(10)
    ks = np.concatenate([rng.integers(1, n + 1, size=30000).astype(np.uint64),
                         np.array([1, fm.fm.primary, fm.fm.primary + 1, n, 32, 64, n - 1], dtype=np.uint64)])
    res = engine.sa_dense(ks)
    assert res is not None, "dense SA was not expanded at upload"
    dense, _ = res
    walk, _, _ = engine.sa(ks)
    assert (dense == walk).all()
    want = np.array([fm.sa_lookup(int(k)) for k in ks[:3000]], dtype=np.uint64)
    assert (dense[:3000] == want).all()
    # exhaustive on a slice of rows: every row, not a sample
    rows = np.arange(1, 70000, dtype=np.uint64)
    d2, _ = engine.sa_dense(rows)
    w2, _, _ = engine.sa(rows)
    assert (d2 == w2).all()


def _rand_pair(rng, qlen, div):
    q = rng.integers(0, 4, size=qlen, dtype=np.uint8)
    out = []
    for b in q:
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
    if not out:
        out = [0]
    return q, np.array(out, dtype=np.uint8)


def test_extend_kernel_matches_oracle(engine):
    opt = engine.opt()
    o = opt.contents
    mat = np.array(list(o.mat), dtype=np.int8)
    rng = np.random.default_rng(21)
    qs, ts, ws, h0s, ebs = [], [], [], [], []
    for it in range(3000):
        qlen = int(rng.choice([1, 2, 17, 63, 64, 65, 126, 127, 128, 129, 131, 150, 190, 191, 192, 193, 231, 300, 319, 320, 321, 400, 640,
                                   int(rng.integers(1, 330))]))
        q, t = _rand_pair(rng, qlen, float(rng.choice([0.0, 0.01, 0.05, 0.15, 0.4])))
        if rng.random() < 0.1:
            q[rng.integers(0, qlen)] = 4
        qs.append(q); ts.append(t)
        ws.append(int(rng.choice([100, 200, 7, 40])))
        h0s.append(int(rng.integers(1, 180)))
        ebs.append(int(rng.choice([5, 0])))
    got, ms, cells = engine.extend(opt, qs, ts, ws, h0s, ebs)
    tot = 0
    for i in range(len(qs)):
        want, c = po.oracle_extend2(qs[i], ts[i], mat, o.o_del, o.e_del, o.o_ins, o.e_ins, ws[i], ebs[i], o.zdrop, h0s[i])
        assert (got[i] == want).all(), (i, len(qs[i]), len(ts[i]), ws[i], h0s[i], got[i], want)
        tot += c
    assert cells == tot
