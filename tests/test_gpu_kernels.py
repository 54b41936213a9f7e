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


def _rc(a):
    a = np.asarray(a, np.uint8)
    return np.where(a > 3, 4, 3 - a)[::-1].astype(np.uint8)


def test_smem_kernel_matches_oracle(engine, genome, reads_pe, reads_var):
    fm = po.OracleFM(genome["prefix"])
    seqs = _flatten(reads_pe) + _flatten(reads_var)
    g = genome["seqs"][0]
    seqs += [np.array([0, 1, 2, 3] * 4, np.uint8), np.full(40, 4, np.uint8), np.zeros(150, np.uint8),
             np.concatenate([g[1000:1060], [4], g[1061:1150]]).astype(np.uint8), g[5000:5019].astype(np.uint8),
             np.where(g[9000:9400] > 3, 0, g[9000:9400]).astype(np.uint8)]
    # the ends of the text and the junction between the strands (the text is the reference followed by its reverse complement)
    first, last = np.asarray(genome["seqs"][0], np.uint8), np.asarray(genome["seqs"][-1], np.uint8)
    if (first[:300] < 4).all() and (last[-300:] < 4).all():
        head, tail = first[:150], last[-150:]
        seqs += [head, _rc(head), tail, _rc(tail), first[:260], _rc(last[-260:]),
                 np.concatenate([last[-75:], _rc(last[-75:])]), np.concatenate([_rc(first[:75]), first[:75]]),
                 np.concatenate([last[-140:], _rc(last[-10:])]), np.concatenate([last[-200:], _rc(last[-100:])]),
                 np.concatenate([[1], head[1:]]).astype(np.uint8), np.concatenate([tail[:-1], [2]]).astype(np.uint8)]
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
    n_intv = sum(len(a) for a in got)
    # the device's own count of occ blocks equals the oracle's instrumented count (SURVEY §8d definition)
    assert nbytes == fm.fm.n_blocks * 64 + total_in + 32 * n_intv


@pytest.mark.parametrize("kmt", ["4", "9", "0", "default"])   # (the last one leaves the module's index as every other test expects it)
def test_production_smem_kernel_matches_oracle(genome, reads_pe, reads_var, kmt, monkeypatch):
    """The instantiation mem_process_seqs launches (no block counting; short results come from the k-mer tables, depth `kmt`:
    MPIBWA_KMT, 0 = every extension through the occ table) against the oracle's mem_collect_intv: reads of all lengths, text ends,
    the strand junction, ambiguous bases, low-complexity reads (re-seeding with min_intv > 1)."""
    from mpibwa_amd import api
    if kmt != "default":
        monkeypatch.setenv("MPIBWA_KMT", kmt)
    eng = api.Engine(genome["prefix"], device=0)   # (the tables are built at upload)
    monkeypatch.setenv("MPIBWA_SMEM_COUNT", "0")
    fm = po.OracleFM(genome["prefix"])
    seqs = _flatten(reads_pe) + _flatten(reads_var)
    g = genome["seqs"][0]
    first, last = np.asarray(genome["seqs"][0], np.uint8), np.asarray(genome["seqs"][-1], np.uint8)
    seqs += [np.array([0, 1, 2, 3] * 40, np.uint8), np.zeros(150, np.uint8), np.array([0, 1] * 75, np.uint8), np.full(40, 4, np.uint8),
             np.concatenate([g[1000:1060], [4], g[1061:1150]]).astype(np.uint8), g[5000:5019].astype(np.uint8),
             np.concatenate([g[2000:2010], [4], g[2011:2030], [4, 4], g[2032:2150]]).astype(np.uint8),
             np.where(g[9000:9400] > 3, 0, g[9000:9400]).astype(np.uint8)]
    if (first[:300] < 4).all() and (last[-300:] < 4).all():
        head, tail = first[:150], last[-150:]
        seqs += [head, _rc(head), tail, _rc(tail), first[:260], _rc(last[-260:]),
                 np.concatenate([last[-75:], _rc(last[-75:])]), np.concatenate([_rc(first[:75]), first[:75]]),
                 np.concatenate([last[-140:], _rc(last[-10:])]), np.concatenate([[1], head[1:]]).astype(np.uint8)]
    got, ms, nbytes = eng.smem(eng.opt(), seqs, cap=512)
    assert nbytes == 0   # (nothing counted in this variant)
    for s, a in zip(seqs, got):
        b = fm.collect_intv(s) if len(s) >= 19 else np.zeros((0, 4), np.uint64)
        assert a.shape == b.shape, (len(s), a.shape, b.shape)
        assert (a == b).all()


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


def _matesw_cases(rng, l_pac, ref, n_req, read_lens):
    """Random mate-rescue problems: windows of the doubled reference, reads planted in them with noise and gaps."""
    dref = np.concatenate([ref, 3 - ref[::-1]]).astype(np.uint8)   # the doubled coordinate: forward, then reverse complement
    reads, rb, re, rd, rev = [], [], [], [], []
    for i in range(n_req):
        ql = int(rng.choice(read_lens))
        tl = int(rng.integers(max(ql // 2, 19), ql + 600))
        strand = int(rng.integers(0, 2))
        b = int(rng.integers(0, l_pac - tl)) + strand * l_pac
        win = dref[b:b + tl]
        mode = rng.random()
        if mode < 0.75 and tl >= ql // 2:
            p = int(rng.integers(-ql // 3, tl - ql // 2))
            src = win[max(p, 0):max(p, 0) + ql].copy()
            q = rng.integers(0, 4, size=ql).astype(np.uint8)
            q[:len(src)] = src
            mut = rng.random(ql) < rng.choice([0.0, 0.02, 0.08, 0.2])
            q[mut] = rng.integers(0, 4, size=int(mut.sum()))
            for _ in range(int(rng.integers(0, 4))):   # gaps, long ones included (they cross the striped segments)
                g, at = int(rng.integers(1, 25)), int(rng.integers(1, ql - 1))
                if rng.random() < 0.5:
                    q = np.concatenate([q[:at], rng.integers(0, 4, size=g).astype(np.uint8), q[at:]])[:ql]
                else:
                    q = np.concatenate([q[:at], q[at + g:], rng.integers(0, 4, size=g).astype(np.uint8)])[:ql]
            if rng.random() < 0.25:   # a second, weaker copy far away -> score2 / te2
                p2 = int(rng.integers(0, max(1, tl - ql)))
                q2 = q.copy()
                m2 = rng.random(ql) < 0.1
                q2[m2] = rng.integers(0, 4, size=int(m2.sum()))
                seg = q2[:min(ql, tl - p2)]
                # the window is what it is: plant into the read instead (repeat inside the read)
                q[:len(seg) // 3] = win[p2:p2 + len(seg) // 3]
        elif mode < 0.9:   # low complexity
            q = np.repeat(rng.integers(0, 4, size=(ql + 2) // 3), 3)[:ql].astype(np.uint8)
        else:
            q = rng.integers(0, 4, size=ql).astype(np.uint8)
        if rng.random() < 0.1:
            q[int(rng.integers(0, ql))] = 4
        is_rev = int(rng.integers(0, 2))
        # the kernel aligns the (reverse-complemented) read: store the read so that its transform is q
        stored = q if not is_rev else np.where(q[::-1] < 4, 3 - q[::-1], 4).astype(np.uint8)
        reads.append(stored)
        rb.append(b); re.append(b + tl); rd.append(len(reads) - 1); rev.append(is_rev)
        # more windows for the same mate in the same orientation, as the anchors of a repeat read ask for: requests next to each other
        # that the library aligns two per quad (msw2_kernel) — the same window shifted, another place, a much shorter / longer one
        for _ in range(int(rng.choice([0, 0, 1, 1, 2, 3]))):
            kind = rng.random()
            if kind < 0.4:
                b2 = min(max(b + int(rng.integers(-200, 200)), strand * l_pac), strand * l_pac + l_pac - tl)
                tl2 = tl
            else:
                tl2 = int(rng.integers(max(ql // 2, 19), ql + 600))
                s2 = int(rng.integers(0, 2))
                b2 = int(rng.integers(0, l_pac - tl2)) + s2 * l_pac
            rb.append(b2); re.append(b2 + tl2); rd.append(len(reads) - 1); rev.append(is_rev)
    return reads, rb, re, rd, rev


def test_matesw_kernel_matches_reference_ksw_align2(engine):
    """msw_kernel == ksw_align2 with mem_matesw's flags (src/bwamem_pair.c:150-177), byte and word flavours."""
    import ctypes as C
    from mpibwa_amd import api
    lib = api.load_library()
    rng = np.random.default_rng(2024)
    l_pac = 40000
    ref = rng.integers(0, 4, size=l_pac).astype(np.uint8)
    ref[5000:5600] = np.tile(ref[5000:5006], 100)   # a tandem repeat
    pac = np.zeros(l_pac // 4 + 1, dtype=np.uint8)
    for k in range(4):
        pac[:l_pac // 4] |= (ref[k::4] << ((3 - k) * 2)).astype(np.uint8)
    opt_p = engine.opt()
    opt = opt_p.contents
    mat = np.frombuffer(bytes(opt.mat), dtype=np.int8).copy()
    reads, rb, re, rd, rev = _matesw_cases(rng, l_pac, ref, 2000, [50, 100, 150, 151, 249, 250, 251, 301])
    assert sum(1 for i in range(1, len(rd)) if rd[i] == rd[i - 1]) > 1000   # (runs of requests for one mate: the two-per-quad kernel's food)
    got, ms = engine.matesw(opt_p, l_pac, pac, reads, rb, re, rd, rev)
    use_ref = po.ref_available()
    if use_ref:
        class kswr_t(C.Structure):
            _fields_ = [(n, C.c_int) for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")]
        rl = po.ref_lib()
        rl.ksw_align2.restype = kswr_t
        rl.ksw_align2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    n_hit = n_second = 0
    dref = np.concatenate([ref, 3 - ref[::-1]]).astype(np.uint8)
    for i in range(len(rb)):
        tl = re[i] - rb[i]
        win = np.ascontiguousarray(dref[rb[i]:re[i]])
        s = reads[rd[i]]
        q = s if not rev[i] else np.where(s[::-1] < 4, 3 - s[::-1], 4).astype(np.uint8)
        q = np.ascontiguousarray(q)
        xtra = 0x40000 | 0x80000 | (0x10000 if len(q) * opt.a < 250 else 0) | (opt.min_seed_len * opt.a)
        want = np.zeros(7, dtype=np.int32)
        lib.mi355x_host_ksw_align2(len(q), q.ctypes.data, tl, win.ctypes.data, mat.ctypes.data, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins,
                                   xtra, 1, want.ctypes.data)
        if use_ref:
            qq, tt = q.copy(), win.copy()
            w = rl.ksw_align2(len(q), qq.ctypes.data, tl, tt.ctypes.data, 5, mat.ctypes.data, opt.o_del, opt.e_del, opt.o_ins, opt.e_ins, xtra, None)
            assert (want == np.array([w.score, w.te, w.qe, w.score2, w.te2, w.tb, w.qb])).all()
        assert got[i, 7] == 0
        g = got[i, :7]
        # below min_seed_len * a the caller drops the result and the reference leaves te/qe of its scan; compare what is used
        if want[0] < opt.min_seed_len * opt.a:
            assert g[0] == want[0] and g[5] == -1 and g[6] == -1, (i, g, want)
        else:
            assert (g == want).all(), (i, len(q), tl, rev[i], g, want)
            n_hit += 1
            n_second += want[3] > 0
    assert n_hit > 1500 and n_second > 50


def test_chain_kernel_matches_host_chaining(engine, genome):
    """The three chaining launches (one tree node / 255 seeds / the B-tree kernel for up to 255 chains) == the host's restatement
    of mem_chain + mem_chain_flt (kbtree rules for equal keys and splits, the introsort, float compares) on adversarial seed sets:
    equal positions, equal weights, up to 9 chains, dozens of chains, > 64 seeds, > 255 seeds (declined), both strands, seeds
    bridging contigs."""
    rng = np.random.default_rng(77)
    opt = engine.opt()
    l_pac = int(engine.bns.contents.l_pac)
    n_seqs = int(engine.bns.contents.n_seqs)
    offs = [int(engine.bns.contents.anns[k].offset) for k in range(n_seqs)] + [l_pac]
    lens, lrep, seedsets = [], [], []
    for it in range(4000):
        lq = int(rng.choice([100, 150, 151, 250]))
        mode = rng.random()
        n_anchor = int(rng.integers(1, 4)) if mode < 0.6 else int(rng.integers(3, 14))
        anchors = []
        for _ in range(n_anchor):
            k = int(rng.integers(0, n_seqs))
            p = int(rng.integers(offs[k], max(offs[k] + 1, offs[k + 1] - lq - 1)))
            if rng.random() < 0.5:
                p = 2 * l_pac - 1 - p - lq      # reverse strand
                p = max(p, l_pac)
            anchors.append(p)
        if rng.random() < 0.15 and len(anchors) > 1:
            anchors[1] = anchors[0]               # two chains anchored at the same position
        if rng.random() < 0.05:
            anchors[0] = offs[int(rng.integers(1, n_seqs))] - 10 if n_seqs > 1 else anchors[0]   # seeds bridging two contigs
        ns = int(rng.integers(1, 30)) if mode < 0.9 else int(rng.integers(60, 90))
        if mode > 0.96:       # dozens of chains, up to and beyond what the B-tree kernel takes
            anchors = anchors + [int(rng.integers(0, 2 * l_pac - 400)) for _ in range(int(rng.integers(15, 120)))]
            ns = int(rng.integers(120, 300))
        sd = []
        for _ in range(ns):
            a = anchors[int(rng.integers(0, len(anchors)))]
            qb = int(rng.integers(0, lq - 19))
            ln = int(rng.integers(19, min(lq - qb, 80) + 1))
            shift = int(rng.choice([0, 0, 0, 1, -1, 3, 120, 20000]))
            sd.append((min(max(a + qb + shift, 0), 2 * l_pac - ln - 1), qb, ln))
        if rng.random() < 0.3 and mode <= 0.96:   # equal weights: several disjoint seeds of the same length
            ln = 25
            sd = [(min(max(anchors[j % len(anchors)] + 30 * j + (0 if j % 2 else 7), 0), 2 * l_pac - ln - 1), (30 * j) % (lq - 25), ln)
                  for j in range(int(rng.integers(3, 9)))]
        # mem_chain visits seeds interval by interval: keep the generated order (arbitrary) — both paths see the same
        lens.append(lq); lrep.append(int(rng.integers(0, lq))); seedsets.append(sd)
    dev = engine.chains(opt, lens, lrep, seedsets, 0)
    host = engine.chains(opt, lens, lrep, seedsets, 1)
    n_dev = n_declined = n_multi = n_big = 0
    for d, h, sd in zip(dev, host, seedsets):
        if d is None:
            n_declined += 1
            # (more than 4096 seeds, or — beyond one tree node — two chains anchored at one reference position: the host's B-tree)
            assert len(sd) > 4096 or len({x[0] for x in sd}) < len(sd), (len(sd), "declined without reason")
            continue
        assert d == h, (sd, d, h)
        n_dev += 1
        n_multi += len(h) >= 3
        n_big += len(sd) > 100
    assert n_dev > 3800 and n_multi > 200 and n_big > 40


@pytest.mark.skipif(not po.chain_inject_available(), reason="oracle/_ref/libchaininj.so not present")
def test_chain_kernel_matches_the_reference_mem_chain(engine, genome):
    """chain_kernel vs the reference's OWN mem_chain + mem_chain_flt (src/bwamem.c:251-385), which oracle/chain_inject.c
    lets run on chosen seed sets: equal positions, equal weights, up to 9 chains, dozens of chains (the B-tree kernel), more than
    255 seeds (declined: the library's host path must get them right), both strands, contig-bridging seeds."""
    rng = np.random.default_rng(78)
    ref = po.RefIndex(genome["prefix"])
    opt, ropt = engine.opt(), ref.opt()
    l_pac = int(engine.bns.contents.l_pac)
    n_seqs = int(engine.bns.contents.n_seqs)
    offs = [int(engine.bns.contents.anns[k].offset) for k in range(n_seqs)] + [l_pac]
    from chain_cases import adversarial_interval_sets, reference_chains
    from chain_cases import repeat_like_interval_sets
    # + reads of high-copy repeats: hundreds to thousands of seeds, nearly as many chains of equal weight (chain_heavy_kernel, both
    # of its LDS footprints; its tree grows to four levels, its kept list to hundreds of chains)
    cases = (adversarial_interval_sets(rng, 3000, l_pac, offs, n_seqs) + repeat_like_interval_sets(rng, 120, l_pac, offs, n_seqs) +
             repeat_like_interval_sets(rng, 24, l_pac, offs, n_seqs, n_copies=(350, 500), n_ivs=(4, 9)))   # (at most max_occ hits per interval)
    lens, seedsets, want = reference_chains(ref, ropt, cases)
    lrep = [0] * len(lens)
    dev = engine.chains(opt, lens, lrep, seedsets, 0)
    host = engine.chains(opt, lens, lrep, seedsets, 1)
    n_dev = n_declined = n_multi = n_big = n_heavy = n_heavy_l = 0
    for d, h, w, sd in zip(dev, host, want, seedsets):
        hh = [(c[0], c[5], c[6]) for c in h]
        assert hh == w, ("host path", len(sd), hh[:3], w[:3])
        if d is None:
            n_declined += 1
            assert len(sd) > 4096 or len({x[0] for x in sd}) < len(sd), (len(sd), "declined without reason")
            continue
        dd = [(c[0], c[5], c[6]) for c in d]
        assert dd == w, ("chain_kernel", len(sd), dd[:3], w[:3])
        n_dev += 1
        n_multi += len(w) >= 3
        n_big += len(sd) > 100
        n_heavy += len(sd) > 255
        n_heavy_l += len(sd) > 1024
    assert n_dev > 2800 and n_multi > 150 and n_big > 40 and n_heavy > 100 and n_heavy_l > 10


def _pack2bit(ref):
    l_pac = len(ref)
    pac = np.zeros(l_pac // 4 + 1, dtype=np.uint8)
    for k in range(4):
        part = ref[k::4]
        pac[:len(part)] |= (part << ((3 - k) * 2)).astype(np.uint8)
    return pac


@pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libbwaref.so not present")
@pytest.mark.parametrize("which,scoring", [(0, None), (1, None), (0, (2, 2, 3, 1, 5, 2)), (0, (1, 9, 2, 1, 2, 1))])
def test_aln_kernel_matches_the_reference_gen_cigar2(engine, which, scoring):
    """aln_kernel (all three instantiations: no-DP, narrow band with hand-off, full size) vs the reference's bwa_gen_cigar2
    under mem_reg2aln's band-doubling loop (src/bwa.c:121-207, src/bwamem.c:1110-1120): score, NM, CIGAR, MD.
    The same-length requests with a band are answered without DP when the ungapped alignment cannot be beaten (aln_kernel.hip);
    the other scoring schemes move that bound (a, b, o_del, e_del, o_ins, e_ins): 2 mismatches / none instead of 3."""
    rng = np.random.default_rng(31 + which + (0 if scoring is None else 7 * scoring[1]))
    l_pac = 60000
    ref_seq = rng.integers(0, 4, size=l_pac).astype(np.uint8)
    ref_seq[7000:7400] = np.tile(ref_seq[7000:7004], 100)        # tandem repeat: gap placement must match
    pac = _pack2bit(ref_seq)
    dbl = np.concatenate([ref_seq, 3 - ref_seq[::-1]]).astype(np.uint8)
    ri = po.RefIndex.__new__(po.RefIndex)
    ri.lib = po.ref_lib()
    opt = engine.opt()
    if scoring is not None:
        o = opt.contents
        o.a, o.b, o.o_del, o.e_del, o.o_ins, o.e_ins = scoring
        engine.lib.bwa_fill_scmat(o.a, o.b, o.mat)
    reads, rb, re, qb, qe, w2, truesc = [], [], [], [], [], [], []
    for i in range(1500):
        lq = int(rng.choice([30, 76, 100, 150, 151, 250]))
        p = int(rng.integers(0, 2 * l_pac - 2 * lq - 80))
        if p < l_pac < p + 2 * lq + 80:
            p = l_pac + 5
        kind = rng.random()
        t = dbl[p:p + lq + 70].copy()
        q = t[:lq].copy()
        rl = lq
        if kind < 0.35:
            pass                                                      # same length: the no-DP shortcut when w2 == 0
        elif kind < 0.7:                                              # one gap
            g = int(rng.choice([1, 2, 3, 8, 20, 45]))
            at = int(rng.integers(5, lq - 5))
            if rng.random() < 0.5:
                q = np.concatenate([t[:at], t[at + g:lq + g]])       # deletion from the read
                rl = lq + g
            else:
                q = np.concatenate([t[:at], rng.integers(0, 4, size=g).astype(np.uint8), t[at:lq - g]])[:lq]
                rl = lq - g
        else:                                                         # two gaps of opposite sign
            at1, at2 = sorted(int(x) for x in rng.integers(5, lq - 5, size=2))
            g = int(rng.integers(1, 6))
            q = np.concatenate([t[:at1], t[at1 + g:at2], rng.integers(0, 4, size=g).astype(np.uint8), t[at2:lq]])[:lq]
        mm = rng.random(len(q)) < float(rng.choice([0, 0.01, 0.02, 0.05]))
        q[mm] = (q[mm] + 1) & 3
        if rng.random() < 0.05:
            q[int(rng.integers(0, len(q)))] = 4
        b, e = 0, len(q)
        if rng.random() < 0.3:                                        # a clipped region of the read
            b = int(rng.integers(0, 10)); e = len(q) - int(rng.integers(0, 10))
        reads.append(q); rb.append(p + b); re.append(p + b + max(rl - b - (len(q) - e), 1)); qb.append(b); qe.append(e)
        w2.append(int(rng.choice([0, 0, 1, 3, 8, 15, 40, 100, 400])) if rl != lq or rng.random() < 0.5 else 0)
        truesc.append(int(rng.choice([0, lq, lq - 10])))
    hdr, cigs, mds, ms = engine.global_align(opt, l_pac, pac, reads, rb, re, np.arange(len(reads)), qb, qe, w2, truesc, which=which, cigar_cap=128, md_cap=800)
    n_dp = n_declined = 0
    for i in range(len(reads)):
        sc, cig, nm, md = ri.reg2aln_loop(opt, l_pac, pac, reads[i][qb[i]:qe[i]], rb[i], re[i], w2[i], truesc[i])
        if hdr[i, 4] != 0:     # band matrix beyond the LDS budget of the full-size instantiation: left to the library's host code
            n_declined += 1
            # (with cheap gaps and a 9-point mismatch an early, narrow round can need more CIGAR operations than the kernel
            # keeps — 75 mismatches become 75 insertion + deletion pairs — although the last round needs nine: also declined)
            if scoring is None:
                assert (w2[i] >= 40 or abs((re[i] - rb[i]) - (qe[i] - qb[i])) >= 20) and qe[i] - qb[i] >= 140, (i, hdr[i], w2[i])
            continue
        assert cig is not None
        assert (hdr[i, 0], hdr[i, 1]) == (sc, nm), (i, hdr[i], sc, nm, w2[i])
        assert (cigs[i] == cig).all() and mds[i] == md, (i, cigs[i], cig, mds[i], md)
        n_dp += len(cig) > 1
    assert n_dp > 500 and n_declined < (120 if scoring is None else 300)


def test_aln_kernel_on_golden_ksw_global2_vectors(engine):
    """The committed glo_* tuples (reference ksw_global2 outputs, src/ksw.c:504-606) through aln_kernel: the tuples whose band
    equals what bwa_gen_cigar2 would choose for them (src/bwa.c:152-161) are reproduced exactly, score and CIGAR."""
    from golden_util import kernel_vectors, ragged
    kv = kernel_vectors()
    qs, ts, ws, ss, cs = ragged(kv, "glo_q"), ragged(kv, "glo_t"), kv["glo_w"], kv["glo_s"], ragged(kv, "glo_c")
    opt = engine.opt()
    o = opt.contents
    toff = np.zeros(len(ts) + 1, dtype=np.int64)
    toff[1:] = np.cumsum([len(t) for t in ts])
    genome = np.concatenate(ts).astype(np.uint8)
    l_pac = len(genome)
    pac = _pack2bit(genome)
    sel = []
    for i, (q, t, w) in enumerate(zip(qs, ts, ws)):
        lq, d = len(q), abs(len(t) - len(q))
        max_gap = max(int(((lq + 1) >> 1) * o.a - o.o_ins) // o.e_ins + 1, 1)
        if lq == len(t) and w == 0:
            continue
        if ((max_gap + d + 1) >> 1) >= w >= d + 3 and w <= (o.w << 2):
            sel.append(i)
    assert len(sel) >= 40
    for which in (0, 1):
        hdr, cigs, mds, ms = engine.global_align(opt, l_pac, pac, [qs[i] for i in sel], [toff[i] for i in sel], [toff[i + 1] for i in sel],
                                                 np.arange(len(sel)), [0] * len(sel), [len(qs[i]) for i in sel], [int(ws[i]) for i in sel],
                                                 [-(1 << 20)] * len(sel), which=which)
        for k, i in enumerate(sel):
            assert hdr[k, 4] == 0 and hdr[k, 0] == ss[i], (i, hdr[k], ss[i])
            assert (cigs[k] == cs[i]).all(), (i, cigs[k], cs[i])
