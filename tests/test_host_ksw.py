"""Host logic (no GPU): the product's ksw_align2 (SSE2 and lane-by-lane forms) against the compiled reference's."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as po

MAT = np.array([1, -4, -4, -4, -1, -4, 1, -4, -4, -1, -4, -4, 1, -4, -1, -4, -4, -4, 1, -1, -1, -1, -1, -1, -1], dtype=np.int8)
XBYTE, XSTOP, XSUBO, XSTART = 0x10000, 0x20000, 0x40000, 0x80000


class kswr_t(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("score", "te", "qe", "score2", "te2", "tb", "qb")]


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    for it in range(n):
        ql = int(rng.choice([20, 75, 150, 151, 250, int(rng.integers(19, 260))]))
        tl = int(rng.integers(ql, ql + 700))
        t = rng.integers(0, 4, size=tl, dtype=np.uint8)
        q = rng.integers(0, 4, size=ql, dtype=np.uint8)
        mode = rng.random()
        if mode < 0.7:   # plant the query (with noise) somewhere in the target, sometimes twice
            for rep in range(1 + (rng.random() < 0.3)):
                p = int(rng.integers(0, tl - ql + 1))
                m = q.copy()
                mut = rng.random(ql) < rng.choice([0.0, 0.02, 0.1])
                m[mut] = rng.integers(0, 4, size=int(mut.sum()))
                t[p:p + ql] = m
        if rng.random() < 0.1:
            q[rng.integers(0, ql)] = 4
        byte = XBYTE if ql < 250 else 0
        xtra = XSUBO | XSTART | byte | 19
        if rng.random() < 0.15:
            xtra = XSTART | (byte if rng.random() < 0.5 else 0)
        yield q, t, xtra


@pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libbwaref.so not built")
def test_align2_matches_reference(built):
    from mpibwa_amd import api
    lib = api.load_library()
    ref = po.ref_lib()
    ref.ksw_align2.restype = kswr_t
    ref.ksw_align2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
    n = 0
    for q, t, xtra in _cases(700, 31):
        qq, tt = q.copy(), t.copy()
        w = ref.ksw_align2(len(q), qq.ctypes.data, len(t), tt.ctypes.data, 5, MAT.ctypes.data, 6, 1, 6, 1, xtra, None)
        want = np.array([w.score, w.te, w.qe, w.score2, w.te2, w.tb, w.qb])
        for portable in (0, 1):
            out = np.zeros(7, dtype=np.int32)
            lib.mi355x_host_ksw_align2(len(q), q.ctypes.data, len(t), t.ctypes.data, MAT.ctypes.data, 6, 1, 6, 1, xtra, portable,
                                       out.ctypes.data)
            assert (out == want).all(), (portable, len(q), len(t), hex(xtra), out, want)
        n += 1
    assert n == 700


def test_align2_sse2_equals_portable(built):
    from mpibwa_amd import api
    lib = api.load_library()
    for q, t, xtra in _cases(300, 77):
        a, b = np.zeros(7, dtype=np.int32), np.zeros(7, dtype=np.int32)
        lib.mi355x_host_ksw_align2(len(q), q.ctypes.data, len(t), t.ctypes.data, MAT.ctypes.data, 6, 1, 6, 1, xtra, 0, a.ctypes.data)
        lib.mi355x_host_ksw_align2(len(q), q.ctypes.data, len(t), t.ctypes.data, MAT.ctypes.data, 6, 1, 6, 1, xtra, 1, b.ctypes.data)
        assert (a == b).all()
