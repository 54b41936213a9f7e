"""TEST INFRASTRUCTURE — ctypes access to (a) the plain-C oracle restatement
(oracle/liboracle.so) and (b) the real reference hot path compiled from the
reference's own sources (oracle/_ref/libbwaref.so, see oracle/Makefile).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product never does.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from mpibwa_amd import abi  # noqa: E402

libc = C.CDLL(None if os.environ.get("MPIBWA_SANITIZER_LIB") else "libc.so.6")   # (under tools/san_host.sh: the sanitizer's malloc / free)
libc.free.argtypes = [C.c_void_p]
libc.calloc.restype = C.c_void_p
libc.calloc.argtypes = [C.c_size_t, C.c_size_t]


def build(verbose=False):
    """make liboracle.so (+ _ref when /root/reference is present)."""
    r = subprocess.run(["make", "-C", HERE, "all"], capture_output=True, text=True)
    if r.returncode != 0 or verbose:
        sys.stderr.write(r.stdout + r.stderr)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed")


class orc_fm_t(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("seq_len", C.c_uint64),
                ("bwt", C.c_void_p), ("sa", C.c_void_p), ("sa_intv", C.c_int),
                ("n_extend", C.c_uint64), ("n_blocks", C.c_uint64), ("n_sa_steps", C.c_uint64),
                ("n_sa_calls", C.c_uint64)]


class orc_intv_t(C.Structure):
    _fields_ = [("x", C.c_uint64 * 3), ("info", C.c_uint64)]


_orc = None


def oracle_lib():
    global _orc
    if _orc is None:
        p = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(p):
            build()
        _orc = C.CDLL(p)
        _orc.orc_sa.restype = C.c_uint64
        _orc.orc_sa.argtypes = [C.POINTER(orc_fm_t), C.c_uint64]
        _orc.orc_occ.restype = C.c_uint64
        _orc.orc_collect_intv.restype = C.c_int
        _orc.orc_collect_intv.argtypes = [C.POINTER(orc_fm_t), C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int,
                                          C.c_uint64, C.POINTER(C.POINTER(orc_intv_t))]
        _orc.orc_extend2.restype = C.c_int64
        _orc.orc_extend2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]
        _orc.orc_global2.restype = C.c_int
        _orc.orc_global2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 5 + \
                                    [C.POINTER(C.c_int), C.c_void_p]
    return _orc


class OracleFM:
    """FM-index view for the oracle, built from the on-disk bwa files."""

    def __init__(self, prefix):
        raw = np.fromfile(prefix + ".bwt", dtype=np.uint8)
        hdr = raw[:40].view(np.uint64)
        self.bwt = np.ascontiguousarray(raw[40:]).view(np.uint32)
        sraw = np.fromfile(prefix + ".sa", dtype=np.uint64)
        self.sa = np.concatenate([np.array([np.uint64(0xFFFFFFFFFFFFFFFF)]), sraw[7:]])
        self.fm = orc_fm_t()
        self.fm.primary = int(hdr[0])
        for i in range(4):
            self.fm.L2[i + 1] = int(hdr[1 + i])
        self.fm.seq_len = int(hdr[4])
        self.fm.bwt = self.bwt.ctypes.data
        self.fm.sa = self.sa.ctypes.data
        self.fm.sa_intv = int(sraw[5])
        self.lib = oracle_lib()

    def reset_counters(self):
        self.fm.n_extend = self.fm.n_blocks = self.fm.n_sa_steps = self.fm.n_sa_calls = 0

    def collect_intv(self, seq, min_seed_len=19, split_factor=1.5, split_width=10, max_mem_intv=20):
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        out = C.POINTER(orc_intv_t)()
        n = self.lib.orc_collect_intv(C.byref(self.fm), len(seq), seq.ctypes.data, min_seed_len, split_factor,
                                      split_width, max_mem_intv, C.byref(out))
        res = np.zeros((n, 4), dtype=np.uint64)
        if n:
            res[:] = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint64)), shape=(n, 4))
        libc.free(C.cast(out, C.c_void_p))
        return res

    def sa_lookup(self, k):
        return int(self.lib.orc_sa(C.byref(self.fm), int(k)))


def oracle_extend2(q, t, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    m = np.ascontiguousarray(mat, dtype=np.int8)
    out = np.zeros(6, dtype=np.int32)
    cells = oracle_lib().orc_extend2(len(q), q.ctypes.data, len(t), t.ctypes.data, m.ctypes.data, o_del, e_del, o_ins,
                                     e_ins, w, end_bonus, zdrop, h0, out.ctypes.data)
    return out, int(cells)


def oracle_global2(q, t, mat, o_del, e_del, o_ins, e_ins, w):
    q = np.ascontiguousarray(q, dtype=np.uint8)
    t = np.ascontiguousarray(t, dtype=np.uint8)
    m = np.ascontiguousarray(mat, dtype=np.int8)
    cig = np.zeros(len(q) + len(t) + 2, dtype=np.uint32)
    n = C.c_int(0)
    sc = oracle_lib().orc_global2(len(q), q.ctypes.data, len(t), t.ctypes.data, m.ctypes.data, o_del, e_del, o_ins,
                                  e_ins, w, C.byref(n), cig.ctypes.data)
    return sc, cig[:n.value].copy()


# ---------------------------------------------------------------------------
# the real reference (oracle/_ref/libbwaref.so)
# ---------------------------------------------------------------------------
_ref = None


def ref_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libbwaref.so"))


def ref_lib():
    global _ref
    if _ref is None:
        p = os.path.join(HERE, "_ref", "libbwaref.so")
        if not os.path.exists(p):
            build()
        _ref = C.CDLL(p)
        _ref.bwa_idx_load_from_disk.restype = C.POINTER(abi.bwaidx_t)
        _ref.bwa_idx_load_from_disk.argtypes = [C.c_char_p, C.c_int]
        _ref.mem_opt_init.restype = C.POINTER(abi.mem_opt_t)
        _ref.mem_process_seqs.restype = None
        _ref.mem_process_seqs.argtypes = [C.POINTER(abi.mem_opt_t), C.POINTER(abi.bwt_t), C.POINTER(abi.bntseq_t),
                                          C.POINTER(C.c_uint8), C.c_int64, C.c_int, C.POINTER(abi.bseq1_t),
                                          C.POINTER(abi.mem_pestat_t)]
        _ref.bwt_sa.restype = C.c_uint64
        _ref.bwt_sa.argtypes = [C.POINTER(abi.bwt_t), C.c_uint64]
        _ref.ksw_extend2.restype = C.c_int
        _ref.ksw_extend2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 8 + \
                                    [C.POINTER(C.c_int)] * 5
        _ref.ksw_global2.restype = C.c_int
        _ref.ksw_global2.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 5 + \
                                    [C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint32))]
        _ref.mem_chain.argtypes = [C.POINTER(abi.mem_opt_t), C.POINTER(abi.bwt_t), C.POINTER(abi.bntseq_t), C.c_int,
                                   C.c_void_p, C.c_void_p]
    return _ref


class _intv_v(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


class _smem_aux(C.Structure):  # smem_aux_t, src/bwamem.c:93-95
    _fields_ = [("mem", _intv_v), ("mem1", _intv_v), ("tmpv", C.c_void_p * 2)]


class _chain_v(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("a", C.c_void_p)]


class RefIndex:
    def __init__(self, prefix):
        self.lib = ref_lib()
        self.idx = self.lib.bwa_idx_load_from_disk(prefix.encode(), 7)
        self.bwt = self.idx.contents.bwt
        self.bns = self.idx.contents.bns
        self.pac = self.idx.contents.pac

    def opt(self, **kw):
        o = self.lib.mem_opt_init()
        for k, v in kw.items():
            setattr(o.contents, k, v)
        return o

    def collect_intv(self, opt, seq):
        """SA intervals of mem_collect_intv (static in the reference) observed through mem_chain()'s aux buffer."""
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        aux = _smem_aux()
        tv = (_intv_v * 2)()
        aux.tmpv[0] = C.addressof(tv[0])
        aux.tmpv[1] = C.addressof(tv[1])
        self.lib.mem_chain.restype = _chain_v
        ch = self.lib.mem_chain(opt, self.bwt, self.bns, len(seq), seq.ctypes.data, C.addressof(aux))
        n = aux.mem.n
        res = np.zeros((n, 4), dtype=np.uint64)
        if n:
            res[:] = np.ctypeslib.as_array(C.cast(aux.mem.a, C.POINTER(C.c_uint64)), shape=(n, 4))
        # leak the small chain arrays (test process only)
        return res

    def sa_lookup(self, k):
        return int(self.lib.bwt_sa(self.bwt, int(k)))

    def extend2(self, q, t, mat, o_del, e_del, o_ins, e_ins, w, end_bonus, zdrop, h0):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        m = np.ascontiguousarray(mat, dtype=np.int8)
        v = [C.c_int(0) for _ in range(5)]
        sc = self.lib.ksw_extend2(len(q), q.ctypes.data, len(t), t.ctypes.data, 5, m.ctypes.data, o_del, e_del, o_ins,
                                  e_ins, w, end_bonus, zdrop, h0, *[C.byref(x) for x in v])
        return np.array([sc] + [x.value for x in v], dtype=np.int32)

    def global2(self, q, t, mat, o_del, e_del, o_ins, e_ins, w):
        q = np.ascontiguousarray(q, dtype=np.uint8)
        t = np.ascontiguousarray(t, dtype=np.uint8)
        m = np.ascontiguousarray(mat, dtype=np.int8)
        n = C.c_int(0)
        cig = C.POINTER(C.c_uint32)()
        sc = self.lib.ksw_global2(len(q), q.ctypes.data, len(t), t.ctypes.data, 5, m.ctypes.data, o_del, e_del, o_ins,
                                  e_ins, w, C.byref(n), C.byref(cig))
        out = np.array([cig[i] for i in range(n.value)], dtype=np.uint32)
        libc.free(C.cast(cig, C.c_void_p))
        return sc, out

    def process(self, opt, reads, n_processed=0, pes0=None, with_qual=True, comment=None):
        """Run the reference mem_process_seqs on [(name, seq1 bytes, seq2 bytes|None)] → list of SAM byte strings."""
        batch = abi.SeqBatch(libc, reads, with_qual=with_qual, comment=comment)
        self.lib.mem_process_seqs(opt, self.bwt, self.bns, self.pac, n_processed, batch.n, batch.arr, pes0)
        return batch.take_sam()

    def gen_cigar2(self, opt, l_pac, pac, query, rb, re, w):
        """The reference's bwa_gen_cigar2 (src/bwa.c:121-207) on a window of a packed reference: (score, cigar u32[], NM, MD bytes);
        cigar is None when the reference rejects the request."""
        o = opt.contents
        q = np.ascontiguousarray(query, dtype=np.uint8).copy()
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        mat = (C.c_int8 * 25)(*o.mat)
        f = self.lib.bwa_gen_cigar2
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p] + [C.c_int] * 5 + [C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64] + [C.POINTER(C.c_int)] * 3
        sc, n, nm = C.c_int(0), C.c_int(0), C.c_int(0)
        p = f(mat, o.o_del, o.e_del, o.o_ins, o.e_ins, int(w), int(l_pac), pac.ctypes.data, len(q), q.ctypes.data, int(rb), int(re),
              C.byref(sc), C.byref(n), C.byref(nm))
        if not p:
            return sc.value, None, nm.value, b""
        cig = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()
        md = C.string_at(p + 4 * n.value)
        libc.free(C.c_void_p(p))
        return sc.value, cig, nm.value, md

    def reg2aln_loop(self, opt, l_pac, pac, query, rb, re, w2, truesc):
        """mem_reg2aln's band-doubling loop (src/bwamem.c:1110-1120) around the reference's bwa_gen_cigar2: the result of its
        last round."""
        o = opt.contents
        last, i = -(1 << 30), 0
        while True:
            w2 = min(w2, o.w << 2)
            res = self.gen_cigar2(opt, l_pac, pac, query, rb, re, w2)
            score = res[0]
            if score == last or w2 == o.w << 2:
                break
            last = score
            w2 <<= 1
            i += 1
            if not (i < 3 and score < truesc - o.a):
                break
        return res


_inj = None


def chain_inject_lib():
    """oracle/_ref/libchaininj.so: the reference's mem_chain / mem_chain_flt fed with test-chosen seeds (oracle/chain_inject.c)"""
    global _inj
    if _inj is None:
        ref_lib()
        _inj = C.CDLL(os.path.join(HERE, "_ref", "libchaininj.so"))
        _inj.inj_chain.restype = C.c_int64
        _inj.inj_chain.argtypes = [C.POINTER(abi.mem_opt_t), C.POINTER(abi.bntseq_t), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_int64]
    return _inj


_pinj = None

# mem_alnreg_t (src/bwamem.h:59-77), 88 bytes
ALNREG_DT = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("score", "<i4"), ("truesc", "<i4"), ("sub", "<i4"),
                      ("alt_sc", "<i4"), ("csub", "<i4"), ("sub_n", "<i4"), ("w", "<i4"), ("seedcov", "<i4"), ("secondary", "<i4"),
                      ("secondary_all", "<i4"), ("seedlen0", "<i4"), ("n_comp_is_alt", "<i4"), ("frac_rep", "<f4"), ("hash", "<u8")])
assert ALNREG_DT.itemsize == 88


def pair_inject_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libpairinj.so"))


def ref_pair(opt, bns, pac, pes, pair_id, l_seq, regs0, regs1):
    """The reference's own mem_sam_pe (oracle/pair_inject.c) on the two region lists (arrays of ALNREG_DT, as mem_chain2aln leaves them:
    its mem_sort_dedup_patch runs first).  -> dict: paired (its paired branch wrote the records), n_align (alignments mem_matesw asked
    for), n_xa per end (hits that got an XA entry), and per reported line the region, flag and MAPQ."""
    global _pinj
    if _pinj is None:
        ref_lib()
        _pinj = C.CDLL(os.path.join(HERE, "_ref", "libpairinj.so"))
        _pinj.inj_sam_pe.restype = C.c_int
        _pinj.inj_sam_pe.argtypes = [C.POINTER(abi.mem_opt_t), C.POINTER(abi.bntseq_t), C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_int, C.c_void_p]
    regs = np.concatenate([np.ascontiguousarray(regs0, dtype=ALNREG_DT), np.ascontiguousarray(regs1, dtype=ALNREG_DT)])
    out = np.zeros(64, dtype=np.int64)
    _pinj.inj_sam_pe(opt, bns, C.cast(pac, C.c_void_p), C.cast(pes, C.c_void_p), int(pair_id), int(l_seq), len(regs0), len(regs1), regs.ctypes.data, 1, out.ctypes.data)
    r = {"n_reg2aln": int(out[0]), "n_lines": int(out[1]), "n_align": int(out[2]), "paired": int(out[3]) == 0, "n_xa": (int(out[4]), int(out[5])), "lines": []}
    for k in range(min(int(out[0]), 4)):
        v = out[6 + 11 * k:6 + 11 * (k + 1)]
        r["lines"].append(dict(rb=int(v[0]), re=int(v[1]), qb=int(v[2]), qe=int(v[3]), score=int(v[4]), sub=int(v[5]), secondary=int(v[6]), truesc=int(v[7]),
                               w=int(v[8]), flag=int(v[9]), mapq=int(v[10])))
    return r


def chain_inject_available():
    return os.path.exists(os.path.join(HERE, "_ref", "libchaininj.so"))


def ref_chains(opt, bns, read_len, intervals, do_flt=True):
    """intervals: [(qbeg, qend, [rbeg, ...])] with distinct (qbeg, qend) -> the reference's chains after mem_chain (+ mem_chain_flt):
    [(rid, w, kept, is_alt, frac_rep_bits, [(rbeg, qbeg, len), ...])]"""
    lib = chain_inject_lib()
    n = len(intervals)
    iv = np.zeros((max(n, 1), 4), dtype=np.uint64)
    sa = []
    for k, (qb, qe, rbs) in enumerate(intervals):
        iv[k] = (len(sa), 0, len(rbs), (qb << 32) | qe)
        sa += list(rbs)
    sa = np.array(sa if sa else [0], dtype=np.int64)
    cap = 16 + 8 * n + 12 * len(sa) + 64   # (a chain of its own per hit: 6 + 3 numbers each)
    out = np.zeros(cap, dtype=np.int64)
    got = lib.inj_chain(opt, bns, int(read_len), n, iv.ctypes.data, sa.ctypes.data, 1 if do_flt else 0, out.ctypes.data, cap)
    assert got >= 0
    p, res = 1, []
    for _ in range(int(out[0])):
        rid, ns, w, kept, is_alt, fb = (int(x) for x in out[p:p + 6])
        p += 6
        res.append((rid, w, kept, is_alt, fb, [tuple(int(x) for x in out[p + 3 * j:p + 3 * j + 3]) for j in range(ns)]))
        p += 3 * ns
    return res
