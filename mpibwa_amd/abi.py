"""ctypes mirror of include/mpibwa_amd.h (the reference's x86-64 struct layouts:
src/bwamem.h:25-85, src/bwa.h:20-33, src/bwt.h:46-62, src/bntseq.h:41-64)."""
import ctypes as C

MEM_F_PE = 0x2
MEM_F_NOPAIRING = 0x4
MEM_F_ALL = 0x8
MEM_F_NO_MULTI = 0x10
MEM_F_NO_RESCUE = 0x20
MEM_F_REF_HDR = 0x100
MEM_F_SOFTCLIP = 0x200
MEM_F_SMARTPE = 0x400
MEM_F_PRIMARY5 = 0x800
MEM_F_KEEP_SUPP_MAPQ = 0x1000


class mem_opt_t(C.Structure):
    _fields_ = [
        ("a", C.c_int), ("b", C.c_int),
        ("o_del", C.c_int), ("e_del", C.c_int),
        ("o_ins", C.c_int), ("e_ins", C.c_int),
        ("pen_unpaired", C.c_int),
        ("pen_clip5", C.c_int), ("pen_clip3", C.c_int),
        ("w", C.c_int),
        ("zdrop", C.c_int),
        ("max_mem_intv", C.c_uint64),
        ("T", C.c_int),
        ("flag", C.c_int),
        ("min_seed_len", C.c_int),
        ("min_chain_weight", C.c_int),
        ("max_chain_extend", C.c_int),
        ("split_factor", C.c_float),
        ("split_width", C.c_int),
        ("max_occ", C.c_int),
        ("max_chain_gap", C.c_int),
        ("n_threads", C.c_int),
        ("chunk_size", C.c_int),
        ("mask_level", C.c_float),
        ("drop_ratio", C.c_float),
        ("XA_drop_ratio", C.c_float),
        ("mask_level_redun", C.c_float),
        ("mapQ_coef_len", C.c_float),
        ("mapQ_coef_fac", C.c_int),
        ("max_ins", C.c_int),
        ("max_matesw", C.c_int),
        ("max_XA_hits", C.c_int), ("max_XA_hits_alt", C.c_int),
        ("mat", C.c_int8 * 25),
    ]


class mem_pestat_t(C.Structure):
    _fields_ = [("low", C.c_int), ("high", C.c_int), ("failed", C.c_int), ("avg", C.c_double), ("std", C.c_double)]


class bseq1_t(C.Structure):
    _fields_ = [("l_seq", C.c_int), ("id", C.c_int), ("name", C.c_void_p), ("comment", C.c_void_p),
                ("seq", C.c_void_p), ("qual", C.c_void_p), ("sam", C.c_void_p)]


class bwt_t(C.Structure):
    _fields_ = [("primary", C.c_uint64), ("L2", C.c_uint64 * 5), ("seq_len", C.c_uint64), ("bwt_size", C.c_uint64),
                ("bwt", C.POINTER(C.c_uint32)), ("cnt_table", C.c_uint32 * 256), ("sa_intv", C.c_int),
                ("n_sa", C.c_uint64), ("sa", C.POINTER(C.c_uint64))]


class bntann1_t(C.Structure):
    _fields_ = [("offset", C.c_int64), ("len", C.c_int32), ("n_ambs", C.c_int32), ("gi", C.c_uint32),
                ("is_alt", C.c_int32), ("name", C.c_char_p), ("anno", C.c_char_p)]


class bntamb1_t(C.Structure):
    _fields_ = [("offset", C.c_int64), ("len", C.c_int32), ("amb", C.c_char)]


class bntseq_t(C.Structure):
    _fields_ = [("l_pac", C.c_int64), ("n_seqs", C.c_int32), ("seed", C.c_uint32), ("anns", C.POINTER(bntann1_t)),
                ("n_holes", C.c_int32), ("ambs", C.POINTER(bntamb1_t)), ("fp_pac", C.c_void_p)]


class bwaidx_t(C.Structure):
    _fields_ = [("bwt", C.POINTER(bwt_t)), ("bns", C.POINTER(bntseq_t)), ("pac", C.POINTER(C.c_uint8)),
                ("is_shm", C.c_int), ("l_mem", C.c_int64), ("mem", C.POINTER(C.c_uint8))]


class mi355x_stats_t(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("total_ms", "h2d_ms", "smem_ms", "sa_ms", "chain_ms", "ext_ms", "regs_ms",
                                          "pestat_ms", "sam_ms", "k_smem_ms", "k_sa_ms", "k_ext_ms")] + \
               [(n, C.c_uint64) for n in ("smem_bytes", "sa_bytes", "ext_cells", "n_reads", "n_intv", "n_seeds",
                                          "n_chains", "n_ext")] + \
               [(n, C.c_double) for n in ("plan_ms", "aln_ms", "k_aln_ms")] + [("n_aln", C.c_uint64), ("phase1_ms", C.c_double), ("msw_ms", C.c_double), ("k_msw_ms", C.c_double), ("n_msw", C.c_uint64), ("emit_ms", C.c_double), ("n_sub", C.c_uint64), ("smem_tab_bytes", C.c_uint64), ("n_sam_dev", C.c_uint64), ("n_pair_dev", C.c_uint64)]


assert C.sizeof(mem_opt_t) == 168
assert C.sizeof(mem_pestat_t) == 32
assert C.sizeof(bseq1_t) == 48
assert C.sizeof(bwt_t) == 1120
assert C.sizeof(bntann1_t) == 40
assert C.sizeof(bntamb1_t) == 16
assert C.sizeof(bntseq_t) == 48


class SeqBatch:
    """Owns the C buffers behind a bseq1_t[n] array, built the way mpiBWA's main does
    (src/mainParallel.c:1257-1301): one chunk buffer, NUL-terminated name/seq/qual strings pointing into it,
    mates interleaved."""

    _DT = None

    @staticmethod
    def dtype():
        """numpy view of bseq1_t (src/bwa.h:30-33)"""
        import numpy as np
        if SeqBatch._DT is None:
            SeqBatch._DT = np.dtype([("l_seq", "<i4"), ("id", "<i4"), ("name", "<u8"), ("comment", "<u8"), ("seq", "<u8"),
                                     ("qual", "<u8"), ("sam", "<u8")])
        return SeqBatch._DT

    def __init__(self, libc, reads, with_qual=True, comment=None):
        # reads: list of (name:str, seq1:bytes, seq2:bytes|None)
        import numpy as np
        self.libc = libc
        names, seqs = [], []
        for name, s1, s2 in reads:
            nb = name.encode() if isinstance(name, str) else name
            names.append(nb); seqs.append(bytes(s1))
            if s2 is not None:
                names.append(nb); seqs.append(bytes(s2))
        n = self.n = len(seqs)
        cb = comment.encode() if comment is not None else None
        parts, lens = [], []
        for nb, sq in zip(names, seqs):
            parts += [nb, b"\0", sq, b"\0"]
            lens += [len(nb) + 1, len(sq) + 1]
            if with_qual:
                parts += [b"I" * len(sq), b"\0"]
            if cb is not None:
                parts += [cb, b"\0"]
        blob = b"".join(parts)
        self._arena = C.create_string_buffer(blob, len(blob) + 1)
        base = C.addressof(self._arena)
        if SeqBatch._DT is None:
            SeqBatch._DT = np.dtype([("l_seq", "<i4"), ("id", "<i4"), ("name", "<u8"), ("comment", "<u8"), ("seq", "<u8"),
                                     ("qual", "<u8"), ("sam", "<u8")])
        rec = np.zeros(max(n, 1), dtype=SeqBatch._DT)
        if n:
            ln = np.array([len(x) for x in names], dtype=np.int64) + 1
            ls = np.array([len(x) for x in seqs], dtype=np.int64) + 1
            per = ln + ls + (ls if with_qual else 0) + ((len(cb) + 1) if cb is not None else 0)
            start = np.concatenate([[0], np.cumsum(per)[:-1]]) + base
            rec["l_seq"][:n] = ls - 1
            rec["name"][:n] = start
            rec["seq"][:n] = start + ln
            if with_qual:
                rec["qual"][:n] = start + ln + ls
            if cb is not None:
                rec["comment"][:n] = start + ln + ls + (ls if with_qual else 0)
        self._rec = rec
        self.arr = C.cast(rec.ctypes.data, C.POINTER(bseq1_t))

    def take_sam(self):
        """Collect seqs[i].sam strings and free() them as the caller in mainParallel.c:1390 does."""
        out = []
        sam = self._rec["sam"]
        for i in range(self.n):
            p = int(sam[i])
            if p:
                out.append(C.string_at(p))
                self.libc.free(C.c_void_p(p))
                sam[i] = 0
            else:
                out.append(b"")
        return out
