"""ctypes binding of libmpibwa_amd.so — the host-side mirror of the reference's
operator interface (mem_opt_init / bwa_idx_load / mem_process_seqs)."""
import ctypes as C
import os

import numpy as np

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

libc = C.CDLL(None if os.environ.get("MPIBWA_SANITIZER_LIB") else "libc.so.6")   # (under tools/san_host.sh: the sanitizer's malloc / free)
libc.free.argtypes = [C.c_void_p]

EXPORTS = [
    "mem_process_seqs", "mem_opt_init", "bwa_fill_scmat", "bwa_idx_load_from_disk", "bwa_mem2idx", "bwa_idx_destroy",
    "mi355x_index_upload", "mi355x_index_alloc", "mi355x_index_buffers", "mi355x_index_d2d", "mi355x_index_commit",
    "mi355x_finalize", "mi355x_index_build", "mi355x_index_build_gpu",
    "mi355x_smem_batch", "mi355x_sa_batch", "mi355x_sa_batch2", "mi355x_sa_dense_info", "mi355x_extend_batch", "mi355x_matesw_batch", "mi355x_chain_batch", "mi355x_pair_batch", "mi355x_pair_maxreg", "mi355x_fastq_scan", "mi355x_fastq_chunks", "mi355x_fastq_fill", "mi355x_last_stats", "mi355x_host_cpus", "mi355x_collect_sam", "mi355x_collect_sam_into", "mi355x_host_ksw_align2",
    "bwa_set_rg", "bwa_insert_header", "bwa_idx2mem", "mi355x_write_map", "mi355x_init", "mi355x_rank_host_threads", "mi355x_index_checksums", "mi355x_init_bcast_seconds", "mi355x_global_batch", "mi355x_device_count", "mi355x_device_memory", "mi355x_buffer_growths", "mi355x_prewarm", "mi355x_max_calls",
]


class mi355x_comm_t(C.Structure):
    """include/mpibwa_amd.h: the caller's transport for the RCCL bootstrap id"""
    BCAST = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
    _fields_ = [("rank", C.c_int), ("size", C.c_int), ("bcast", BCAST), ("user", C.c_void_p)]



def load_library(build_if_missing=True):
    """Load the product library.  Fails loudly if it is absent and cannot be built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(HERE, "libmpibwa_amd.so")
    # tools/san_host.sh: the library's HOST sources alone, built with AddressSanitizer / UBSan — no kernels and none of the entry points
    # that need the GPU; only the host-logic tests run on it
    host_only = os.environ.get("MPIBWA_SANITIZER_LIB")
    if host_only:
        path = host_only
    if not os.path.exists(path):
        if not build_if_missing or host_only:
            raise RuntimeError("%s is missing: run python -m mpibwa_amd.build" % path)
        from .build import build
        build()
    lib = C.CDLL(path)
    P = C.POINTER

    def sig(name, restype, argtypes):
        try:
            fn = getattr(lib, name)  # AttributeError here = the library does not export what include/mpibwa_amd.h declares
        except AttributeError:
            if host_only:
                return
            raise
        fn.restype = restype
        fn.argtypes = argtypes

    sig("mem_opt_init", P(abi.mem_opt_t), [])
    sig("bwa_idx_load_from_disk", P(abi.bwaidx_t), [C.c_char_p, C.c_int])
    sig("mi355x_index_build", C.c_int, [C.c_char_p, C.c_char_p])
    sig("mi355x_index_build_gpu", C.c_int, [C.c_int, C.c_void_p, C.c_int64, C.c_char_p, P(C.c_double)])
    sig("mem_process_seqs", None, [P(abi.mem_opt_t), P(abi.bwt_t), P(abi.bntseq_t), P(C.c_uint8), C.c_int64, C.c_int,
                                   P(abi.bseq1_t), P(abi.mem_pestat_t)])
    sig("mi355x_index_upload", C.c_int, [C.c_int, P(abi.bwt_t), P(abi.bntseq_t), P(C.c_uint8)])
    sig("mi355x_index_alloc", C.c_int, [C.c_int, P(abi.bwt_t), P(abi.bntseq_t)])
    sig("mi355x_index_buffers", C.c_int, [P(C.c_void_p), P(C.c_size_t)] * 3)
    sig("mi355x_index_d2d", C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.c_int])
    sig("mi355x_index_commit", C.c_int, [])
    sig("mi355x_smem_batch", C.c_int, [P(abi.mem_opt_t), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_void_p, P(C.c_double), P(C.c_uint64)])
    sig("mi355x_sa_batch", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, P(C.c_double), P(C.c_uint64)])
    sig("mi355x_sa_batch2", C.c_int, [C.c_int, C.c_void_p, C.c_void_p, P(C.c_double), C.c_int])
    sig("mi355x_sa_dense_info", C.c_double, [P(C.c_size_t)])
    sig("mi355x_extend_batch", C.c_int, [P(abi.mem_opt_t), C.c_int] + [C.c_void_p] * 8 + [P(C.c_double), P(C.c_uint64)])
    sig("mi355x_chain_batch", C.c_int64, [P(abi.mem_opt_t), C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [C.c_int, C.c_void_p, C.c_int64, C.c_void_p])
    sig("mi355x_fastq_scan", C.c_int64, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p])
    sig("mi355x_fastq_chunks", C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p])
    sig("mi355x_fastq_fill", C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p])
    sig("mi355x_matesw_batch", C.c_int, [P(abi.mem_opt_t), C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 5 + [P(C.c_double)])
    sig("mi355x_last_stats", None, [P(abi.mi355x_stats_t)])
    sig("mi355x_finalize", None, [])
    sig("mi355x_host_ksw_align2", None, [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p])
    sig("mi355x_host_cpus", C.c_int, [])
    sig("mi355x_host_sort_dedup_patch", C.c_int, [P(abi.mem_opt_t), P(abi.bntseq_t), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int])
    sig("mi355x_host_reg2sam_se", None, [P(abi.mem_opt_t), P(abi.bntseq_t), C.c_void_p, P(abi.bseq1_t), C.c_void_p, C.c_int, C.c_int64])
    sig("mi355x_host_pestat", None, [P(abi.mem_opt_t), C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int])
    sig("mi355x_host_flt_chained_seeds", None, [P(abi.mem_opt_t), P(abi.bntseq_t), C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p])
    sig("mi355x_host_sam_pe", C.c_int, [P(abi.mem_opt_t), P(abi.bntseq_t), C.c_void_p, C.c_void_p, C.c_uint64, P(abi.bseq1_t), C.c_void_p, C.c_int, C.c_void_p, C.c_int])
    sig("mi355x_collect_sam", C.c_void_p, [P(abi.bseq1_t), C.c_int, P(C.c_size_t)])
    sig("mi355x_fixmate_pair", C.c_int, [P(abi.bseq1_t), P(abi.bseq1_t), P(abi.bntseq_t)])
    sig("mi355x_fixmate", C.c_int64, [P(abi.bseq1_t), C.c_int, P(abi.bntseq_t)])
    sig("mi355x_bgzf_bound", C.c_size_t, [C.c_size_t])
    sig("mi355x_bgzf_compress", C.c_size_t, [C.c_char_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t])
    sig("mi355x_bgzf_eof", C.c_size_t, [C.c_void_p])
    sig("mi355x_route_by_chr", C.c_int64, [C.c_char_p, C.c_size_t, P(abi.bntseq_t), C.c_int, P(C.c_void_p), P(C.c_size_t)])
    sig("bwa_set_rg", C.c_void_p, [C.c_char_p])
    sig("bwa_insert_header", C.c_void_p, [C.c_char_p, C.c_void_p])
    sig("bwa_idx2mem", C.c_int, [P(abi.bwaidx_t)])
    sig("bwa_mem2idx", C.c_int, [C.c_int64, C.c_void_p, P(abi.bwaidx_t)])
    sig("bwa_idx_destroy", None, [P(abi.bwaidx_t)])
    sig("mi355x_write_map", C.c_int, [C.c_char_p, C.c_char_p])
    sig("mi355x_init", C.c_int, [C.c_int, P(abi.bwaidx_t), P(mi355x_comm_t)])
    sig("mi355x_init_bcast_seconds", C.c_double, [])
    sig("mi355x_index_checksums", C.c_int, [C.c_void_p])
    sig("mi355x_device_count", C.c_int, [])
    sig("mi355x_device_memory", C.c_int, [P(C.c_size_t), P(C.c_size_t)])
    sig("mi355x_buffer_growths", C.c_ulonglong, [])
    sig("mi355x_pair_batch", C.c_int, [P(abi.mem_opt_t), C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
    sig("mi355x_global_batch", C.c_int, [P(abi.mem_opt_t), C.c_int64, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 7 +
        [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, P(C.c_double)])
    _LIB = lib
    return lib


_HIP = None


def _hip():
    global _HIP
    if _HIP is None:
        _HIP = C.CDLL("libamdhip64.so")
        _HIP.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return _HIP


def build_index(fasta, prefix):
    lib = load_library()
    if lib.mi355x_index_build(fasta.encode(), prefix.encode()) != 0:
        raise RuntimeError("index build failed")


class Engine:
    """One rank = one GPU: loads a bwa index, uploads it to HBM and aligns batches."""

    def __init__(self, prefix, device=0, upload=True, dist=None, rank=0, map_path=None, comm=None):
        """dist given (torch.distributed, nccl = RCCL): rank 0 uploads the index from host memory, every other rank only
        allocates the device buffers and receives occ blocks, SA and pac by broadcast over xGMI.
        map_path given: the index is attached from mpiBWA's `.map` image (bwa_mem2idx) instead of the five bwa files.
        comm given (mi355x_comm_t): residency through mi355x_init(), i.e. the C-side RCCL broadcast."""
        self.lib = load_library()
        if map_path is not None:
            import numpy as np
            self._map = np.fromfile(map_path, dtype=np.uint8)          # writable: bwa_mem2idx rebuilds the pointers inside
            self._idx_obj = abi.bwaidx_t()
            self.lib.bwa_mem2idx(len(self._map), self._map.ctypes.data, C.byref(self._idx_obj))
            self.idx = C.pointer(self._idx_obj)
        else:
            self.idx = self.lib.bwa_idx_load_from_disk(prefix.encode(), 7)
        self.bwt = self.idx.contents.bwt
        self.bns = self.idx.contents.bns
        self.pac = self.idx.contents.pac
        self.uploaded = False
        self.bcast_seconds = None
        if upload and comm is not None:
            if self.lib.mi355x_init(device, self.idx, C.byref(comm)) != 0:
                raise RuntimeError("mi355x_init failed")
            self.uploaded = True
            self.bcast_seconds = self.lib.mi355x_init_bcast_seconds()
        elif upload and dist is None:
            if self.lib.mi355x_index_upload(device, self.bwt, self.bns, self.pac) != 0:
                raise RuntimeError("mi355x_index_upload failed")
            self.uploaded = True
        elif upload:
            self._upload_by_broadcast(device, dist, rank)

    def _upload_by_broadcast(self, device, dist, rank):
        """One collective per index array, in place: every rank allocates the three device arrays, rank 0 fills its own from
        host memory (one H2D), then torch.distributed.broadcast (nccl = RCCL over xGMI) runs directly on views of those
        arrays — no staging tensor, no device-to-device copy, no synchronisation between pieces (they are queued back to
        back on the collective's stream and pipeline along the ring); one synchronize at the end, then every rank expands
        its own dense SA and jump table."""
        import time
        import torch
        os.environ["MPIBWA_SA_DENSE"] = os.environ.get("MPIBWA_SA_DENSE", "1")
        if self.lib.mi355x_index_alloc(device, self.bwt, self.bns) != 0:
            raise RuntimeError("mi355x_index_alloc failed")
        pv = [C.c_void_p() for _ in range(3)]
        sv = [C.c_size_t() for _ in range(3)]
        self.lib.mi355x_index_buffers(C.byref(pv[0]), C.byref(sv[0]), C.byref(pv[1]), C.byref(sv[1]), C.byref(pv[2]), C.byref(sv[2]))
        if rank == 0:
            HIP = _hip()
            srcs = [(C.cast(self.bwt.contents.bwt, C.c_void_p), int(self.bwt.contents.bwt_size) * 4),
                    (C.cast(self.bwt.contents.sa, C.c_void_p), int(self.bwt.contents.n_sa) * 8),
                    (C.cast(self.pac, C.c_void_p), int(self.bns.contents.l_pac) // 4 + 1)]
            for which, (src, nbytes) in enumerate(srcs):
                if HIP.hipMemcpy(C.c_void_p(pv[which].value), src, C.c_size_t(nbytes), 1) != 0:   # hipMemcpyHostToDevice
                    raise RuntimeError("hipMemcpy H2D failed")

        class _DevView:   # zero-copy torch view of a raw device allocation
            def __init__(self, ptr, nbytes):
                self.__cuda_array_interface__ = {"data": (ptr, False), "shape": (nbytes,), "typestr": "|u1", "version": 2}
        t0 = time.time()
        piece = int(os.environ.get("MPIBWA_BCAST_PIECE_MB", "256")) << 20
        views = []
        for which in range(3):
            v = torch.as_tensor(_DevView(pv[which].value, sv[which].value), device="cuda:%d" % device)
            views.append(v)
            for off in range(0, sv[which].value, piece):
                dist.broadcast(v[off:off + piece], src=0)
        torch.cuda.synchronize(device)
        self.bcast_seconds = time.time() - t0
        del views
        # every rank hashes what it now holds on its device; rank 0's numbers go round; a rank that differs stops the run here
        mine = (C.c_uint64 * 3)()
        if self.lib.mi355x_index_checksums(mine) != 0:
            raise RuntimeError("mi355x_index_checksums failed")
        self.index_checksums = [int(x) for x in mine]
        objs = [self.index_checksums if rank == 0 else None]
        dist.broadcast_object_list(objs, src=0)
        if objs[0] != self.index_checksums:
            raise RuntimeError("rank %d received a damaged index by broadcast: checksums %s, rank 0 sent %s" % (rank, self.index_checksums, objs[0]))
        if self.lib.mi355x_index_commit() != 0:
            raise RuntimeError("mi355x_index_commit failed")
        self.uploaded = True

    def opt(self, **kw):
        o = self.lib.mem_opt_init()
        for k, v in kw.items():
            setattr(o.contents, k, v)
        return o

    def process(self, opt, reads, n_processed=0, pes0=None, with_qual=True, comment=None):
        batch = abi.SeqBatch(libc, reads, with_qual=with_qual, comment=comment)
        self.lib.mem_process_seqs(opt, self.bwt, self.bns, self.pac, n_processed, batch.n, batch.arr, pes0)
        return batch.take_sam()

    def prewarm(self, opt, n_reads, read_len=150, n_calls=1):
        """First-use cost of n_calls call contexts paid now (include/mpibwa_amd.h: mi355x_prewarm); seconds it took."""
        self.lib.mi355x_prewarm.restype = C.c_double
        self.lib.mi355x_prewarm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        as_p = lambda x: C.cast(x, C.c_void_p)
        return float(self.lib.mi355x_prewarm(as_p(opt), as_p(self.bwt), as_p(self.bns), as_p(self.pac), int(n_reads), int(read_len), int(n_calls)))

    def process_batch(self, opt, batch, n_processed=0, pes0=None):
        self.lib.mem_process_seqs(opt, self.bwt, self.bns, self.pac, n_processed, batch.n, batch.arr, pes0)

    def collect_sam(self, batch):
        """All SAM text of the batch as one bytes object (frees the per-read strings like mpiBWA's writer does)."""
        n = C.c_size_t(0)
        p = self.lib.mi355x_collect_sam(batch.arr, batch.n, C.byref(n))
        out = C.string_at(p, n.value)
        libc.free(C.c_void_p(p))
        return out

    def stats(self):
        st = abi.mi355x_stats_t()
        self.lib.mi355x_last_stats(C.byref(st))
        return {k: getattr(st, k) for k, _ in st._fields_}

    # ---- stage-level kernels ----
    def smem(self, opt, seqs, cap=256):
        """seqs: list of uint8 code arrays → list of (n_i,4) uint64 arrays sorted by info."""
        off = np.zeros(len(seqs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        flat = np.concatenate(seqs).astype(np.uint8) if len(seqs) else np.zeros(0, np.uint8)
        out = np.zeros((len(seqs), cap, 4), dtype=np.uint64)
        cnt = np.zeros(len(seqs), dtype=np.int32)
        ms = C.c_double(0)
        nbytes = C.c_uint64(0)
        rc = self.lib.mi355x_smem_batch(opt, len(seqs), flat.ctypes.data, off.ctypes.data, cap, out.ctypes.data,
                                        cnt.ctypes.data, C.byref(ms), C.byref(nbytes))
        if rc != 0:
            raise RuntimeError("mi355x_smem_batch overflowed cap=%d" % cap)
        return [out[i, :cnt[i]].copy() for i in range(len(seqs))], ms.value, nbytes.value

    def sa(self, ks):
        ks = np.ascontiguousarray(ks, dtype=np.uint64)
        out = np.zeros(len(ks), dtype=np.uint64)
        ms = C.c_double(0)
        nbytes = C.c_uint64(0)
        self.lib.mi355x_sa_batch(len(ks), ks.ctypes.data, out.ctypes.data, C.byref(ms), C.byref(nbytes))
        return out, ms.value, nbytes.value

    def sa_dense(self, ks):
        """Lookups through the dense SA table (None if the table was not expanded)."""
        ks = np.ascontiguousarray(ks, dtype=np.uint64)
        out = np.zeros(len(ks), dtype=np.uint64)
        ms = C.c_double(0)
        rc = self.lib.mi355x_sa_batch2(len(ks), ks.ctypes.data, out.ctypes.data, C.byref(ms), 1)
        return (out, ms.value) if rc == 0 else None

    def chains(self, opt, lens, l_rep, seeds_per_read, which):
        """Chaining stage for reads given by their seeds [(rbeg, qbeg, len), ...]; which = 0 device kernel, 1 host path.
        Returns per read None (device declines) or a list of chains (rid, far_beg, far_end, rmax0, rmax1, frac_bits, [(rbeg, qbeg, len)...])."""
        n = len(seeds_per_read)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in seeds_per_read])
        S = int(off[n])
        rbeg = np.zeros(max(S, 1), dtype=np.uint64)
        ql = np.zeros(2 * max(S, 1), dtype=np.int32)
        k = 0
        for s in seeds_per_read:
            for rb, qb, ln in s:
                rbeg[k] = rb; ql[2 * k] = qb; ql[2 * k + 1] = ln
                k += 1
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        l_rep = np.ascontiguousarray(l_rep, dtype=np.int32)
        cap = n + 16 * S + 64 * n + 64
        out = np.zeros(cap, dtype=np.int64)
        out_off = np.zeros(n + 1, dtype=np.int64)
        rc = self.lib.mi355x_chain_batch(opt, C.cast(self.bns, C.c_void_p), n, lens.ctypes.data, l_rep.ctypes.data, off.ctypes.data, rbeg.ctypes.data,
                                         ql.ctypes.data, which, out.ctypes.data, cap, out_off.ctypes.data)
        assert rc >= 0
        res = []
        for r in range(n):
            p = int(out_off[r])
            nc = int(out[p]); p += 1
            if nc < 0:
                res.append(None)
                continue
            chs = []
            for _ in range(nc):
                rid, ns, fb, fe, r0, r1, fr = (int(x) for x in out[p:p + 7]); p += 7
                sd = [tuple(int(x) for x in out[p + 3 * j:p + 3 * j + 3]) for j in range(ns)]
                p += 3 * ns
                chs.append((rid, fb, fe, r0, r1, fr, sd))
            res.append(chs)
        return res

    REG_DT = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("rid", "<i4"), ("score", "<i4"), ("truesc", "<i4"), ("w", "<i4"),
                       ("seedcov", "<i4"), ("seedlen0", "<i4"), ("frac_rep", "<f4"), ("pad", "<i4")])
    DESC_DT = np.dtype([("rb", "<i8"), ("re", "<i8"), ("qb", "<i4"), ("qe", "<i4"), ("req", "<i4"), ("rid", "<i4"), ("flag", "<i4"), ("mapq", "<i4"),
                        ("score", "<i4"), ("sub", "<i4")])
    AREQ_DT = np.dtype([("rb", "<i8"), ("re", "<i8"), ("read", "<i4"), ("qb", "<i4"), ("qe", "<i4"), ("w2", "<i4"), ("truesc", "<i4"), ("pad", "<i4")])

    def pairs(self, opt, pes, regs, n_regs, max_len=150, n_processed=0):
        """pair_simple_kernel on pairs given by their regions: regs (2 n_pairs, mi355x_pair_maxreg()) of REG_DT, n_regs (2 n_pairs).
        -> status (n_pairs,) uint8, desc (2 n_pairs,) DESC_DT, req (2 n_pairs,) AREQ_DT"""
        regs = np.ascontiguousarray(regs, dtype=self.REG_DT)
        assert regs.shape[1] == self.lib.mi355x_pair_maxreg()
        n_regs = np.ascontiguousarray(n_regs, dtype=np.int32)
        n_pairs = len(n_regs) // 2
        status = np.zeros(n_pairs, dtype=np.uint8)
        desc = np.zeros(2 * n_pairs, dtype=self.DESC_DT)
        req = np.zeros(2 * n_pairs, dtype=self.AREQ_DT)
        rc = self.lib.mi355x_pair_batch(opt, C.cast(self.bns, C.c_void_p), C.cast(pes, C.c_void_p), n_processed, n_pairs, regs.ctypes.data, n_regs.ctypes.data,
                                        max_len, status.ctypes.data, desc.ctypes.data, req.ctypes.data)
        if rc != 0:
            raise RuntimeError("mi355x_pair_batch: the kernel cannot use these insert-size statistics")
        return status, desc, req

    def matesw(self, opt, l_pac, pac, reads, rb, re, read, is_rev):
        """mem_matesw's ksw_align2 for windows of `pac`; returns (n_req x 8 int32, kernel ms)."""
        n = len(reads)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in reads])
        flat = np.concatenate(reads).astype(np.uint8)
        rb = np.ascontiguousarray(rb, dtype=np.int64)
        re = np.ascontiguousarray(re, dtype=np.int64)
        read = np.ascontiguousarray(read, dtype=np.int32)
        is_rev = np.ascontiguousarray(is_rev, dtype=np.int32)
        pac = np.ascontiguousarray(pac, dtype=np.uint8)
        out = np.zeros((len(rb), 8), dtype=np.int32)
        ms = C.c_double(0)
        self.lib.mi355x_matesw_batch(opt, int(l_pac), pac.ctypes.data, n, flat.ctypes.data, off.ctypes.data, len(rb), rb.ctypes.data,
                                     re.ctypes.data, read.ctypes.data, is_rev.ctypes.data, out.ctypes.data, C.byref(ms))
        return out, ms.value

    def global_align(self, opt, l_pac, pac, reads, rb, re, read, qb, qe, w, truesc, which=0, cigar_cap=128, md_cap=800):
        """mem_reg2aln's DP loop (CIGAR / MD / NM) for regions of `reads` against windows of `pac`, by aln_kernel.
        Returns (hdr (n,5) int32: score, NM, n_cigar, md_len, flags; list of cigar arrays; list of MD bytes; kernel ms)."""
        n = len(reads)
        off = np.zeros(n + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(s) for s in reads])
        flat = np.concatenate(reads).astype(np.uint8)
        arr = lambda x, t: np.ascontiguousarray(x, dtype=t)
        rb, re = arr(rb, np.int64), arr(re, np.int64)
        read, qb, qe, w, truesc = (arr(x, np.int32) for x in (read, qb, qe, w, truesc))
        pac = arr(pac, np.uint8)
        nr = len(rb)
        hdr = np.zeros((nr, 5), dtype=np.int32)
        cig = np.zeros((nr, cigar_cap), dtype=np.uint32)
        md = np.zeros((nr, md_cap), dtype=np.uint8)
        ms = C.c_double(0)
        rc = self.lib.mi355x_global_batch(opt, int(l_pac), pac.ctypes.data, n, flat.ctypes.data, off.ctypes.data, nr, rb.ctypes.data,
                                          re.ctypes.data, read.ctypes.data, qb.ctypes.data, qe.ctypes.data, w.ctypes.data,
                                          truesc.ctypes.data, which, hdr.ctypes.data, cig.ctypes.data, cigar_cap, md.ctypes.data, md_cap,
                                          C.byref(ms))
        if rc != 0:
            raise RuntimeError("mi355x_global_batch: result beyond cigar_cap / md_cap")
        cigs = [cig[i, :hdr[i, 2]].copy() for i in range(nr)]
        mds = [md[i, :hdr[i, 3]].tobytes() for i in range(nr)]
        return hdr, cigs, mds, ms.value

    def extend(self, opt, qs, ts, w, h0, end_bonus):
        n = len(qs)
        qoff = np.zeros(n + 1, dtype=np.int64)
        qoff[1:] = np.cumsum([len(s) for s in qs])
        toff = np.zeros(n + 1, dtype=np.int64)
        toff[1:] = np.cumsum([len(s) for s in ts])
        qf = np.concatenate(qs).astype(np.uint8)
        tf = np.concatenate(ts).astype(np.uint8)
        w = np.ascontiguousarray(w, dtype=np.int32)
        h0 = np.ascontiguousarray(h0, dtype=np.int32)
        eb = np.ascontiguousarray(end_bonus, dtype=np.int32)
        out = np.zeros((n, 6), dtype=np.int32)
        ms = C.c_double(0)
        cells = C.c_uint64(0)
        self.lib.mi355x_extend_batch(opt, n, qf.ctypes.data, qoff.ctypes.data, tf.ctypes.data, toff.ctypes.data,
                                     w.ctypes.data, h0.ctypes.data, eb.ctypes.data, out.ctypes.data, C.byref(ms),
                                     C.byref(cells))
        return out, ms.value, cells.value
