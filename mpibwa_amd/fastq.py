"""The caller's side of `mem_process_seqs`: FASTQ files -> mpiBWA's chunks -> bseq1_t[] -> SAM stream.

Host-side mirror of what mpiBWA's `main` does around the hot path for one rank (src/mainParallel.c:730-1055 equal-size
pairs, :1780-2400 trimmed pairs, :2700-3120 single end), on top of the C ABI (`mi355x_fastq_scan / _chunks / _fill`):
the chunk boundaries, the strings behind every `bseq1_t` and the order of the SAM records are the reference's; the
reference's chunk count is carried from rank to rank, so they do not depend on the number of ranks either.  Reading,
aligning and writing overlap (the reference's docs/TODO:4): the SAM strings of chunk i are concatenated, written and freed
by a writer thread while chunk i+1 is on the GPU (src/mainParallel.c:103-127 `copy_buffer_thr` does the same).
"""
import ctypes as C
import gzip
import os
import queue
import threading

import numpy as np

from . import abi

_libc = C.CDLL(None if os.environ.get("MPIBWA_SANITIZER_LIB") else "libc.so.6")   # (under tools/san_host.sh: the sanitizer's malloc / free)
_libc.free.argtypes = [C.c_void_p]


def _load(path):
    """(keep-alive object, address, length) of the file's bytes in writable memory with one spare byte after them: the
    strings are NUL-terminated (and the bases nt4-encoded) in place, the last quality line even when the file has no final
    newline.  Plain files are read straight into the buffer (one copy out of the page cache).  A copy-on-write mapping of
    the file looks cheaper but is not: every page is written to later, and 16 threads taking copy-on-write faults on one
    address space cost more inside mem_process_seqs than the read costs here."""
    path = str(path)
    if path.endswith(".gz"):
        data = gzip.open(path, "rb").read()
        buf = C.create_string_buffer(data, len(data) + 1)
        return buf, C.addressof(buf), len(data)
    size = os.path.getsize(path)
    buf = np.empty(size + 1, dtype=np.uint8)
    with open(path, "rb", buffering=0) as f:
        view, got = memoryview(buf)[:size], 0
        while got < size:
            n = f.readinto(view[got:])
            if not n:
                raise IOError("%s: short read" % path)
            got += n
    buf[size] = 0
    return buf, buf.ctypes.data, size


class FastqFile:
    """One FASTQ file in memory: record offsets and bases per record (find_reads_size_and_offsets, src/parallel_aux.c:682)."""

    def __init__(self, lib, path):
        self.lib = lib
        self.buf, self._addr, self.len = _load(path)
        cap = max(16, self.len // 32)          # a record has at least 8 bytes; grow if the guess is short
        while True:
            off = np.zeros(cap + 1, dtype=np.int64)
            bases = np.zeros(cap, dtype=np.int32)
            n = lib.mi355x_fastq_scan(self._addr, self.len, cap, off.ctypes.data, bases.ctypes.data)
            if n < 0:
                raise ValueError("%s: malformed FASTQ record at byte %d" % (path, -n - 1))
            if n <= cap:
                break
            cap = n
        self.n = int(n)
        self.off = off[:self.n + 1].copy()
        self.bases = bases[:self.n].copy()

    @property
    def addr(self):
        return self._addr


def chunk_starts(lib, bases1, bases2, maxsiz):
    """mpiBWA's chunk rule (src/parallel_aux.c:1520-1546): first record of every chunk, plus n at the end."""
    n = len(bases1)
    cap = n + 1
    first = np.zeros(cap + 1, dtype=np.int64)
    b1 = np.ascontiguousarray(bases1, dtype=np.int32)
    b2 = None if bases2 is None else np.ascontiguousarray(bases2, dtype=np.int32)
    k = lib.mi355x_fastq_chunks(b1.ctypes.data, None if b2 is None else b2.ctypes.data, n, int(maxsiz), cap, first.ctypes.data)
    return first[:k + 1].copy()


class FastqSource:
    """R1 (+R2) -> chunks of bseq1_t.  mode: 'pe' (equal-size pairs, K/2 per file), 'pe_trim' (K over both files), 'se'."""

    def __init__(self, lib, r1, r2=None, K=10_000_000, copy_comment=False, mode=None):
        self.lib = lib
        if r2 is not None:   # the two files are scanned side by side (the scan is one pass in C, outside the GIL)
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=2) as pool:
                a, b = pool.submit(FastqFile, lib, r1), pool.submit(FastqFile, lib, r2)
                self.f1, self.f2 = a.result(), b.result()
        else:
            self.f1, self.f2 = FastqFile(lib, r1), None
        if self.f2 is not None and self.f1.n != self.f2.n:
            raise ValueError("the two FASTQ files hold %d and %d reads" % (self.f1.n, self.f2.n))
        if mode is None:
            # mpiBWA takes the equal-size branch when both files have the same byte size (src/mainParallel.c:703-728)
            mode = "se" if self.f2 is None else ("pe" if self.f1.len == self.f2.len else "pe_trim")
        self.mode, self.copy_comment = mode, copy_comment
        if mode == "pe":
            self.starts = chunk_starts(lib, self.f1.bases, None, K // 2)
        elif mode == "pe_trim":
            self.starts = chunk_starts(lib, self.f1.bases, self.f2.bases, K)
        else:
            self.starts = chunk_starts(lib, self.f1.bases, None, K)

    @property
    def n_chunks(self):
        return len(self.starts) - 1

    def chunk(self, c):
        """(bseq1_t array, n_reads, keepalive) for chunk c; the strings live inside the file buffers."""
        first, count = int(self.starts[c]), int(self.starts[c + 1] - self.starts[c])
        files = 1 if self.f2 is None else 2
        rec = np.zeros(max(files * count, 1), dtype=abi.SeqBatch.dtype())
        rc = self.lib.mi355x_fastq_fill(self.f1.addr, self.f1.off.ctypes.data, self.f2.addr if self.f2 else None,
                                        self.f2.off.ctypes.data if self.f2 else None, first, count, int(self.copy_comment),
                                        int(self.mode == "pe"), rec.ctypes.data)
        if rc < 0:
            raise ValueError("malformed FASTQ record %d" % (first - rc - 1))
        return rec, files * count


def align_files(engine, opt, r1, r2=None, out=None, K=10_000_000, copy_comment=False, mode=None, n_processed0=0, rank=0, world=1,
                in_flight=4):
    """`mpiBWA mem` for one rank: its chunks through mem_process_seqs, SAM records in read order.

    The chunk list is the same on every rank (it does not depend on the number of ranks); rank r of `world` takes chunks
    r, r + world, ... — the reference hands chunks out with a fetch-and-add counter (src/mainParallel.c:1112-1119), which
    balances better on uneven data but yields the same set of records.  With world > 1 the trimmed branch's running
    n_processed (reads this rank has already done, :2355-2357) is counted over this rank's chunks, as in the reference.
    in_flight: caller threads inside mem_process_seqs at once (the library takes up to eight: the GPU half of chunk i+1 overlaps the
    host half of chunk i); the records are written in chunk order whatever the completion order.
    out: a binary file object (the SAM body is written to it) or None (the body is returned as bytes).
    Returns (bytes or None, per-chunk read counts of this rank)."""
    from concurrent.futures import ThreadPoolExecutor
    lib = engine.lib
    src = FastqSource(lib, r1, r2, K=K, copy_comment=copy_comment, mode=mode)
    pieces = [] if out is None else None
    todo = queue.Queue(maxsize=max(2, in_flight + 1))
    err = []

    def writer():   # copy_buffer_thr: concatenate + free the chunk's SAM strings while the next chunks are being aligned
        while True:
            fut = todo.get()
            if fut is None:
                return
            try:
                rec, n = fut.result()
                tot = C.c_size_t(0)
                p = lib.mi355x_collect_sam(C.cast(rec.ctypes.data, C.POINTER(abi.bseq1_t)), n, C.byref(tot))
                try:
                    if out is None:
                        pieces.append(C.string_at(p, tot.value))
                    else:   # straight from the C buffer to the file, no Python copy of the chunk's SAM text
                        out.write(memoryview((C.c_char * tot.value).from_address(p)))
                finally:
                    _libc.free(C.c_void_p(p))
            except Exception as e:  # pragma: no cover
                err.append(e)

    def align(c, npz):
        rec, n = src.chunk(c)
        lib.mem_process_seqs(opt, engine.bwt, engine.bns, engine.pac, npz, n, C.cast(rec.ctypes.data, C.POINTER(abi.bseq1_t)), None)
        return rec, n

    th = threading.Thread(target=writer, daemon=True)
    th.start()
    counts = []
    files = 1 if src.f2 is None else 2
    with ThreadPoolExecutor(max_workers=max(1, in_flight)) as pool:
        for c in range(rank, src.n_chunks, world):
            # mpiBWA passes n_processed = 0 for equal-size pairs and single end, and the reads already done by the rank in
            # the trimmed branch (src/mainParallel.c:1314, 2355-2357, 3093)
            npz = n_processed0 + (sum(counts) if src.mode == "pe_trim" else 0)
            counts.append(files * int(src.starts[c + 1] - src.starts[c]))
            todo.put(pool.submit(align, c, npz))   # blocks while in_flight + 1 chunks are pending: bounded memory
        todo.put(None)
        th.join()
    if err:
        raise err[0]
    return (b"".join(pieces) if out is None else None), counts
