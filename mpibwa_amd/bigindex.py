"""GRCh38-sized synthetic references for the benchmark: seeded genome in packed form, index built on the GPU
(csrc/index_gpu.hip), device-side broadcast over RCCL for the other ranks, and a vectorised read simulator.

Generator: `n_contigs` contigs, uniform base composition, ~5 % of the bases inside
planted repeat families (200-2000 bp units, 2-40 copies, 0.5-5 % divergence, half of the copies reverse-complemented),
no N.  Everything derives from `seed`."""
import ctypes as C
import os
import time

import numpy as np

from . import abi, api

_REV_BYTE = None
BUILDER_VERSION = 3   # bump whenever index_gpu.hip changes, so a cached index of an older build is never reused


def _revcomp_bytes(a):
    """Reverse-complement a run of packed bytes (4 bases per byte, MSB first)."""
    global _REV_BYTE
    if _REV_BYTE is None:
        t = np.zeros(256, dtype=np.uint8)
        for b in range(256):
            v = 0
            for k in range(4):
                base = (b >> (2 * k)) & 3          # k-th base from the right
                v |= (3 - base) << (2 * (3 - k))   # becomes k-th from the left, complemented
            t[b] = v
        _REV_BYTE = t
    return _REV_BYTE[a[::-1]]


def synth_packed_genome(total_bases, seed=38, n_contigs=24, repeat_frac=0.05):
    total_bases = int(total_bases) // 4 * 4
    rng = np.random.default_rng(seed)
    nbytes = total_bases // 4
    pac = rng.integers(0, 256, size=nbytes + 1, dtype=np.uint8)
    pac[nbytes] = 0
    target = int(total_bases * repeat_frac)
    placed = 0
    while placed < target:
        unit = int(rng.integers(50, 500))            # bytes = 200..2000 bases
        copies = int(rng.integers(2, 40))
        div = float(rng.uniform(0.005, 0.05))
        src = int(rng.integers(0, nbytes - unit))
        u = pac[src:src + unit].copy()
        for _ in range(copies):
            dst = int(rng.integers(0, nbytes - unit))
            cp = _revcomp_bytes(u) if rng.random() < 0.5 else u.copy()
            nmut = rng.binomial(unit * 4, div)
            if nmut:
                where = rng.integers(0, unit * 4, size=nmut)
                delta = rng.integers(1, 4, size=nmut).astype(np.uint8)
                sh = ((3 - (where & 3)) * 2).astype(np.uint8)
                np.bitwise_xor.at(cp, where >> 2, (delta << sh).astype(np.uint8))
            pac[dst:dst + unit] = cp
            placed += unit * 4
    lens = np.full(n_contigs, total_bases // n_contigs // 4 * 4, dtype=np.int64)
    lens[-1] += total_bases - lens.sum()
    return pac, lens


def write_meta_files(prefix, pac, lens):
    l_pac = int(lens.sum())
    with open(prefix + ".pac", "wb") as f:
        body = pac[:l_pac // 4 + (1 if l_pac % 4 else 0)]
        f.write(body.tobytes())
        if l_pac % 4 == 0:
            f.write(b"\x00")
        f.write(bytes([l_pac % 4]))
    with open(prefix + ".ann", "w") as f:
        f.write("%d %d %u\n" % (l_pac, len(lens), 11))
        off = 0
        for i, L in enumerate(lens):
            f.write("0 chrS%d (null)\n" % (i + 1))
            f.write("%d %d 0\n" % (off, L))
            off += int(L)
    with open(prefix + ".amb", "w") as f:
        f.write("%d %d 0\n" % (l_pac, len(lens)))


def unpack_windows(pac, starts, width):
    idx = starts[:, None] + np.arange(width, dtype=np.int64)[None, :]
    return ((pac[idx >> 2] >> ((~idx & 3) << 1).astype(np.uint8)) & 3).astype(np.uint8)


def synth_packed_genome_grch38like(total_bases, seed=38, n_contigs=24, repeat_frac=0.5):
    """The generator of SURVEY.md §8d: order-3 Markov base composition with GC ~ 41 %, and `repeat_frac` of the bases
    inside planted repeat families with copy numbers 2 .. 10^4 (log-uniform), units of 100 .. 6000 bp, 0 .. 5 % divergence
    of every copy from its family's consensus, half of the copies reverse-complemented; no N.  Packed 2 bit/base."""
    total_bases = int(total_bases) // 4 * 4
    rng = np.random.default_rng(seed)
    nbytes = total_bases // 4
    # order-3 Markov chain over bases, advanced a byte (4 bases) at a time on many interleaved streams: the next byte is
    # drawn given the last three bases through a 4096-entry quantile table per context
    base_p = np.array([0.295, 0.205, 0.205, 0.295])                      # A C G T: GC = 41 %
    trans = rng.dirichlet(base_p * 12.0, size=64)                       # P(next base | 3 previous bases)
    lut = np.zeros((64, 4096), dtype=np.uint8)
    for ctx in range(64):
        pb = np.zeros(256)
        for b in range(256):
            c, p = ctx, 1.0
            for k in range(4):
                x = (b >> (6 - 2 * k)) & 3
                p *= trans[c, x]
                c = ((c << 2) | x) & 63
            pb[b] = p
        edges = np.floor(np.cumsum(pb) / pb.sum() * 4096 + 0.5).astype(np.int64)
        lut[ctx] = np.repeat(np.arange(256, dtype=np.uint8), np.diff(np.concatenate([[0], edges])).clip(min=0))[:4096] if edges[-1] >= 4096 else 0
    n_streams = 1 << 16 if nbytes < (1 << 26) else 1 << 20
    ls = (nbytes + n_streams - 1) // n_streams
    arr = np.empty((ls, n_streams), dtype=np.uint8)
    ctx = rng.integers(0, 64, size=n_streams)
    for t in range(ls):
        b = lut[ctx, rng.integers(0, 4096, size=n_streams)]
        arr[t] = b
        ctx = (b & 63).astype(np.int64)
    pac = np.empty(nbytes + 1, dtype=np.uint8)
    pac[:nbytes] = arr.T.reshape(-1)[:nbytes]
    pac[nbytes] = 0
    del arr
    target = int(total_bases * repeat_frac)
    placed = 0
    while placed < target:
        unit = int(rng.integers(25, 1500))                                # bytes = 100 .. 6000 bases
        copies = int(np.exp(rng.uniform(np.log(2.0), np.log(1e4))))
        copies = max(2, min(copies, (target - placed) // (unit * 4) + 2))
        div = float(rng.uniform(0.0, 0.05))
        src = int(rng.integers(0, nbytes - unit))
        u = pac[src:src + unit].copy()
        cp = np.tile(u, (copies, 1))
        rc = rng.random(copies) < 0.5
        if rc.any():
            cp[rc] = _revcomp_bytes(u)
        nmut = int(rng.binomial(copies * unit * 4, div))
        if nmut:
            where = rng.integers(0, copies * unit * 4, size=nmut)
            delta = rng.integers(1, 4, size=nmut).astype(np.uint8)
            sh = ((3 - (where & 3)) * 2).astype(np.uint8)
            np.bitwise_xor.at(cp.reshape(-1), where >> 2, (delta << sh).astype(np.uint8))
        dst = rng.integers(0, nbytes - unit, size=copies)
        for i in range(copies):
            pac[dst[i]:dst[i] + unit] = cp[i]
        placed += copies * unit * 4
    lens = np.full(n_contigs, total_bases // n_contigs // 4 * 4, dtype=np.int64)
    lens[-1] += total_bases - lens.sum()
    return pac, lens


class BigIndex:
    def __init__(self, prefix, pac, lens, engine):
        self.prefix, self.pac, self.lens, self.engine = prefix, pac, lens, engine
        self.l_pac = int(lens.sum())
        self.blk_bytes = int(engine.bwt.contents.bwt_size) * 4
        self.sa_bytes = int(engine.bwt.contents.n_sa) * 8

    def simulate_pairs(self, n_pairs, seed=1, read_len=150, frag_mean=400.0, frag_sd=50.0, sub=0.01, indel=0.001,
                       frac_random=0.02):
        """2 x read_len paired-end reads as [(name, bytes, bytes)], vectorised over the packed reference."""
        rng = np.random.default_rng(seed)
        bounds = np.concatenate([[0], np.cumsum(self.lens)])
        W = read_len + 12                                   # slack for deletions
        fl = np.clip(rng.normal(frag_mean, frag_sd, size=n_pairs), read_len + 20, None).astype(np.int64)
        contig = rng.choice(len(self.lens), size=n_pairs, p=self.lens / self.lens.sum())
        room = self.lens[contig] - fl - W - 1
        p = bounds[contig] + (rng.random(n_pairs) * room).astype(np.int64)
        w1 = unpack_windows(self.pac, p, W)                             # fragment head, forward
        w2 = unpack_windows(self.pac, p + fl - W, W)                    # fragment tail, forward
        w2rc = (3 - w2)[:, ::-1]                                        # mate reads the other strand from the tail
        flip = rng.random(n_pairs) < 0.5
        a = np.where(flip[:, None], w2rc, w1)
        b = np.where(flip[:, None], w1, w2rc)     # fragment from the reverse strand: read 1 = revcomp(tail), read 2 = head
        out = []
        lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
        for arr in (a, b):
            m = rng.random(arr.shape) < sub
            arr[m] = (arr[m] + rng.integers(1, 4, size=int(m.sum()), dtype=np.uint8)) & 3
        rand_pair = rng.random(n_pairs) < frac_random
        nr = int(rand_pair.sum())
        a[rand_pair] = rng.integers(0, 4, size=(nr, W), dtype=np.uint8)
        b[rand_pair] = rng.integers(0, 4, size=(nr, W), dtype=np.uint8)
        reads = [None] * n_pairs
        has_indel = rng.random((n_pairs, 2)) < indel * read_len
        A, B = lut[a], lut[b]
        for i in range(n_pairs):
            r = []
            for k, row in enumerate((A[i], B[i])):
                if has_indel[i, k]:
                    pos = int(rng.integers(5, read_len - 5))
                    ln = min(int(rng.geometric(0.5)), 6)
                    if rng.random() < 0.5:
                        row = np.concatenate([row[:pos], row[pos + ln:]])          # deletion from the read
                    else:
                        row = np.concatenate([row[:pos], lut[rng.integers(0, 4, size=ln)], row[pos:]])
                r.append(row[:read_len].tobytes())
            reads[i] = ("r%d" % i, r[0], r[1])
        return reads


def _load_packed_genome(prefix):
    """The packed genome and contig lengths back from the node-local files rank 0 wrote (.pac mapped, not copied)."""
    with open(prefix + ".ann") as f:
        l_pac, n_seqs, _ = f.readline().split()
        lens = []
        for _ in range(int(n_seqs)):
            f.readline()
            lens.append(int(f.readline().split()[1]))
    lens = np.array(lens, dtype=np.int64)
    assert int(l_pac) == int(lens.sum())
    pac = np.memmap(prefix + ".pac", dtype=np.uint8, mode="r")[:int(l_pac) // 4 + 1]
    return pac, lens


def make_or_get(workdir, genome_mbp=3100.0, seed=38, rank=0, world=1, local_rank=0, dist=None, log=None, repeat_frac=0.05, model="uniform"):
    lib = api.load_library()
    lib.mi355x_index_build_gpu.restype = C.c_int
    lib.mi355x_index_build_gpu.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_char_p, C.POINTER(C.c_double)]
    tag = "" if repeat_frac == 0.05 else "_r%d" % int(round(repeat_frac * 100))
    if model != "uniform":
        tag += "_" + model
    prefix = os.path.join(workdir, "synth_%dM_s%d%s_b%d.fa" % (int(genome_mbp), seed, tag, BUILDER_VERSION))
    # the genome is generated ONCE per node: rank 0 writes it (with the index), every rank maps .pac from the node-local file
    if rank == 0 and not os.path.exists(prefix + ".ok"):
        t0 = time.time()
        gen = synth_packed_genome_grch38like if model == "grch38like" else synth_packed_genome
        pac, lens = gen(genome_mbp * 1e6, seed=seed, repeat_frac=repeat_frac)
        if log:
            log("synthetic genome: %.1f Mbp in %d contigs, generated in %.1f s" % (lens.sum() / 1e6, len(lens), time.time() - t0))
        write_meta_files(prefix, pac, lens)
        secs = C.c_double(0)
        lib.mi355x_index_build_gpu(local_rank, pac.ctypes.data, int(lens.sum()), prefix.encode(), C.byref(secs))
        open(prefix + ".ok", "w").write("built in %.1f s\n" % secs.value)   # only a completed build is ever reused
        if log:
            log("FM-index built on the GPU in %.1f s" % secs.value)
        del pac
    if dist is not None:
        dist.barrier()
    pac, lens = _load_packed_genome(prefix)
    # one rank per GPU.  Every rank maps the host side of the index from the node-local files (the host stages need pac
    # and the contig table); the device side is uploaded once by rank 0 and broadcast to the other GPUs over RCCL/xGMI.
    eng = api.Engine(prefix, device=local_rank, dist=dist, rank=rank)
    if log and eng.bcast_seconds is not None:
        log("index broadcast over RCCL: %.2f s" % eng.bcast_seconds)
    return BigIndex(prefix, pac, lens, eng)
