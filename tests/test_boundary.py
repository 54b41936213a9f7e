"""The drop-in boundary without a GPU: the header is plain C with the reference's layouts, the helper symbols behave like the
reference's (src/bwa.c:413-476), the `.map` image is the reference's (src/bwa.c:310-386) in both directions, and the
reference's own driver objects link against the product library with no undefined symbol."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
have_ref_src = os.path.isdir(REF)
needs_ref = pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libbwaref.so not built")


def test_header_compiles_as_c_with_the_asserted_layouts(built):
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "csrc", "abi_check.c")])


@pytest.mark.skipif(not have_ref_src, reason="reference headers only exist in the build container")
def test_same_asserts_hold_for_the_reference_headers(tmp_path):
    """tests/csrc/abi_check.c again, with `mpibwa_amd.h` resolving to a two-line shim that includes the reference's headers:
    every size and offset the product header asserts is the reference's."""
    (tmp_path / "mpibwa_amd.h").write_text('#include "bwamem.h"\n#include "bwa.h"\n')
    subprocess.check_call(["gcc", "-std=c11", "-fsyntax-only", "-w", "-I", str(tmp_path), "-I", REF,
                           os.path.join(ROOT, "tests", "csrc", "abi_check.c")])


RG_CASES = [
    b"@RG\\tID:grp1\\tSM:s\\tPL:illumina", b"@RG\\tID:x", b"@RG\\tSM:s\\tID:abc\\nrest", b"@RG\\tSM:noid", b"RG\\tID:a",
    b"@RG\tID:literal_tab", b"@RG\\tID:" + b"k" * 255, b"@RG\\tID:" + b"k" * 256, b"@RG\\tID:q\\\\z\\rend\\x", b"@RG\\tID:a\\",
]
HDR_CASES = [(b"@CO\\tfirst", None), (b"@CO\\tsecond\\nline", b"@HD\tVN:1.0"), (b"no-at-sign", b"@HD\tVN:1.0"), (b"no-at-sign", None),
             (b"@PG\\tID:x\\\\y", b"@CO\tkeeps\\tits escapes")]


@needs_ref
def test_set_rg_and_insert_header_match_the_reference(built):
    from mpibwa_amd import api
    ours, ref = api.load_library(), po.ref_lib()
    for lib in (ours, ref):
        lib.bwa_set_rg.restype = C.c_void_p
        lib.bwa_set_rg.argtypes = [C.c_char_p]
        lib.bwa_insert_header.restype = C.c_void_p
        lib.bwa_insert_header.argtypes = [C.c_char_p, C.c_void_p]
        C.c_int.in_dll(lib, "bwa_verbose").value = 0

    def take(p):
        if not p:
            return None
        s = C.string_at(p)
        po.libc.free(C.c_void_p(p))
        return s

    for s in RG_CASES:
        got = [(take(lib.bwa_set_rg(s)), bytes((C.c_char * 256).in_dll(lib, "bwa_rg_id"))) for lib in (ours, ref)]
        assert got[0] == got[1], s
    assert any(g is not None for g in [take(ours.bwa_set_rg(s)) for s in RG_CASES[:3]])
    for lib in (ours, ref):      # leave no read group behind: both libraries would tag every later record with it
        take(lib.bwa_set_rg(b"none"))
        assert bytes((C.c_char * 256).in_dll(lib, "bwa_rg_id")) == bytes(256)
    po.libc.strdup = po.libc.strdup
    po.libc.strdup.restype = C.c_void_p
    po.libc.strdup.argtypes = [C.c_char_p]
    for s, hdr in HDR_CASES:
        res = []
        for lib in (ours, ref):
            h = po.libc.strdup(hdr) if hdr is not None else None
            r = lib.bwa_insert_header(s, h)
            res.append(take(r))
        assert res[0] == res[1], (s, hdr)
    C.c_int.in_dll(ours, "bwa_verbose").value = 3
    C.c_int.in_dll(ref, "bwa_verbose").value = 3


def _scrub(img, lib_like_sizes):
    """zero the pointer fields of a `.map` image (heap addresses of the process that packed it)"""
    from mpibwa_amd import abi
    img = img.copy()
    bwt = abi.bwt_t.from_buffer(img, 0)
    k = C.sizeof(abi.bwt_t) + int(bwt.bwt_size) * 4 + int(bwt.n_sa) * 8
    img[abi.bwt_t.bwt.offset:abi.bwt_t.bwt.offset + 8] = 0
    img[abi.bwt_t.sa.offset:abi.bwt_t.sa.offset + 8] = 0
    bns = abi.bntseq_t.from_buffer(img, k)
    n_seqs, n_holes = int(bns.n_seqs), int(bns.n_holes)
    for f in (abi.bntseq_t.anns, abi.bntseq_t.ambs, abi.bntseq_t.fp_pac):
        img[k + f.offset:k + f.offset + 8] = 0
    k += C.sizeof(abi.bntseq_t) + n_holes * C.sizeof(abi.bntamb1_t)
    for i in range(n_seqs):
        o = k + i * C.sizeof(abi.bntann1_t)
        img[o + abi.bntann1_t.name.offset:o + abi.bntann1_t.name.offset + 16] = 0
    del bwt, bns
    return img


@needs_ref
def test_map_image_round_trip_with_the_reference(built, genome, tmp_path):
    """bwa_idx2mem: our packer writes the image the reference's packer writes (pointer fields aside); bwa_mem2idx: each
    library attaches the other's image; the reference aligns the golden reads from OUR `.map` file to the same SAM."""
    from mpibwa_amd import abi, api
    ours, ref = api.load_library(), po.ref_lib()
    prefix = genome["prefix"].encode()
    ref.bwa_idx2mem.argtypes = [C.POINTER(abi.bwaidx_t)]
    ref.bwa_mem2idx.argtypes = [C.c_int64, C.c_void_p, C.POINTER(abi.bwaidx_t)]
    imgs = []
    for lib in (ours, ref):
        idx = lib.bwa_idx_load_from_disk(prefix, 7)
        assert lib.bwa_idx2mem(idx) == 0
        n = int(idx.contents.l_mem)
        assert n > 0 and C.addressof(idx.contents.bwt.contents) == C.addressof(idx.contents.mem.contents)
        imgs.append(np.ctypeslib.as_array(idx.contents.mem, shape=(n,)).copy())
    assert len(imgs[0]) == len(imgs[1])
    # our loader leaves anno = "" where the reference keeps the FASTA comment "(null)" replaced alike: compare scrubbed images
    a, b = _scrub(imgs[0], None), _scrub(imgs[1], None)
    # cnt_table is derived data both sides fill the same way; everything must be equal
    assert (a == b).all(), np.flatnonzero(a != b)[:10]

    # the file mpiBWAIdx would write, by the product's one-call packer
    map_path = str(tmp_path / "g.fa.map")
    assert ours.mi355x_write_map(prefix, map_path.encode()) == 0
    filed = np.fromfile(map_path, dtype=np.uint8)
    assert (filed == a).all()

    # cross attach: the reference attaches our file and aligns; we attach the reference's image and read the same fields
    rimg = filed.copy()
    ridx = abi.bwaidx_t()
    assert ref.bwa_mem2idx(len(rimg), rimg.ctypes.data, C.byref(ridx)) == 0
    oimg = imgs[1].copy()
    oidx = abi.bwaidx_t()
    assert ours.bwa_mem2idx(len(oimg), oimg.ctypes.data, C.byref(oidx)) == 0
    assert int(oidx.bwt.contents.seq_len) == int(ridx.bwt.contents.seq_len)
    assert int(oidx.bns.contents.n_seqs) == int(ridx.bns.contents.n_seqs) == 3
    for k in range(3):
        assert oidx.bns.contents.anns[k].name == ridx.bns.contents.anns[k].name
        assert oidx.bns.contents.anns[k].offset == ridx.bns.contents.anns[k].offset
    assert C.addressof(oidx.pac.contents) - oimg.ctypes.data == C.addressof(ridx.pac.contents) - rimg.ctypes.data

    from mpibwa_amd import simulate
    reads = simulate.reads_to_ascii(simulate.simulate_reads(genome["seqs"], 60, 150, paired=True, seed=3))
    disk = po.RefIndex(genome["prefix"])
    want = disk.process(disk.opt(flag=abi.MEM_F_PE), reads)
    batch = abi.SeqBatch(po.libc, reads)
    ref.mem_process_seqs(disk.opt(flag=abi.MEM_F_PE), ridx.bwt, ridx.bns, ridx.pac, 0, batch.n, batch.arr, None)
    assert batch.take_sam() == want


@needs_ref
def test_alt_file_is_read_like_the_reference_and_travels_in_the_map_image(built, genome_alt):
    """`.alt` next to the index (src/bntseq.c:179-204): same is_alt flags from our loader and the reference's, and the
    flags survive bwa_idx2mem / bwa_mem2idx (src/bwa.c:310-386) in both directions."""
    from mpibwa_amd import abi, api
    ours, ref = api.load_library(), po.ref_lib()
    prefix = genome_alt["prefix"].encode()
    ref.bwa_idx2mem.argtypes = [C.POINTER(abi.bwaidx_t)]
    ref.bwa_mem2idx.argtypes = [C.c_int64, C.c_void_p, C.POINTER(abi.bwaidx_t)]
    want = [1 if n in genome_alt["alt"] else 0 for n in genome_alt["names"]]
    imgs = []
    for lib in (ours, ref):
        idx = lib.bwa_idx_load_from_disk(prefix, 7)
        n = int(idx.contents.bns.contents.n_seqs)
        assert [int(idx.contents.bns.contents.anns[i].is_alt) for i in range(n)] == want
        assert lib.bwa_idx2mem(idx) == 0
        imgs.append(np.ctypeslib.as_array(idx.contents.mem, shape=(int(idx.contents.l_mem),)).copy())
    for lib, img in ((ours, imgs[1]), (ref, imgs[0])):
        back = abi.bwaidx_t()
        img = img.copy()
        assert lib.bwa_mem2idx(len(img), img.ctypes.data, C.byref(back)) == 0
        assert [int(back.bns.contents.anns[i].is_alt) for i in range(len(want))] == want


@pytest.mark.skipif(not have_ref_src, reason="the reference's driver sources only exist in the build container")
def test_reference_driver_objects_link_against_the_product(built, tmp_path):
    """mpiBWA's own main (mainParallel.c) with parallel_aux.c, fixmate.c, tokenizer.c and the I/O helpers it keeps
    (utils.c, kstring.c, malloc_wrap.c), compiled from the reference tree: whatever they leave undefined besides libc,
    libm, libz, pthread and MPI must be exported by libmpibwa_amd.so — and the real link must succeed."""
    from mpibwa_amd import api
    api.load_library()
    lib = os.path.join(ROOT, "mpibwa_amd", "libmpibwa_amd.so")
    mpi_inc = "/opt/conda/include"
    if not os.path.exists(os.path.join(mpi_inc, "mpi.h")):
        pytest.skip("no MPI headers in this container")
    objs = []
    for f in ("mainParallel", "parallel_aux", "fixmate", "tokenizer", "utils", "kstring", "malloc_wrap"):
        o = str(tmp_path / (f + ".o"))
        subprocess.check_call(["gcc", "-O1", "-w", '-DVERSION="1.5.5"', "-DUSE_MALLOC_WRAPPERS", "-DHAVE_PTHREAD", "-I", REF, "-I", mpi_inc,
                               "-c", os.path.join(REF, f + ".c"), "-o", o])
        objs.append(o)

    def syms(args):
        out = subprocess.check_output(["nm"] + args, text=True)
        return [ln.split() for ln in out.splitlines() if ln.strip() and not ln.endswith(":")]
    undefined = {t[-1] for t in syms(["-u"] + objs) if t[0] == "U"}
    defined_here = {t[-1] for t in syms(["--defined-only"] + objs) if len(t) == 3}
    exported = {t[-1] for t in syms(["-D", "--defined-only", lib]) if len(t) == 3}
    need = {s for s in undefined - defined_here if not s.startswith(("MPI_", "MPIR_"))}
    from_product = need & exported
    assert {"mem_process_seqs", "mem_opt_init", "bwa_fill_scmat", "bwa_set_rg", "bwa_insert_header", "bwa_mem2idx", "bwa_verbose"} <= from_product
    # what is left must come from the C library, libm, zlib or pthread: link for real to prove it
    exe = str(tmp_path / "mpiBWA_amd")
    r = subprocess.run(["gcc", "-o", exe] + objs + [lib, "/opt/conda/lib/libmpi.so", "-lz", "-lm", "-lpthread",
                                                     "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath-link,/usr/lib/x86_64-linux-gnu"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_declared_symbols_are_exported(built):
    """every function include/mpibwa_amd.h declares is in the dynamic symbol table (no compute call without a GPU)"""
    import re
    from mpibwa_amd import api
    lib = api.load_library()
    hdr = open(os.path.join(ROOT, "include", "mpibwa_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b((?:mi355x|bwa|mem)_[a-z0-9_]+)\s*\(", hdr))
    names -= {"mem_pestat_t"}
    assert len(names) >= 35
    for n in sorted(names):
        getattr(lib, n)
    assert set(api.EXPORTS) <= names | {"bwa_verbose", "bwa_rg_id"}


IDX_TOOL = os.path.join(ROOT, "mpibwa_amd", "mpibwa_idx")
REF_IDX_TOOL = os.path.join(ROOT, "oracle", "_ref", "mpiBWAIdx_ref")


def test_mpibwa_idx_writes_the_image_of_the_reference_mpibwaidx_program(built, tmp_path):
    """mpibwa_amd/mpibwa_idx (driver/mpibwa_idx.c), the counterpart of the reference's third program: on the reference's example genome
    its REF.fa.map is the file the reference's own mpiBWAIdx (src/pidx.c compiled in place, oracle/_ref/mpiBWAIdx_ref) writes — the heap
    addresses the packing process leaves in the pointer fields aside — and --build makes the five bwa files of the example from the FASTA,
    byte for byte."""
    import hashlib
    import tarfile
    assert os.path.exists(IDX_TOOL)
    for who in ("own", "ref", "built"):
        os.makedirs(tmp_path / who)
        with tarfile.open(os.path.join(ROOT, "tests", "golden", "mpibwa_examples", "hg19.small.tar.gz")) as t:
            t.extractall(tmp_path / who)
    env = dict(os.environ)
    env.pop("LD_LIBRARY_PATH", None)
    fa = lambda who: str(tmp_path / who / "hg19.small.fa")
    r = subprocess.run([IDX_TOOL, fa("own")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    own = np.fromfile(fa("own") + ".map", dtype=np.uint8)
    if os.path.exists(REF_IDX_TOOL):
        r = subprocess.run([REF_IDX_TOOL, fa("ref")], capture_output=True, text=True, env=env)
        assert r.returncode == 0, r.stderr
        theirs = np.fromfile(fa("ref") + ".map", dtype=np.uint8)
        assert len(own) == len(theirs) and (_scrub(own, None) == _scrub(theirs, None)).all()
        assert (own != theirs).sum() <= 8 * (5 + 2)   # (bwt, sa, anns, ambs, fp_pac and one contig's name / anno pointers)
    # --build: the index files from the FASTA alone
    shipped = {}
    for ext in ("amb", "ann", "bwt", "pac", "sa"):
        shipped[ext] = hashlib.md5(open(fa("built") + "." + ext, "rb").read()).hexdigest()
        os.remove(fa("built") + "." + ext)
    r = subprocess.run([IDX_TOOL, "--build", fa("built")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    for ext, md5 in shipped.items():
        assert hashlib.md5(open(fa("built") + "." + ext, "rb").read()).hexdigest() == md5, ext
    assert (_scrub(np.fromfile(fa("built") + ".map", dtype=np.uint8), None) == _scrub(own, None)).all()
    assert subprocess.run([IDX_TOOL], capture_output=True, env=env).returncode == 1
