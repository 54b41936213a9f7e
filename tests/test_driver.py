"""mpibwa_amd/mpibwa_gpu (driver/mpibwa_gpu.c), the product's own MPI host program: its multi-rank FASTQ partition (byte
slices -> record boundaries -> the chunk rule carried from rank to rank -> replicated chunk table) must give the chunk table
of the single-process rule (mi355x_fastq_chunks, itself pinned to the reference's chunk structure in test_fastq.py) for any
number of ranks.  CPU only: --dry-run stops before the index is touched.  The aligned output is checked in
test_gpu_driver.py."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "mpibwa_amd", "mpibwa_gpu")
EX = os.path.join(ROOT, "tests", "golden", "mpibwa_examples")


def mpiexec():
    for p in (shutil.which("mpiexec"), "/opt/conda/bin/mpiexec"):
        if p and os.path.exists(p):
            return p
    return None


def _dry(ranks, args, cwd):
    env = dict(os.environ)
    env.pop("LD_LIBRARY_PATH", None)
    r = subprocess.run([mpiexec(), "-n", str(ranks), EXE, "mem", "--dry-run"] + args, capture_output=True, text=True, timeout=300, env=env, cwd=cwd)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    head = lines[0].split()
    table = [tuple(int(x) for x in ln.split()[1::2]) for ln in lines[1:]]
    return dict(zip(head[0::2], head[1::2])), table


def _write_fastq(path, names, seqs, quals=None):
    with open(path, "wb") as f:
        for i, (n, s) in enumerate(zip(names, seqs)):
            q = quals[i] if quals is not None else b"I" * len(s)
            f.write(b"@" + n + b"\n" + s + b"\n+\n" + q + b"\n")


@pytest.mark.skipif(mpiexec() is None, reason="no mpiexec in this container")
def test_partition_is_the_single_process_chunk_rule_for_any_rank_count(built, tmp_path):
    from mpibwa_amd import api, fastq
    if not os.path.exists(EXE):
        pytest.skip("mpibwa_gpu not built (no MPI installation)")
    lib = api.load_library()
    rng = np.random.default_rng(3)
    # (a) the reference's example pair (equal sizes: lockstep mode), small -K -> many chunks
    fq = []
    for k in (1, 2):
        dst = str(tmp_path / ("R%d.fastq" % k))
        with gzip.open(os.path.join(EX, "HCC1187C_R%d_10K.fastq.gz" % k), "rb") as g, open(dst, "wb") as f:
            f.write(g.read())
        fq.append(dst)
    # (b) ragged pairs: different lengths per mate, quality lines that start with '@' and '+', no final newline in R2
    n = 3000
    names = [b"q%d" % i for i in range(n)]
    s1 = [bytes(rng.choice(list(b"ACGT"), int(rng.integers(30, 160))).tolist()) for _ in range(n)]
    s2 = [bytes(rng.choice(list(b"ACGT"), int(rng.integers(30, 160))).tolist()) for _ in range(n)]
    q1 = [bytes([64]) + bytes(rng.choice(list(b"@+IJ#"), len(s) - 1).tolist()) for s in s1]
    t1, t2 = str(tmp_path / "T1.fastq"), str(tmp_path / "T2.fastq")
    _write_fastq(t1, names, s1, q1)
    _write_fastq(t2, names, s2)
    with open(t2, "rb+") as f:
        f.seek(-1, 2); f.truncate()
    cases = [(fq, 1_000_000, "pe"), (fq, 200_000, "pe"), ([fq[0]], 300_000, "se"), ([t1, t2], 50_000, "pe_trim"), ([t1], 20_000, "se")]
    for files, K, mode in cases:
        src = fastq.FastqSource(lib, files[0], files[1] if len(files) > 1 else None, K=K)
        assert src.mode == mode
        want_first = [int(x) for x in src.starts]
        want_off1 = [int(src.f1.off[i]) for i in want_first]
        for ranks in (1, 2, 3, 5):
            head, table = _dry(ranks, ["-K", str(K), "PREFIX_UNUSED"] + files, str(tmp_path))
            assert head["mode"] == mode and int(head["chunks"]) == src.n_chunks and int(head["reads"]) == src.f1.n
            assert [t[1] for t in table] == want_first, (files, K, ranks)
            assert [t[2] for t in table] == want_off1
            if len(files) > 1:
                assert [t[3] for t in table] == [int(src.f2.off[i]) for i in want_first]


@pytest.fixture(scope="module")
def shim_env(built, tmp_path_factory):
    """environment that puts tests/csrc/driver_shim.c and the reference's library under a driver (see that file's header)"""
    from oracle import pyoracle as po
    if mpiexec() is None or not os.path.exists(EXE) or not po.ref_available():
        pytest.skip("mpiexec, mpibwa_gpu or oracle/_ref/libbwaref.so not present")
    shim = str(tmp_path_factory.mktemp("shim") / "driver_shim.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-o", shim, os.path.join(ROOT, "tests", "csrc", "driver_shim.c"), "-ldl"])
    return {"LD_PRELOAD": shim, "MPIBWA_TEST_REFLIB": os.path.join(ROOT, "oracle", "_ref", "libbwaref.so")}


def test_output_options_on_the_cpu_with_the_reference_under_the_driver(shim_env, genome, tmp_path, tmp_path_factory):
    """-f, -g, -b and --by-chr of mpibwa_gpu end to end without a GPU: tests/csrc/driver_shim.c (LD_PRELOAD) answers the driver's device
    entry points and hands mem_process_seqs to the reference's own library, so the driver's host side runs as it does on the GPU box
    and the checks of tests/test_gpu_driver.py apply unchanged (there the records come from the product's kernels)."""
    import tarfile
    import test_gpu_driver as g
    d = tmp_path_factory.mktemp("drv_cpu")
    with tarfile.open(os.path.join(EX, "hg19.small.tar.gz")) as t:
        t.extractall(d)
    fq = []
    for k in (1, 2):
        dst = str(d / ("R%d.fastq" % k))
        with gzip.open(os.path.join(EX, "HCC1187C_R%d_10K.fastq.gz" % k), "rb") as gz, open(dst, "wb") as f:
            f.write(gz.read())
        fq.append(dst)
    g._EXTRA_ENV.update(shim_env)
    try:
        g.check_output_options((str(d), os.path.join(str(d), "hg19.small.fa"), fq), genome, tmp_path)
    finally:
        g._EXTRA_ENV.clear()


REF_MAIN = os.path.join(ROOT, "oracle", "_ref", "mpiBWA_amd")
REF_BYCHR = os.path.join(ROOT, "oracle", "_ref", "mpiBWAByChr_amd")


def _records(path):
    return sorted(ln for ln in open(path, "rb").read().splitlines(keepends=True) if not ln.startswith(b"@"))


@pytest.mark.skipif(not (os.path.exists(REF_MAIN) and os.path.exists(REF_BYCHR)), reason="the reference's programs are not built (oracle/_ref)")
def test_fixmate_and_by_chromosome_files_are_those_of_the_reference_programs(shim_env, genome, tmp_path):
    """The reference's own programs — main() of mainParallel.c and of mainParallelByChromosome.c, compiled in place (oracle/Makefile) — and
    mpibwa_gpu on the same input with the same library under both: `-f` gives the same records, `--by-chr` (with and without -f) the same
    files with the same records and as many header lines, for pairs of every kind on three contigs.  Their -b / -g writers are not a
    yardstick: they lose the last read of every thread's slice (src/parallel_aux.c:2953-2956), which this test shows; ours lose nothing
    (tests/test_gpu_driver.py::check_output_options)."""
    from mpibwa_amd import api, simulate
    from test_sampost import _pairs_of_every_kind
    d = str(tmp_path)
    prefix = os.path.join(d, "g.fa")
    for ext in ("", ".amb", ".ann", ".bwt", ".pac", ".sa"):
        os.symlink(genome["prefix"] + ext, prefix + ext)
    assert api.load_library().mi355x_write_map(prefix.encode(), (prefix + ".map").encode()) == 0   # (the reference's programs attach PREFIX.map)
    reads = _pairs_of_every_kind(genome, n=3000, seed=14)
    fq = [os.path.join(d, "r1.fastq"), os.path.join(d, "r2.fastq")]
    simulate.write_fastq(fq[0], reads, 0)
    simulate.write_fastq(fq[1], reads, 1)
    env = dict(os.environ)
    env.pop("LD_LIBRARY_PATH", None)
    env.update(shim_env)

    def run(exe, args):
        r = subprocess.run([mpiexec(), "-n", "2", exe, "mem", "-t", "4", "-K", "300000"] + args + [prefix] + fq, capture_output=True, text=True, timeout=900, env=env, cwd=d)
        assert r.returncode == 0, r.stderr[-3000:]
    run(REF_MAIN, ["-o", os.path.join(d, "ref_plain")])
    plain = _records(os.path.join(d, "ref_plain.sam"))
    run(EXE, ["-o", os.path.join(d, "own_plain.sam")])
    assert _records(os.path.join(d, "own_plain.sam")) == plain and len(plain) > 6000
    # -x: the reference's preset of the options the user left alone (src/mainParallel.c:398-428), beside options the user did set
    run(REF_MAIN, ["-x", "intractg", "-B", "7", "-o", os.path.join(d, "ref_x")])
    run(EXE, ["-x", "intractg", "-B", "7", "-o", os.path.join(d, "own_x.sam")])
    preset = _records(os.path.join(d, "ref_x.sam"))
    assert _records(os.path.join(d, "own_x.sam")) == preset and preset != plain
    # --ordered: one rank with four chunks in flight writes the file one chunk in flight writes (record for record, unsorted)
    bodies = []
    for extra in (["--in-flight", "1"], ["--in-flight", "4", "--ordered"]):
        o = os.path.join(d, "own_%s.sam" % extra[1])
        r = subprocess.run([mpiexec(), "-n", "1", EXE, "mem", "-K", "100000"] + extra + ["-o", o, prefix] + fq, capture_output=True, text=True, timeout=900, env=env, cwd=d)
        assert r.returncode == 0, r.stderr[-3000:]
        bodies.append([ln for ln in open(o, "rb").read().splitlines(keepends=True) if not ln.startswith(b"@")])
    assert bodies[0] == bodies[1] and len(bodies[0]) > 6000
    # -f
    run(REF_MAIN, ["-f", "-o", os.path.join(d, "ref_f")])
    run(EXE, ["-f", "-o", os.path.join(d, "own_f.sam")])
    fixed = _records(os.path.join(d, "ref_f.sam"))
    assert _records(os.path.join(d, "own_f.sam")) == fixed and fixed != plain
    # the per-contig files, with the discordant file (no -f) and without (-f)
    for extra, src, files in (([], plain, ["chrS1", "chrS2", "chrS3", "discordant", "unmapped"]), (["-f"], fixed, ["chrS1", "chrS2", "chrS3", "unmapped"])):
        tag = "f" if extra else "p"
        for who, exe, more in (("ref", REF_BYCHR, []), ("own", EXE, ["--by-chr"])):
            os.makedirs(os.path.join(d, who + tag))
            run(exe, extra + more + ["-o", os.path.join(d, who + tag, "x.sam")])
        assert sorted(os.listdir(os.path.join(d, "ref" + tag))) == sorted(os.listdir(os.path.join(d, "own" + tag))) == [f + ".sam" for f in files]
        n = 0
        for f in files:
            a, b = os.path.join(d, "ref" + tag, f + ".sam"), os.path.join(d, "own" + tag, f + ".sam")
            ra = _records(a)
            assert ra == _records(b) and len(ra) > 100, f
            assert open(a, "rb").read().count(b"\n@") == open(b, "rb").read().count(b"\n@")
            n += len(ra) if f != "discordant" else 0
        assert n == len(src)
    # trimmed pairs (files of different sizes: the chunk rule runs over both files; mpiBWAByChr passes n_processed = 0 in every mode, so its
    # records do not depend on which rank gets which chunk), with -f.  The reference's trimmed branch stops on an assertion of its own read
    # count with two ranks: one there.
    rng = np.random.default_rng(15)
    trimmed = [(n, a, b[:int(rng.integers(60, 151))]) for n, a, b in reads[:1500]]
    tq = [os.path.join(d, "t1.fastq"), os.path.join(d, "t2.fastq")]
    simulate.write_fastq(tq[0], trimmed, 0)
    simulate.write_fastq(tq[1], trimmed, 1)
    names = ["chrS1", "chrS2", "chrS3", "unmapped"]
    for who, exe, more, ranks in (("ref", REF_BYCHR, [], "1"), ("own", EXE, ["--by-chr"], "2")):
        os.makedirs(os.path.join(d, who + "t"))
        r = subprocess.run([mpiexec(), "-n", ranks, exe, "mem", "-t", "4", "-K", "200000", "-f"] + more + ["-o", os.path.join(d, who + "t", "x.sam"), prefix] + tq,
                           capture_output=True, text=True, timeout=900, env=env, cwd=d)
        assert r.returncode == 0, r.stderr[-3000:]
    assert sorted(os.listdir(os.path.join(d, "reft"))) == sorted(os.listdir(os.path.join(d, "ownt"))) == [f + ".sam" for f in names]
    for f in names:
        ra = _records(os.path.join(d, "reft", f + ".sam"))
        assert ra == _records(os.path.join(d, "ownt", f + ".sam")) and len(ra) > 50, f
    # trimmed pairs through mpiBWA itself: n_processed = the reads the rank has done before the chunk (src/mainParallel.c:2355-2357) seeds the hash
    # tie-breaks; one rank each (with more ranks it depends on which rank gets which chunk), three chunks in flight on our side
    for exe, out, more in ((REF_MAIN, os.path.join(d, "ref_trim"), []), (EXE, os.path.join(d, "own_trim.sam"), ["--in-flight", "3"])):
        r = subprocess.run([mpiexec(), "-n", "1", exe, "mem", "-t", "4", "-K", "100000"] + more + ["-o", out, prefix] + tq, capture_output=True, text=True, timeout=900, env=env, cwd=d)
        assert r.returncode == 0, r.stderr[-3000:]
    trimmed_records = _records(os.path.join(d, "ref_trim.sam"))
    assert _records(os.path.join(d, "own_trim.sam")) == trimmed_records and len(trimmed_records) > 3000
    # single end: the reference's mpiBWAByChr is no yardstick there (its single-end writer loses 17 to a few hundred of 4 997 records from run
    # to run and sometimes ends on a signal); ours are the records of the reference's mpiBWA on the same file, each in the file of its RNAME
    r = subprocess.run([mpiexec(), "-n", "2", REF_MAIN, "mem", "-t", "4", "-K", "200000", "-o", os.path.join(d, "ref_se"), prefix, fq[0]],
                       capture_output=True, text=True, timeout=900, env=env, cwd=d)
    assert r.returncode == 0, r.stderr[-3000:]
    os.makedirs(os.path.join(d, "owns"))
    r = subprocess.run([mpiexec(), "-n", "2", EXE, "mem", "-K", "200000", "--by-chr", "-o", os.path.join(d, "owns", "x.sam"), prefix, fq[0]],
                       capture_output=True, text=True, timeout=900, env=env, cwd=d)
    assert r.returncode == 0, r.stderr[-3000:]
    se = _records(os.path.join(d, "ref_se.sam"))
    assert sorted(os.listdir(os.path.join(d, "owns"))) == [f + ".sam" for f in names]
    for f in names:
        want = [ln for ln in se if ln.split(b"\t")[2] == (b"*" if f == "unmapped" else f.encode())]
        assert _records(os.path.join(d, "owns", f + ".sam")) == want and len(want) > 50, f
    # the reference's compressed writers lose records
    run(REF_MAIN, ["-b", "-o", os.path.join(d, "ref_b")])
    theirs = [ln for ln in gzip.decompress(open(os.path.join(d, "ref_b.bam"), "rb").read()).splitlines(keepends=True) if not ln.startswith(b"@")]
    assert 0 < len(theirs) < len(plain) and set(theirs) <= set(plain)


OPTION_SETS = [
    ["-A", "2"],                                                      # -A alone scales what the user left alone (src/mainParallel.c:430-440)
    ["-A", "2", "-B", "5", "-T", "45", "-O", "5,7", "-E", "2,3", "-L", "3,9", "-U", "11"],
    ["-k", "23", "-w", "60", "-c", "100", "-d", "70", "-r", "1.2", "-D", "0.6", "-m", "10", "-s", "5", "-G", "3000", "-N", "20", "-W", "25", "-y", "8", "-X", "0.4"],
    ["-h", "3,9", "-a", "-Y", "-5", "-q"],
    ["-I", "300,50,600,100", "-M", "-V", "-C", "-v", "2"],
    ["-P", "-S", "-j", "-Q", "30", "-R", "@RG\\tID:x1\\tSM:s"],
]


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="the reference's program is not built (oracle/_ref)")
@pytest.mark.parametrize("k", range(len(OPTION_SETS)))
def test_option_letters_mean_what_they_mean_to_the_reference_program(shim_env, genome, tmp_path, k):
    """Every `mem` option of the reference's getopt string (src/mainParallel.c:291-398) through both programs' own parsers — mpibwa_gpu and the
    reference's main() compiled in place — with the same library under both: the same records.  Pairs of every kind with comments on the read
    names (-C), on three contigs."""
    from mpibwa_amd import api, simulate
    from test_sampost import _pairs_of_every_kind
    d = str(tmp_path)
    prefix = os.path.join(d, "g.fa")
    for ext in ("", ".amb", ".ann", ".bwt", ".pac", ".sa"):
        os.symlink(genome["prefix"] + ext, prefix + ext)
    assert api.load_library().mi355x_write_map(prefix.encode(), (prefix + ".map").encode()) == 0
    reads = simulate.reads_to_ascii(_pairs_of_every_kind(genome, n=500, seed=50 + k))
    fq = [os.path.join(d, "r1.fastq"), os.path.join(d, "r2.fastq")]
    for which in (0, 1):
        with open(fq[which], "wb") as f:
            for name, a, b in reads:
                sq = (a, b)[which]
                f.write(b"@" + name.encode() + b" BC:Z:ACGT\n" + sq + b"\n+\n" + b"I" * len(sq) + b"\n")
    env = dict(os.environ)
    env.pop("LD_LIBRARY_PATH", None)
    env.update(shim_env)
    outs = []
    for exe, out in ((REF_MAIN, os.path.join(d, "ref")), (EXE, os.path.join(d, "own.sam"))):
        r = subprocess.run([mpiexec(), "-n", "1", exe, "mem", "-t", "4", "-K", "150000"] + OPTION_SETS[k] + ["-o", out, prefix] + fq,
                           capture_output=True, text=True, timeout=900, env=env, cwd=d)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append(_records(out if out.endswith(".sam") else out + ".sam"))
    assert outs[0] == outs[1] and len(outs[0]) >= 1000, (OPTION_SETS[k], len(outs[0]), len(outs[1]))
