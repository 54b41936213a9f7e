"""The drop-in boundary on the GPU: index attached from mpiBWA's `.map` image, residency through mi355x_init() (RCCL), the
resident-index check, and the reference's own unmodified driver (its main(), MPI-IO FASTQ chunking and SAM writer, built
by oracle/Makefile into oracle/_ref/mpiBWA_amd) running end to end on the product library."""
import ctypes as C
import gzip
import hashlib
import os
import shutil
import subprocess
import sys
import tarfile

import pytest

from golden_util import golden_index, load_reads, load_sam, sam_cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "tests", "golden", "mpibwa_examples")
KNOWN_MD5 = "51ce7ba0592d4a199eac49526b6c9d8c"   # SAM body of the real mpiBWA on the example data


@pytest.fixture(scope="module")
def gold(tmp_path_factory, built):
    return golden_index(tmp_path_factory.mktemp("gold_b"))


def _golden_ok(eng):
    kw = sam_cases()["pe_default"]
    return b"".join(eng.process(eng.opt(**kw), load_reads("reads_pe150.tsv.gz"))) == load_sam("pe_default")


def test_index_attached_from_map_image(gold):
    from mpibwa_amd import api
    lib = api.load_library()
    lib.mi355x_finalize()
    assert lib.mi355x_write_map(gold.encode(), (gold + ".map").encode()) == 0
    eng = api.Engine(None, device=0, map_path=gold + ".map")
    assert _golden_ok(eng)


def test_mi355x_init_broadcasts_through_rccl(gold):
    """A communicator of one rank still takes the RCCL path: ncclGetUniqueId -> the caller's bootstrap broadcast ->
    ncclCommInitRank -> ncclBroadcast of the three index arrays in place -> commit.  In a process of its own, like a rank of
    mpibwa_gpu: the test runner's process may carry a second HIP runtime (an imported torch brings its own librccl and
    libamdhip64), and a communicator created by that pair cannot take this library's device buffers."""
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from mpibwa_amd import api
from golden_util import load_reads, load_sam, sam_cases
seen = []
def bcast(buf, nbytes, root, user):
    seen.append((int(nbytes), int(root)))
cb = api.mi355x_comm_t.BCAST(bcast)
comm = api.mi355x_comm_t(0, 1, cb, None)
eng = api.Engine(%r, device=0, comm=comm)
assert seen == [(128, 0), (24, 0)], seen      # the RCCL bootstrap id, then rank 0's checksums of the three index arrays
assert eng.bcast_seconds is not None and eng.bcast_seconds > 0
kw = sam_cases()["pe_default"]
assert b"".join(eng.process(eng.opt(**kw), load_reads("reads_pe150.tsv.gz"))) == load_sam("pe_default")
print("RCCL_PATH_OK")
""" % (ROOT, os.path.join(ROOT, "tests"), gold)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_PATH_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_mi355x_init_refuses_an_index_that_differs_from_rank_0s(gold):
    """After its broadcast every rank hashes the three index arrays on its device and compares with rank 0's numbers, which travel
    through the caller's host broadcast: numbers that differ (here: bent on the way, as a rank with a damaged copy would see them)
    end the process with a message instead of letting it align against a damaged index."""
    code = r"""
import sys, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, %r)
from mpibwa_amd import api
def bcast(buf, nbytes, root, user):
    if int(nbytes) == 24:
        C.cast(buf, C.POINTER(C.c_uint64))[1] ^= 1
cb = api.mi355x_comm_t.BCAST(bcast)
comm = api.mi355x_comm_t(0, 1, cb, None)
eng = api.Engine(%r, device=0, comm=comm)
print("NOT_REACHED")
""" % (ROOT, os.path.join(ROOT, "tests"), gold)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "NOT_REACHED" not in r.stdout and "damaged index" in r.stderr and "sampled SA" in r.stderr, (r.stdout[-500:], r.stderr[-2000:])


def test_a_call_with_another_index_than_the_resident_one_aborts(gold, genome):
    """Two indexes in one process: switching with mi355x_finalize() + upload works in both directions; a call that passes
    an index which is not the resident one must die instead of aligning against the wrong reference."""
    from mpibwa_amd import api
    lib = api.load_library()
    lib.mi355x_finalize()
    a = api.Engine(gold, device=0)
    assert _golden_ok(a)
    lib.mi355x_finalize()
    b = api.Engine(genome["prefix"], device=0)
    kw = sam_cases()["pe_default"]
    assert b"".join(b.process(b.opt(**kw), load_reads("reads_pe150.tsv.gz"))) != load_sam("pe_default")
    lib.mi355x_finalize()
    a2 = api.Engine(gold, device=0, upload=False)
    assert lib.mi355x_index_upload(0, a2.bwt, a2.bns, a2.pac) == 0
    assert _golden_ok(a2)
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from mpibwa_amd import api\n"
            "from golden_util import load_reads, sam_cases\n"
            "a = api.Engine(%r, device=0)\n"
            "b = api.Engine(%r, device=0, upload=False)\n"
            "b.process(b.opt(**sam_cases()['pe_default']), load_reads('reads_pe150.tsv.gz')[:4])\n"
            "print('SURVIVED')\n") % (ROOT, os.path.join(ROOT, "tests"), gold, genome["prefix"])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "SURVIVED" not in r.stdout
    assert "not the one resident" in r.stderr


def _mpiexec():
    for p in (shutil.which("mpiexec"), "/opt/conda/bin/mpiexec"):
        if p and os.path.exists(p):
            return p
    return None


DRIVER = os.path.join(ROOT, "oracle", "_ref", "mpiBWA_amd")


@pytest.mark.skipif(not os.path.exists(DRIVER) or _mpiexec() is None, reason="oracle/_ref/mpiBWA_amd or mpiexec not present")
@pytest.mark.parametrize("ranks,threads", [(2, 8), (1, 4)])
def test_reference_driver_runs_end_to_end_on_the_product(tmp_path, built, ranks, threads):
    """examples/standard.sh with the reference's own main(): `mpiexec -n 2 mpiBWA mem -t 8 -o out.sam hg19.small.fa R1 R2`,
    the binary being mpiBWA's unmodified driver linked against libmpibwa_amd.so; the `.map` image comes from the
    product's packer.  The SAM body must hash to what the real mpiBWA wrote."""
    from mpibwa_amd import api
    lib = api.load_library()
    lib.mi355x_finalize()                         # the ranks below bring their own residency
    with tarfile.open(os.path.join(EX, "hg19.small.tar.gz")) as t:
        t.extractall(tmp_path)
    prefix = str(tmp_path / "hg19.small.fa")
    assert lib.mi355x_write_map(prefix.encode(), (prefix + ".map").encode()) == 0
    fq = []
    for k in (1, 2):
        dst = str(tmp_path / ("R%d.fastq" % k))
        with gzip.open(os.path.join(EX, "HCC1187C_R%d_10K.fastq.gz" % k), "rb") as g, open(dst, "wb") as f:
            f.write(g.read())
        fq.append(dst)
    out = str(tmp_path / "out")                    # the driver appends ".sam" (src/mainParallel.c:702)
    env = dict(os.environ)
    env.pop("LD_LIBRARY_PATH", None)              # the binary's run path is complete
    r = subprocess.run([_mpiexec(), "-n", str(ranks), DRIVER, "mem", "-t", str(threads), "-o", out, prefix] + fq,
                       capture_output=True, text=True, timeout=900, env=env, cwd=str(tmp_path))
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-2000:] + "\n" + r.stderr[-4000:])
    assert r.returncode == 0
    sam = open(out + ".sam", "rb").read()
    body = b"".join(ln for ln in sam.splitlines(keepends=True) if not ln.startswith(b"@"))
    assert body.count(b"\n") == 20036
    # with two ranks the shared file pointer interleaves their blocks: the reference's own known answer is order-free
    # only when everything fits one chunk, which is the case here (one chunk -> one rank writes everything)
    assert hashlib.md5(body).hexdigest() == KNOWN_MD5
    assert b"@SQ\tSN:" in sam[:4096]
