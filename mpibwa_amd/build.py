"""In-tree build of libmpibwa_amd.so (HIP kernels for gfx950 + host C++), no JIT cache.

    python -m mpibwa_amd.build          # incremental
    python -m mpibwa_amd.build --force
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libmpibwa_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    srcs = sources()
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp", ".cuh"))]
    hdrs.append(os.path.join(HERE, "..", "include", "mpibwa_amd.h"))
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    extra = os.environ.get("MPIBWA_CXXFLAGS", "").split()
    common = extra + ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-g1", "-Wall", "-Wno-unused-function", "-Wno-unused-result",
              "-I", CSRC, "-I", os.path.join(HERE, "..", "include")]
    jobs = []
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            if s.endswith(".hip"):
                cmd = [HIPCC, f"--offload-arch={ARCH}", "-x", "hip"] + common + ["-c", s, "-o", o]
            else:
                cmd = [HIPCC, "-x", "c++"] + common + ["-c", s, "-o", o]
            jobs.append(cmd)
    if jobs:   # independent translation units: a few compilers side by side (MPIBWA_BUILD_JOBS, default: the CPUs at hand, at most 8)
        from concurrent.futures import ThreadPoolExecutor
        n_par = int(os.environ.get("MPIBWA_BUILD_JOBS", "0")) or max(1, min(8, len(os.sched_getaffinity(0))))

        def run(cmd):
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=n_par) as ex:
            list(ex.map(run, jobs))
    if force or _stale(OUT, objs):
        cmd = [HIPCC, "-shared", "-o", OUT] + objs + ["-Wl,-Bsymbolic", "-Wl,-soname,libmpibwa_amd.so", "-lpthread", "-lm", "-ldl", "-lz"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    build_driver(force=force, verbose=verbose)
    build_idx_tool(force=force, verbose=verbose)
    return OUT


def build_idx_tool(force=False, verbose=False):
    """mpibwa_amd/mpibwa_idx: the counterpart of the reference's mpiBWAIdx (driver/mpibwa_idx.c): REF.fa.map from the bwa index files,
    --build for the files themselves.  Plain C on include/mpibwa_amd.h, no MPI."""
    src = os.path.join(HERE, "driver", "mpibwa_idx.c")
    exe = os.path.join(HERE, "mpibwa_idx")
    if force or _stale(exe, [src, OUT, os.path.join(HERE, "..", "include", "mpibwa_amd.h")]):
        cmd = ["gcc", "-O2", "-std=gnu99", "-Wall", "-I", os.path.join(HERE, "..", "include"), src, "-o", exe, "-L", HERE, "-lmpibwa_amd",
               "-Wl,-rpath,/usr/lib/x86_64-linux-gnu:$ORIGIN", "-Wl,-rpath-link,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return exe


def build_driver(force=False, verbose=False):
    """mpibwa_amd/mpibwa_gpu: the MPI host program (driver/mpibwa_gpu.c), when an MPI installation is around (MPI_HOME or the
    image's MPICH under /opt/conda).  Plain C against include/mpibwa_amd.h; the system library directory goes first in its run
    path so that the product library's libstdc++ is the system's, not an older one next to the MPI library."""
    src = os.path.join(HERE, "driver", "mpibwa_gpu.c")
    exe = os.path.join(HERE, "mpibwa_gpu")
    home = os.environ.get("MPI_HOME", "/opt/conda")
    inc, libdir = os.path.join(home, "include"), os.path.join(home, "lib")
    if not (os.path.exists(os.path.join(inc, "mpi.h")) and os.path.exists(os.path.join(libdir, "libmpi.so"))):
        return None
    if force or _stale(exe, [src, OUT, os.path.join(HERE, "..", "include", "mpibwa_amd.h")]):
        cmd = ["gcc", "-O2", "-std=gnu99", "-Wall", "-I", os.path.join(HERE, "..", "include"), "-I", inc, src, "-o", exe,
               "-L", HERE, "-lmpibwa_amd", os.path.join(libdir, "libmpi.so"),
               "-Wl,-rpath,/usr/lib/x86_64-linux-gnu:$ORIGIN:" + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-lpthread", "-lm"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return exe


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
