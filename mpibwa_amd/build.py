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
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            if s.endswith(".hip"):
                cmd = [HIPCC, f"--offload-arch={ARCH}", "-x", "hip"] + common + ["-c", s, "-o", o]
            else:
                cmd = [HIPCC, "-x", "c++"] + common + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.check_call(cmd)
    if force or _stale(OUT, objs):
        cmd = [HIPCC, "-shared", "-o", OUT] + objs + ["-Wl,-Bsymbolic", "-Wl,-soname,libmpibwa_amd.so", "-lpthread", "-lm", "-ldl"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
