"""Mutated SAM text (truncated lines, missing tabs and newlines, random bytes, absurd numbers, empty texts) through the passes behind the call
— mi355x_fixmate_pair, mi355x_route_by_chr, mi355x_bgzf_compress — on the sanitizer build of the host sources:
    tools/san_host.sh --collect-only >/dev/null; MPIBWA_SANITIZER_LIB=/tmp/mpibwa_san/libhost_san.so LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
        ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=halt_on_error=1 FUZZ_SEED=11 FUZZ_N=25000 python tools/fuzz_sampost.py
Nothing may crash or trip a sanitizer; text that is not a pair's is refused (-1) and left alone."""
import ctypes as C, gzip, json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from mpibwa_amd import abi, api
from golden_util import G, golden_index
import tempfile
lib = api.load_library()
eng = api.Engine(golden_index(tempfile.mkdtemp()), upload=False)
gold = json.load(gzip.open(os.path.join(G, "fixmate_cases.json.gz"), "rt"))
libc = api.libc
libc.strdup.restype = C.c_void_p; libc.strdup.argtypes = [C.c_char_p]
rnd = random.Random(int(os.environ.get("FUZZ_SEED", "5")))
def mutate(t):
    b = bytearray(t)
    k = rnd.randrange(8)
    if not b: return bytes(b)
    if k == 0: del b[rnd.randrange(len(b)):]                       # truncate
    elif k == 1:
        i = rnd.randrange(len(b)); del b[i:i + rnd.randrange(1, 40)]  # cut a piece
    elif k == 2: b = b.replace(b"\t", b" ", rnd.randrange(1, 5))   # tabs gone
    elif k == 3: b = b.replace(b"\n", b"", 1)                       # a newline gone
    elif k == 4:
        for _ in range(rnd.randrange(1, 6)): b[rnd.randrange(len(b))] = rnd.randrange(256) or 1
    elif k == 5: b = b"\n" * rnd.randrange(0, 3) + b
    elif k == 6: b = bytearray(b"x\t99999999999999999999\t*\t-5\t1e9\t*\t=\t\t\t\t\n")
    elif k == 7: b = bytearray(b"")
    return bytes(b).replace(b"\0", b"?")
n_ok = n_bad = 0
for it in range(int(os.environ.get("FUZZ_N", "6000"))):
    c = gold["cases"][rnd.randrange(len(gold["cases"]))]
    t1, t2 = c["in"][0].encode(), c["in"][1].encode()
    if rnd.random() < 0.7: t1 = mutate(t1)
    if rnd.random() < 0.7: t2 = mutate(t2)
    nm = C.create_string_buffer(c["name"].encode())
    arr = (abi.bseq1_t * 2)()
    for k, t in enumerate((t1, t2)):
        arr[k].name = C.addressof(nm); arr[k].sam = libc.strdup(t)
    r = lib.mi355x_fixmate_pair(C.byref(arr[0]), C.byref(arr[1]), eng.bns)
    n_ok += r >= 0; n_bad += r < 0
    text = C.string_at(arr[0].sam) + C.string_at(arr[1].sam)
    for k in range(2): libc.free(C.c_void_p(arr[k].sam))
    nd = eng.bns.contents.n_seqs + 2
    o = (C.c_void_p * nd)(); l = (C.c_size_t * nd)()
    rr = lib.mi355x_route_by_chr(text, len(text), eng.bns, 1, o, l)
    if rr >= 0:
        for d in range(nd):
            if o[d]: libc.free(C.c_void_p(o[d]))
    cap = lib.mi355x_bgzf_bound(len(text)); z = C.create_string_buffer(max(cap, 1))
    lib.mi355x_bgzf_compress(text, len(text), rnd.choice([-1, 0, 1, 9, 77]), z, cap)
print("fixmate accepted", n_ok, "refused", n_bad)
