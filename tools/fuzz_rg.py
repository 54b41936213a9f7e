"""Random -R / -H strings through bwa_set_rg and bwa_insert_header of the library and of the reference (src/bwa.c:413-476): same result or same
refusal, same bwa_rg_id; run on the sanitizer build like tools/fuzz_sampost.py (30 000 strings: equal, no finding)."""
import ctypes as C, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpibwa_amd import api
from oracle import pyoracle as po
lib = api.load_library(); ref = po.ref_lib()
for L in (lib, ref):
    L.bwa_set_rg.restype = C.c_void_p; L.bwa_set_rg.argtypes = [C.c_char_p]
    L.bwa_insert_header.restype = C.c_void_p; L.bwa_insert_header.argtypes = [C.c_char_p, C.c_void_p]
    C.c_int.in_dll(L, "bwa_verbose").value = 0
po.libc.strdup.restype = C.c_void_p; po.libc.strdup.argtypes = [C.c_char_p]
rnd = random.Random(3)
alpha = b"@RGIDSM:\\tn\tab x1-_"
n_ok = 0
for it in range(30000):
    s = bytes(rnd.choice(alpha) for _ in range(rnd.randrange(0, 40)))
    if rnd.random() < 0.5: s = b"@RG\\tID:" + s
    s = s.replace(b"\0", b"")
    a, b = lib.bwa_set_rg(s), ref.bwa_set_rg(s)
    assert (a is None) == (b is None), s
    if a:
        assert C.string_at(a) == C.string_at(b), s
        assert bytes((C.c_char * 256).in_dll(lib, "bwa_rg_id").raw).split(b"\0")[0] == bytes((C.c_char * 256).in_dll(ref, "bwa_rg_id").raw).split(b"\0")[0], s
        n_ok += 1
        po.libc.free(C.c_void_p(a)); po.libc.free(C.c_void_p(b))
    h = bytes(rnd.choice(alpha) for _ in range(rnd.randrange(1, 30)))
    prev = rnd.choice([None, b"@CO\tx", b"@HD\tVN:1\n@CO\ty"])
    pa = po.libc.strdup(prev) if prev else None; pb = po.libc.strdup(prev) if prev else None
    x, y = lib.bwa_insert_header(h, pa), ref.bwa_insert_header(h, pb)
    assert (x is None) == (y is None), (h, prev)
    if x:
        assert C.string_at(x) == C.string_at(y), (h, prev)
        po.libc.free(C.c_void_p(x)); po.libc.free(C.c_void_p(y))
print("read groups accepted", n_ok)
