"""Random 64-B gather ceiling on MI355X in the access shapes of the FM-index kernels."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpibwa_amd import api
lib = api.load_library()
lib.mi355x_gather_probe.restype = C.c_double
lib.mi355x_gather_probe.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]
names = {0: "quad, 16 B/lane", 1: "quad, 2x8 B/lane", 2: "lane, 4x16 B/lane"}
size = int(float(sys.argv[1]) * 1e9) if len(sys.argv) > 1 else 3100000000
only = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2]
for shape in only:
    for dep in (0, 1):
        for wpc in (8, 16, 32):
            ms = C.c_double(0)
            iters = 2000 if shape != 2 else 500
            g = lib.mi355x_gather_probe(shape, dep, size, wpc, iters, C.byref(ms))
            print("%-20s dep=%d waves/CU=%2d  %8.1f GB/s  (%.2f ms)" % (names[shape], dep, wpc, g, ms.value), flush=True)
