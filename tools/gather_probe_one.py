"""One random-gather dispatch of known size in the SMEM kernel's access shape (calibration point for FETCH_SIZE)."""
import ctypes as C, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpibwa_amd import api
lib = api.load_library()
lib.mi355x_gather_probe.restype = C.c_double
lib.mi355x_gather_probe.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_double)]
ms = C.c_double(0)
shape, dep, size, wpc, iters = 1, 1, 3100000000, 16, 2000
g = lib.mi355x_gather_probe(shape, dep, size, wpc, iters, C.byref(ms))
print(json.dumps({"shape": "quad, 2x8 B/lane", "table_bytes": size, "GBps": g, "ms": ms.value, "bytes": g * 1e9 * ms.value * 1e-3}))
