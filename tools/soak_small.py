"""tests/test_gpu_soak.py's workload with marks every ten chunks: resident set, the heap's bytes in use and the heap's size
(mallinfo2 over all arenas) and device memory — tells a leak (bytes in use grow) from allocator slack (only the arenas grow).
    python tools/soak_small.py [chunks=120] [callers=4]"""
import ctypes as C, os, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mpibwa_amd import abi, api, simulate

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
n_callers = int(sys.argv[2]) if len(sys.argv) > 2 else 4


class Mallinfo2(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in ("arena", "ordblks", "smblks", "hblks", "hblkhd", "usmblks", "fsmblks", "uordblks", "fordblks", "keepcost")]


libc = C.CDLL("libc.so.6")
libc.mallinfo2.restype = Mallinfo2


def heap():
    m = libc.mallinfo2()
    return (m.uordblks + m.hblkhd) / 1e6, (m.arena + m.hblkhd) / 1e6


def rss():
    for l in open("/proc/self/status"):
        if l.startswith("VmRSS"):
            return int(l.split()[1]) / 1e3


d = tempfile.mkdtemp(prefix="soak_small")
names, seqs = simulate.make_genome(360_000, 3, seed=7)
fa = os.path.join(d, "g.fa")
simulate.write_fasta(fa, names, seqs)
api.build_index(fa, fa)
lib = api.load_library()
eng = api.Engine(fa, device=0)
C.c_int.in_dll(lib, "bwa_verbose").value = 1
opt = eng.opt(flag=abi.MEM_F_PE)
batches = [abi.SeqBatch(api.libc, simulate.reads_to_ascii(simulate.simulate_reads(seqs, 20000 + 1500 * k, 150, paired=True, seed=60 + k))) for k in range(n_callers)]


def dev_used():
    fr, tot = C.c_size_t(0), C.c_size_t(0)
    lib.mi355x_device_memory(C.byref(fr), C.byref(tot))
    return (tot.value - fr.value) / 1e6


lock, todo, gate, t0 = threading.Lock(), iter(range(steps)), threading.Barrier(n_callers), time.time()


def caller(t):
    b = batches[t]
    for _ in range(2):
        gate.wait(timeout=600)
        eng.process_batch(opt, b)
        api.libc.free(C.c_void_p(lib.mi355x_collect_sam(b.arr, b.n, C.byref(C.c_size_t(0)))))
    while True:
        with lock:
            s = next(todo, None)
        if s is None:
            return
        eng.process_batch(opt, b)
        api.libc.free(C.c_void_p(lib.mi355x_collect_sam(b.arr, b.n, C.byref(C.c_size_t(0)))))
        if s % 10 == 9:
            with lock:
                u, a = heap()
                print("chunk %3d: RSS %.1f MB, heap in use %.1f MB, heap size %.1f MB, device %.1f MB, %.1f s" % (s + 1, rss(), u, a, dev_used(), time.time() - t0), flush=True)


th = [threading.Thread(target=caller, args=(t,)) for t in range(n_callers)]
for x in th:
    x.start()
for x in th:
    x.join()
