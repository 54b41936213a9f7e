"""Soak run: many chunks through mem_process_seqs, resident set size and device memory watched for growth."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mpibwa_amd import abi, api, bigindex
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 150      # 250: BASELINE config 4's shape
n_callers = int(sys.argv[3]) if len(sys.argv) > 3 else 4         # caller threads in flight (the library runs up to twelve calls side by side)
pairs0 = int(sys.argv[4]) if len(sys.argv) > 4 else 200000
os.makedirs("/tmp/mpibwa_bench", exist_ok=True)
idx = bigindex.make_or_get("/tmp/mpibwa_bench", genome_mbp=3100, seed=38, log=lambda *a: None)
eng = idx.engine
lib = eng.lib
C.c_int.in_dll(lib, "bwa_verbose").value = 1
opt = eng.opt(flag=abi.MEM_F_PE, n_threads=int(lib.mi355x_host_cpus()))
batches = [abi.SeqBatch(api.libc, idx.simulate_pairs(pairs0 + 1111 * k, seed=50 + k, read_len=read_len, frag_mean=max(400.0, 2.2 * read_len)))
           for k in range(n_callers)]   # one chunk per caller, different sizes
def rss():
    for l in open("/proc/self/status"):
        if l.startswith("VmRSS"):
            return int(l.split()[1]) / 1e6
import threading
t0 = time.time()
lock = threading.Lock()
todo = iter(range(steps))
def caller(t):   # four callers in flight, each with its own chunk
    b = batches[t]
    while True:
        with lock:
            s = next(todo, None)
        if s is None:
            return
        eng.process_batch(opt, b)
        n = C.c_size_t(0)
        p = lib.mi355x_collect_sam(b.arr, b.n, C.byref(n))
        api.libc.free(C.c_void_p(p))
        if s in (8, 32, 120, 240, 360, steps - 1) or s == steps // 2:
            free, total = torch.cuda.mem_get_info()
            print("step %d: RSS %.2f GB, device used %.2f GB, %.1f s" % (s, rss(), (total - free) / 1e9, time.time() - t0), flush=True)
th = [threading.Thread(target=caller, args=(t,)) for t in range(n_callers)]   # disjoint seqs[] per caller, as the ABI asks
for x in th: x.start()
for x in th: x.join()
