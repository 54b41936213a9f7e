"""Soak: 60 chunks (after two warm-up rounds) of four different sizes from four caller threads in flight.  Device memory must stay where it is once
every call context has seen its largest chunk, and the resident set must not creep (the SAM strings are allocated by the
library's helper threads and freed by the caller: per-thread malloc arenas are where such a creep comes from)."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rss_mb():
    for l in open("/proc/self/status"):
        if l.startswith("VmRSS"):
            return int(l.split()[1]) / 1e3
    return 0.0


def test_sixty_chunks_leave_memory_where_it_was(genome, built):
    from mpibwa_amd import abi, api, simulate
    lib = api.load_library()
    lib.mi355x_finalize()
    eng = api.Engine(genome["prefix"], device=0)
    def dev_used():   # (through the library's own HIP runtime: the test runner's process may carry a second one, e.g. an imported torch's)
        fr, tot = C.c_size_t(0), C.c_size_t(0)
        assert lib.mi355x_device_memory(C.byref(fr), C.byref(tot)) == 0
        return (tot.value - fr.value) / 1e6
    C.c_int.in_dll(lib, "bwa_verbose").value = 1
    opt = eng.opt(flag=abi.MEM_F_PE)
    batches = [abi.SeqBatch(api.libc, simulate.reads_to_ascii(simulate.simulate_reads(genome["seqs"], 20000 + 1500 * k, 150, paired=True, seed=60 + k)))
               for k in range(4)]
    steps, lock, todo = 60, threading.Lock(), iter(range(60))
    marks = {}

    gate = threading.Barrier(4)

    def caller(t):
        b = batches[t]
        for _ in range(2):   # two rounds started together: all four call contexts are in use at once and get their work buffers
            gate.wait(timeout=600)
            eng.process_batch(opt, b)
            p = lib.mi355x_collect_sam(b.arr, b.n, C.byref(C.c_size_t(0)))
            api.libc.free(C.c_void_p(p))
        while True:
            with lock:
                s = next(todo, None)
            if s is None:
                return
            eng.process_batch(opt, b)
            n = C.c_size_t(0)
            p = lib.mi355x_collect_sam(b.arr, b.n, C.byref(n))
            assert n.value > 0
            api.libc.free(C.c_void_p(p))
            if s in (29, steps - 1):
                with lock:
                    marks[s] = (_rss_mb(), dev_used())
    th = [threading.Thread(target=caller, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    (rss_mid, dev_mid), (rss_end, dev_end) = marks[29], marks[steps - 1]
    assert abs(dev_end - dev_mid) < 64, (dev_mid, dev_end)          # MB: no device allocation after the first rounds
    assert rss_end - rss_mid < 50, (rss_mid, rss_end)               # MB over the last 30 chunks
