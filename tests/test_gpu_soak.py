"""Soak: 60 chunks (after two warm-up rounds on the largest chunk) of four different sizes from four caller threads in flight.  Device memory
must stay where it is, the heap's bytes in use must not grow (a leak: mallinfo2 over all arenas — the SAM strings are allocated by the
library's helper threads and freed by the caller), and the resident set may only move by what the library's own counter of work-buffer
reallocations explains: a call context that meets a larger chunk late regrows its device and page-locked buffers in one step of
a few hundred MB (seen as 3 879 -> 4 081 MB between two marks of tools/soak_small.py, with the heap in use flat at 300 MB); without
such an event the bound is 150 MB over the last 30 chunks."""
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


class _Mallinfo2(C.Structure):
    _fields_ = [(k, C.c_size_t) for k in ("arena", "ordblks", "smblks", "hblks", "hblkhd", "usmblks", "fsmblks", "uordblks", "fordblks", "keepcost")]


def _heap_in_use_mb():
    libc = C.CDLL("libc.so.6")
    libc.mallinfo2.restype = _Mallinfo2
    m = libc.mallinfo2()
    return (m.uordblks + m.hblkhd) / 1e6


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
    # (the warm-up rounds run the largest chunk on every caller, each on its own seqs[]: every call context then owns buffers of that size)
    warm = [abi.SeqBatch(api.libc, simulate.reads_to_ascii(simulate.simulate_reads(genome["seqs"], 20000 + 1500 * 3, 150, paired=True, seed=63))) for _ in range(4)]
    lib.mi355x_buffer_growths.restype = C.c_ulonglong
    steps, lock, todo = 60, threading.Lock(), iter(range(60))
    marks = {}

    gate = threading.Barrier(4)

    def caller(t):
        b = warm[t]
        for _ in range(2):   # two rounds started together: all four call contexts are in use at once and get their work buffers
            gate.wait(timeout=600)
            eng.process_batch(opt, b)
            p = lib.mi355x_collect_sam(b.arr, b.n, C.byref(C.c_size_t(0)))
            api.libc.free(C.c_void_p(p))
        b = batches[t]
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
                    marks[s] = (_rss_mb(), dev_used(), _heap_in_use_mb(), int(lib.mi355x_buffer_growths()))
    th = [threading.Thread(target=caller, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    (rss_mid, dev_mid, heap_mid, grow_mid), (rss_end, dev_end, heap_end, grow_end) = marks[29], marks[steps - 1]
    assert abs(dev_end - dev_mid) < 64, (dev_mid, dev_end)          # MB: no device allocation after the first rounds
    assert heap_end - heap_mid < 80, (heap_mid, heap_end)           # MB of heap in use over the last 30 chunks (the marks fall anywhere inside four calls: +-35 MB seen): nothing leaks
    regrown = grow_end - grow_mid                                   # work buffers reallocated between the marks (each one a step of the resident set)
    # (the leak check is the line above; this one only catches pages that are neither heap in use nor a counted reallocation — a new 64-MB
    # malloc arena or two for helper threads that start late are within it)
    assert rss_end - rss_mid < 150 + (400 if regrown else 0), (rss_mid, rss_end, regrown)
