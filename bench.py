#!/usr/bin/env python3
"""bench.py — throughput of the mem_process_seqs() hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one call of mem_process_seqs() (through the C ABI of libmpibwa_amd.so) on one chunk of
synthetic 2x150 bp paired-end reads, i.e. exactly what mpiBWA's main loop does per chunk
(src/mainParallel.c:1314).  One process per GPU; for N > 1 the driver launches this script under
torch.distributed.run and the reads are sharded (independent chunks per rank, no data-path collective;
the only collective is the one-off RCCL broadcast of the index from rank 0, as the north star asks).

The JSON line carries, besides the contract fields,
  roofline      — the SMEM seeding kernel: algorithmic bytes (64 B per occ block touched + read + output,
                  SURVEY.md §8d, counted on the device) / its HIP-event time, against 8 TB/s HBM3E
  cpu_baseline  — the reference's own mem_process_seqs (oracle/_ref/libbwaref.so, compiled from the
                  reference sources) on a bounded sample of the same reads, on this box's host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def cgroup_throttle():
    """(events, microseconds) the container's CPU quota has stopped this process group so far (cgroup v2 cpu.stat)."""
    ev = us = 0
    try:
        for l in open("/sys/fs/cgroup/cpu.stat"):
            k, v = l.split()
            if k == "nr_throttled":
                ev = int(v)
            if k == "throttled_usec":
                us = int(v)
    except OSError:
        pass
    return ev, us


def kernel_sources_sha256(root):
    """what tools/pmc_smem.sh records next to its FETCH_SIZE figures: the seeding kernels' sources"""
    import hashlib
    h = hashlib.sha256()
    for f in ("fm_kernels.hip", "smem_kernels.hip"):
        h.update(open(os.path.join(root, "mpibwa_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)   # (three rounds of the eight calls in flight: with 12 the start-up and the tail of the timed region weigh too much)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome-mbp", type=float, default=float(os.environ.get("MPIBWA_BENCH_GENOME_MBP", "3100")))
    ap.add_argument("--pairs", type=int, default=int(os.environ.get("MPIBWA_BENCH_PAIRS", "333334")),
                    help="read pairs per step per GPU (mpiBWA -K 100000000 closes a chunk at 10^8 bases)")
    ap.add_argument("--cpu-sample-pairs", type=int, default=int(os.environ.get("MPIBWA_BENCH_CPU_PAIRS", "0")),
                    help="pairs given to the reference for the CPU baseline and the parity check (0 = the whole step batch)")
    ap.add_argument("--read-len", type=int, default=int(os.environ.get("MPIBWA_BENCH_READ_LEN", "150")),
                    help="read length (150 = the headline config; 250 = BASELINE config 4's shape, a parity case)")
    ap.add_argument("--seed", type=int, default=1000, help="seed of the read simulator (rank r, chunk c use seed + 16 r + c)")
    ap.add_argument("--chunks", type=int, default=int(os.environ.get("MPIBWA_BENCH_CHUNKS", "3")),
                    help="distinct chunks per rank the steps cycle through (config 1: 1 M pairs = 3 chunks of 333 334)")
    ap.add_argument("--repeat-frac", type=float, default=float(os.environ.get("MPIBWA_BENCH_REPEAT_FRAC", "0.05")),
                    help="share of the synthetic genome covered by planted repeat families (SURVEY §8d config 1: 0.05; real GRCh38 is ~0.5: see README)")
    ap.add_argument("--genome-model", default=os.environ.get("MPIBWA_BENCH_GENOME_MODEL", "grch38like"), choices=["uniform", "grch38like"],
                    help="grch38like (default since round 4): SURVEY §8d's generator — order-3 Markov base composition, GC 41 %, repeat families of "
                         "up to 10^4 copies, 0-5 % divergence; uniform: i.i.d. bases + planted repeats of 2-40 copies (the headline workload of rounds 1-3)")
    ap.add_argument("--alt-legs", default=os.environ.get("MPIBWA_BENCH_ALT_LEGS", "grch38like:0.5,uniform:0.05"),
                    help="after the headline (N = 1 only): the same bench, shortened, on these other references (model:repeat_frac,...), each in a "
                         "child process of its own so that the headline line survives whatever happens there; '' = none")
    ap.add_argument("--quick", action="store_true", help="no one-call-in-flight leg and no counting pass (the alt legs run this way: no roofline figures)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check-parity", action="store_true",
                    help="every rank also aligns its chunks with the reference (oracle/_ref) and compares the SAM: `parity_all_ranks` "
                         "(N > 1 runs carry no CPU baseline; for small configurations)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("MPIBWA_BENCH_IN_FLIGHT", "8")),
                    help="caller threads inside mem_process_seqs at once (the library runs up to eight calls side by side: "
                         "the GPU-bound half of one chunk overlaps the host-bound half of the previous one)")
    ap.add_argument("--workdir", default=os.environ.get("MPIBWA_BENCH_DIR", "/tmp/mpibwa_bench"))
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    backend = os.environ.get("MPIBWA_BENCH_BACKEND", "nccl")   # "gloo" only for dry runs of the multi-rank path on one GPU
    n_dev = torch.cuda.device_count()
    dev = local_rank % max(n_dev, 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    torch.cuda.set_device(dev)

    # ---- the same bench, shortened, on other references: measured by this run, each in a child process of its own, BEFORE this
    # process touches the GPU (a child has the whole device, and the headline survives whatever happens there, e.g. a timeout);
    # attached to the line as alt_workloads, not part of `value` ----
    alt_workloads = None
    if rank == 0 and world == 1 and args.alt_legs and not args.quick:
        import subprocess
        alt_workloads = []
        n_fly_legs = max(1, min(args.in_flight, 12))
        for spec in args.alt_legs.split(","):
            model, frac = spec.split(":")
            if model == args.genome_model and float(frac) == args.repeat_frac:
                continue
            cmd = [sys.executable, os.path.abspath(__file__), "--quick", "--alt-legs", "", "--genome-model", model, "--repeat-frac", frac, "--chunks", "2",
                   "--steps", "8", "--warmup", "2", "--in-flight", str(n_fly_legs), "--cpu-sample-pairs", "30000", "--genome-mbp", str(args.genome_mbp),
                   "--pairs", str(args.pairs), "--read-len", str(args.read_len), "--workdir", args.workdir]
            t_leg = time.perf_counter()
            leg = {"workload": spec, "command": " ".join(cmd[1:])}
            try:
                r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=float(os.environ.get("MPIBWA_BENCH_LEG_TIMEOUT", "300")))
                lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
                if lines:
                    d = json.loads(lines[-1])
                    leg.update({k: d.get(k) for k in ("value", "unit", "ms_per_step", "steps", "host_cpu_s_per_step", "host_cpu_busy_frac", "timed_region_disturbances",
                                                      "parity_on_sample", "sam_records_written_by_device_frac", "work_per_step")})
                    leg["workload"] = d["config"]["workload"]
                    leg["calls_in_flight"] = d["config"]["calls_in_flight"]
                    leg["cpu_baseline"] = {k: d["cpu_baseline"].get(k) for k in ("value", "unit", "cores", "kind", "sample")} if isinstance(d.get("cpu_baseline"), dict) else None
                else:
                    leg["error"] = "exit code %d: %s" % (r.returncode, r.stderr.decode()[-400:])
            except Exception as e:
                leg["error"] = repr(e)[:600]
            leg["wall_s"] = round(time.perf_counter() - t_leg, 1)
            alt_workloads.append(leg)
            log("alt leg %s: %s" % (spec, {k: leg.get(k) for k in ("value", "parity_on_sample", "error", "wall_s")}))

    from mpibwa_amd import abi, api, bigindex, simulate
    from mpibwa_amd.build import build
    if rank == 0:
        build()
    if world > 1:
        dist.barrier()
    lib = api.load_library(build_if_missing=False)

    # ---- reference genome + index: built once by rank 0 on its GPU, broadcast over RCCL/xGMI ----
    os.makedirs(args.workdir, exist_ok=True)
    t0 = time.time()
    idx = bigindex.make_or_get(args.workdir, genome_mbp=args.genome_mbp, seed=38, rank=rank, world=world,
                               local_rank=dev, dist=dist if world > 1 else None, log=log if rank == 0 else None,
                               repeat_frac=args.repeat_frac, model=args.genome_model)
    eng = idx.engine
    if rank == 0:
        log("index ready in %.1f s: l_pac=%d, occ blocks %.2f GB, SA %.2f GB" %
            (time.time() - t0, idx.l_pac, idx.blk_bytes / 1e9, idx.sa_bytes / 1e9))

    # ---- reads: seeded per rank and chunk, same generator as the tests (2 % unmappable, 1 % subst., 0.1 % indel) ----
    # Config 1 is 1 M pairs: `--chunks` distinct chunks of `--pairs` pairs; step s aligns chunk s mod n_chunks, so the
    # arenas, branch histories and cached index blocks of consecutive steps differ as they do in a real run.
    n_chunks = max(1, args.chunks)
    chunk_reads = [idx.simulate_pairs(args.pairs, seed=args.seed + 16 * rank + c, read_len=args.read_len, frag_mean=max(400.0, 2.2 * args.read_len))
                   for c in range(n_chunks)]
    import hashlib
    import threading
    n_fly = max(1, min(args.in_flight, 12))
    # every caller thread owns a bseq1_t[] (and its .sam) per chunk
    batches = [[abi.SeqBatch(api.libc, chunk_reads[c]) for c in range(n_chunks)] for _ in range(n_fly)]
    cores = int(lib.mi355x_host_cpus())
    opt = eng.opt(flag=abi.MEM_F_PE, n_threads=cores)
    lib_verbose = C.c_int.in_dll(eng.lib, "bwa_verbose")
    lib_verbose.value = 1   # keep the per-chunk stderr chatter out of the timed region

    pending = []   # (thread, seqs[i].sam pointers) of finished steps; the caller (mpiBWA's writer thread, src/mainParallel.c:103-127) owns them
    lock = threading.Lock()

    step_no = [0]   # steps handed out so far (all phases): step s aligns chunk s mod n_chunks

    def run_steps(k_steps, acc, fly=None, by_chunk=None, in_step=False):
        """Exactly k_steps calls of mem_process_seqs, at most `fly` (default n_fly) of them in flight.  in_step (the warm-up): the
        caller threads start their calls together, round by round, so that all `fly` call contexts of the library are in use at
        once and get their work buffers before the timed region (otherwise the eighth context may see its first chunk there)."""
        fly = n_fly if fly is None else fly
        if in_step:
            k_steps = (k_steps + fly - 1) // fly * fly
        first = step_no[0]
        step_no[0] += k_steps
        todo = iter(range(first, first + k_steps))
        gate = threading.Barrier(fly) if in_step and fly > 1 else None

        def worker(t):
            torch.cuda.set_device(dev)
            while True:
                if gate is not None:
                    gate.wait(timeout=600)
                with lock:
                    s_ = next(todo, None)
                    if s_ is None:
                        return
                c = s_ % n_chunks
                b = batches[t][c]   # (its .sam pointers of an earlier, undrained step were copied into `pending`)
                eng.process_batch(opt, b)
                st = eng.stats()                                          # this thread's last call
                with lock:
                    pending.append((t, c, b._rec["sam"].copy()))          # hand the output over, one pointer per read
                    for k, v in st.items():
                        acc[k] = acc.get(k, 0) + v
                    if by_chunk is not None:
                        by_chunk[c] = st

        th = [threading.Thread(target=worker, args=(t,)) for t in range(1, fly)]
        for x in th:
            x.start()
        worker(0)
        for x in th:
            x.join()

    digests = [set() for _ in range(n_chunks)]   # md5 of every step's SAM, per chunk
    sam_bytes_seen = [0]
    dg_lock = threading.Lock()

    import xxhash

    def sam_digest(buf):
        return xxhash.xxh3_128_hexdigest(buf)

    def collect_one(entry):
        # concatenate + free one step's SAM strings, as mpiBWA's copy_buffer_thr does (src/mainParallel.c:103-127); on a record array of
        # its own, so that the caller thread may already be aligning the same chunk again
        t, c, ptrs = entry
        tmp = np.zeros(len(ptrs), dtype=abi.SeqBatch.dtype())
        tmp["sam"] = ptrs
        n = C.c_size_t(0)
        p = lib.mi355x_collect_sam(C.cast(tmp.ctypes.data, C.POINTER(abi.bseq1_t)), len(ptrs), C.byref(n))
        d = sam_digest((C.c_char * n.value).from_address(p))
        api.libc.free(C.c_void_p(p))
        with dg_lock:
            digests[c].add(d)
            sam_bytes_seen[0] += n.value

    def drain():
        before = sam_bytes_seen[0]
        while True:
            with lock:
                e = pending.pop(0) if pending else None
            if e is None:
                break
            collect_one(e)
        return sam_bytes_seen[0] - before

    class Writer:
        """The caller's writer thread: takes the finished steps' SAM off the aligner threads' hands while they align the next chunks
        (the reference's copy_buffer_thr runs next to its chunk loop the same way); the timed region contains it as it would a real run."""

        def __init__(self):
            self.stop = False
            self.th = threading.Thread(target=self.run)
            self.th.start()

        def run(self):
            torch.cuda.set_device(dev)
            while True:
                with lock:
                    e = pending.pop(0) if pending else None
                if e is None:
                    if self.stop:
                        return
                    time.sleep(0.002)
                    continue
                collect_one(e)

        def finish(self):
            self.stop = True
            self.th.join()

    run_steps(max(args.warmup, 1) * max(n_fly, n_chunks), {}, in_step=True)   # every call context warms its work buffers
    drain()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    seen0 = sam_bytes_seen[0]
    writer = Writer() if os.environ.get("MPIBWA_BENCH_WRITER", "0") == "1" else None   # (A/B: 8.5-8.7 vs 10.9-13.1 Mreads/s with eight calls in flight: it takes 0.45 CPU-s per step)
    t0 = time.perf_counter()
    c0 = time.process_time()
    sys0 = os.times().system
    grow0 = int(lib.mi355x_buffer_growths())
    if os.environ.get("MPIBWA_GROWTH_LOG"):
        print("[bench] timed region starts", file=sys.stderr, flush=True)
    thr0 = cgroup_throttle()
    acc = {}
    run_steps(args.steps, acc)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    host_cpu_s = time.process_time() - c0
    host_sys_s = os.times().system - sys0
    grown = int(lib.mi355x_buffer_growths()) - grow0
    if os.environ.get("MPIBWA_GROWTH_LOG"):
        print("[bench] timed region ends", file=sys.stderr, flush=True)
    thr1 = cgroup_throttle()
    if writer:
        writer.finish()      # what the writer had not got to yet is collected outside the timed region
    drain()
    sam_bytes = sam_bytes_seen[0] - seen0
    # the latency mode next to it, outside the timed region: one call in flight, every chunk once (the kernels' durations when
    # they have the GPU to themselves; inside the timed region a launch shares the chip with the kernels of the other calls)
    alone = {}
    alone_s = None
    if not args.quick:
        time.sleep(2.2)   # the library treats a caller as a busy one for two seconds after it last had three calls in flight
        run_steps(1, {}, fly=1)   # (the lone caller's mode has work buffers of its own: first touch outside the figure)
        drain()
        ta = time.perf_counter()
        run_steps(n_chunks, alone, fly=1)
        torch.cuda.synchronize()
        alone_s = time.perf_counter() - ta
        drain()
    # the algorithmic bytes of the seeding kernel are a property of the reads (64 B per occ block the reference
    # touches + read + output): counted on the device by the counting variant of the kernel, every chunk once, outside the
    # timed region (the production variant leaves the per-extension block arithmetic out); a timed step on chunk c moved
    # exactly those bytes
    if not args.quick:
        os.environ["MPIBWA_SMEM_COUNT"] = "1"
        counted = {}
        run_steps(n_chunks, {}, fly=1, by_chunk=counted)
        os.environ["MPIBWA_SMEM_COUNT"] = "0"
        drain()
        first_timed = step_no[0] - 2 * n_chunks - args.steps
        acc["smem_bytes"] = sum(counted[s_ % n_chunks]["smem_bytes"] for s_ in range(first_timed, first_timed + args.steps))
        alone["smem_bytes"] = sum(counted[c]["smem_bytes"] for c in range(n_chunks))
    else:
        acc["smem_bytes"] = 0
    # the same kernel pair with the chip to itself and a whole chunk per launch (the production variant through the stage-level entry;
    # no call in flight): what the kernel does when nothing else competes for the CUs' outstanding misses — the timed region's figure
    # shares the chip with seven other calls' kernels, the lone caller's launches hold half a chunk.  Never part of `value`; a failure
    # here only leaves the field out.
    smem_alone = None
    if not args.quick and rank == 0 and os.environ.get("MPIBWA_BENCH_SMEM_ALONE", "1") != "0":
        try:
            tr = bytes.maketrans(b"ACGTN", bytes([0, 1, 2, 3, 4]))
            ends = [e.translate(tr) for _, a, b in chunk_reads[0] for e in (a, b)]
            flat = np.frombuffer(b"".join(ends), dtype=np.uint8)
            off = np.zeros(len(ends) + 1, dtype=np.int64)
            off[1:] = np.cumsum([len(e) for e in ends])
            cap = 96
            out_iv = np.empty((len(ends), cap, 4), dtype=np.uint64)
            out_n = np.zeros(len(ends), dtype=np.int32)
            os.environ["MPIBWA_SMEM_COUNT"] = "0"
            times = []
            for _ in range(4):
                ms, nb = C.c_double(0), C.c_uint64(0)
                if lib.mi355x_smem_batch(opt, len(ends), flat.ctypes.data, off.ctypes.data, cap, out_iv.ctypes.data, out_n.ctypes.data, C.byref(ms), C.byref(nb)) != 0:
                    raise RuntimeError("more than %d intervals for a read" % cap)
                times.append(ms.value)
            ms_alone = sum(times[1:]) / len(times[1:])   # (the first launch pays the allocation of its buffers' pages)
            a_alone = counted[0]["smem_bytes"] / (ms_alone * 1e-3) / 1e9
            smem_alone = {"launch_ms": round(ms_alone, 3), "launches": len(times) - 1, "reads_per_launch": len(ends), "achieved": round(a_alone, 1),
                          "frac": round(a_alone / 8000.0, 4), "algo_bytes_per_launch": int(counted[0]["smem_bytes"])}
            del out_iv, flat, ends
        except Exception as e:   # (measurement extra: the bench line does not depend on it)
            log("WARNING: seeding kernels alone on the chip not measured: %r" % (e,))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_reads_total = 2 * args.pairs * args.steps * world
    value = n_reads_total / elapsed / 1e6

    # ---- roofline of the dominant kernel (rank 0's launches; every rank runs the same kernel on its own shard) ----
    # A chunk is worked off as n_sub sub-batches, i.e. n_sub launches of the kernel per step: per-launch figures are the
    # totals over the timed region divided by the number of launches (HIP events on the launching streams, pipeline.hip).
    ach = acc["smem_bytes"] / (acc["k_smem_ms"] * 1e-3) / 1e9 if acc.get("k_smem_ms") else 0.0
    n_launch = max(1, int(acc.get("n_sub", args.steps)))
    # HBM traffic per launch: PMC counters cannot be collected inside this run (rocprofv3 has to wrap the process), so the
    # per-read figure measured by tools/pmc_smem.sh on this workload is scaled to the reads of one launch — but only while
    # the summary under profiles/ was taken from the kernel sources as they are now (it records their sha256); otherwise null.
    traffic, traffic_src = None, None
    try:
        here = os.path.dirname(os.path.abspath(__file__))
        pj = json.load(open(os.path.join(here, "profiles", "r04_pmc_smem.json")))
        now = kernel_sources_sha256(here)
        if pj.get("kernel_sources_sha256") != now:
            log("WARNING: profiles/r04_pmc_smem.json was measured on other kernel sources (%s..., now %s...): roofline.traffic = null; "
                "re-run tools/pmc_smem.sh" % (str(pj.get("kernel_sources_sha256"))[:12], now[:12]))
        elif args.genome_mbp >= 3000 and args.read_len == 150:
            traffic = int(pj["smem_kernel"]["traffic_bytes_per_read"] * 2 * args.pairs * args.steps / n_launch)
            traffic_src = "profiles/r04_pmc_smem.json (FETCH_SIZE pass of the same workload, per read) x reads per launch"
    except Exception as e:
        log("WARNING: no usable PMC summary for roofline.traffic: %r" % (e,))
    tab = acc.get("smem_tab_bytes", 0)
    roofline = {"kernel": "smem_kernel + smem_p3_kernel (one seeding launch = both)", "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0,
                "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_src,
                "launch_ms": round(acc["k_smem_ms"] / n_launch, 3), "algo_bytes_per_launch": int(acc["smem_bytes"] / n_launch),
                "launches_per_step": round(n_launch / args.steps, 2),
                # the same with the occ blocks left out that the third pass takes from its jump table instead of fetching
                "frac_without_jump_table_blocks": round((acc["smem_bytes"] - tab) / (acc["k_smem_ms"] * 1e-3) / 1e9 / 8000.0, 4) if acc.get("k_smem_ms") else None,
                "note": "achieved = the REFERENCE's occ-block fetches for these reads (SURVEY 8d: 64 B per block bwt_extend touches + read + output, counted on "
                        "the device by the counting variant of the kernel) / the production launches' duration; the production kernel takes results of up to 14 "
                        "bases from k-mer tables and results of the third pass's first 12 steps from a jump table instead of fetching those blocks: the HBM bytes "
                        "it really moves are `traffic` (PMC, ~0.46 x algorithmic)"}
    if smem_alone:
        roofline["alone_whole_chunk"] = smem_alone
    if alone.get("k_smem_ms") and alone.get("n_sub"):
        a1 = alone["smem_bytes"] / (alone["k_smem_ms"] * 1e-3) / 1e9
        roofline["one_call_in_flight"] = {"launch_ms": round(alone["k_smem_ms"] / alone["n_sub"], 3), "achieved": round(a1, 1),
                                          "frac": round(a1 / 8000.0, 4)}

    out = {
        "metric": "Mreads/s (whole node) 2x150 bp PE vs GRCh38-size reference; SAM bit-match",
        "value": round(value, 4), "unit": "Mreads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "config": {"workload": "%d x %d pairs of 2x%d bp PE reads per GPU vs seeded synthetic %.0f Mbp reference (GRCh38 absent on the box; "
                               "%s, %.0f %% of it in planted repeat families%s)" % (n_chunks, args.pairs, args.read_len, idx.l_pac / 1e6,
                                                                                     "uniform base composition" if args.genome_model == "uniform" else "order-3 Markov base composition with GC 41 %",
                                                                                     100 * args.repeat_frac, "" if args.genome_model == "uniform" else " of up to 10^4 copies"),
                   "pairs_per_step_per_gpu": args.pairs, "distinct_chunks": n_chunks, "reference_mbp": round(idx.l_pac / 1e6, 1),
                   "chunking": "one mem_process_seqs chunk per step (mpiBWA -K 1e8 semantics)", "parallelism": "reads sharded, 1 rank/GPU",
                   "calls_in_flight": n_fly,
                   "caller": "aligner threads; the finished chunks' SAM is concatenated, hashed and freed after the timed region (MPIBWA_BENCH_WRITER=1: by a writer thread inside it)"},
        "sam_bytes_per_step": int(sam_bytes / args.steps),
        "sam_records_written_by_device_frac": round(acc.get("n_sam_dev", 0) / max(1, acc.get("n_reads", 1)), 4),
        "one_call_in_flight": {"value": round(2 * args.pairs * n_chunks * world / alone_s / 1e6, 4), "unit": "Mreads/s",
                               "ms_per_step": round(alone_s / n_chunks * 1e3, 2), "steps": n_chunks} if alone_s else None,
        "host_cpu_s_per_step": round(host_cpu_s / args.steps, 3), "host_sys_s_per_step": round(host_sys_s / args.steps, 3),
        "timed_region_disturbances": {"work_buffer_reallocations": grown, "cgroup_cpu_throttle_events": thr1[0] - thr0[0], "cgroup_cpu_throttled_ms": round((thr1[1] - thr0[1]) / 1e3, 1)}, "host_cpu_busy_frac": round(host_cpu_s / (elapsed * max(cores, 1)), 3),
        "roofline": roofline,
        "stage_ms_per_step": {k: round(acc[k] / args.steps, 2) for k in
                              ("total_ms", "h2d_ms", "phase1_ms", "smem_ms", "sa_ms", "chain_ms", "ext_ms", "regs_ms", "pestat_ms", "sam_ms", "msw_ms", "plan_ms", "aln_ms", "emit_ms", "k_smem_ms", "k_sa_ms", "k_ext_ms", "k_msw_ms", "k_aln_ms")},
        "work_per_step": {k: int(acc.get(k, 0) / args.steps) for k in ("n_intv", "n_seeds", "n_chains", "n_ext", "n_msw", "n_aln", "n_pair_dev", "n_sam_dev")},
        "aux_kernels": {
            "sa_kernel_GBps": round(acc["sa_bytes"] / (acc["k_sa_ms"] * 1e-3) / 1e9, 1) if acc.get("k_sa_ms") else None,
            "c2a_kernel_GCUPS": round(acc["ext_cells"] / (acc["k_ext_ms"] * 1e-3) / 1e9, 2) if acc.get("k_ext_ms") else None},
    }
    if acc.get("k_ext_ms"):
        # second roofline: the extension kernel is bound by instruction issue, not by memory.  Ceiling: 256 CUs x 4 SIMDs issue one
        # 64-lane vector instruction per 4 cycles at 2.4 GHz = 39.3 T lane-operations/s; the row loop of wave_ext.cuh spends 81 vector
        # instructions on a strip of 64 cells (54 in the strip, 27 per row; counted in the ISA), i.e. 81 lane-operations
        # per cell when every lane holds a live cell -> 485 GCUPS.  The achieved figure counts the cells the kernel computed.
        g = acc["ext_cells"] / (acc["k_ext_ms"] * 1e-3) / 1e9
        out["roofline_c2a"] = {"kernel": "c2a_kernel", "bound": "valu", "achieved": round(g, 1), "peak": 485.0, "unit": "GCUPS",
                               "frac": round(g / 485.0, 4), "launch_ms_per_step": round(acc["k_ext_ms"] / args.steps, 2)}

    # ---- CPU baseline + parity: the reference itself on this box's host cores, every chunk once, rank 0 at N=1 only ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import pyoracle as po
            if po.ref_available():
                ref = po.RefIndex(idx.prefix)
                C.c_int.in_dll(ref.lib, "bwa_verbose").value = 1
                ropt = ref.opt(flag=abi.MEM_F_PE, n_threads=cores)
                n_ref, t_ref, ok, same = 0, 0.0, True, True
                for c in range(n_chunks):
                    sample = chunk_reads[c][:min(args.cpu_sample_pairs or len(chunk_reads[c]), len(chunk_reads[c]))]
                    rb = abi.SeqBatch(po.libc, sample)
                    t0 = time.perf_counter()
                    ref.lib.mem_process_seqs(ropt, ref.bwt, ref.bns, ref.pac, 0, rb.n, rb.arr, None)
                    t_ref += time.perf_counter() - t0
                    n_ref += rb.n
                    want = rb.take_sam()
                    got = eng.process(opt, sample)
                    ok = ok and got == want
                    if len(sample) == len(chunk_reads[c]):   # every step on this chunk (warm-up included) produced exactly this SAM
                        same = same and digests[c] == {sam_digest(b"".join(want))}
                    else:
                        same = None
                out["cpu_baseline"] = {"value": round(n_ref / t_ref / 1e6, 5), "unit": "Mreads/s", "cores": cores, "kind": "reference",
                                       "sample": "%s of the %d chunks of the run (%d reads), one reference mem_process_seqs -t %d call each (%.1f s in all)" %
                                                 ("all reads" if not args.cpu_sample_pairs else "the first %d pairs of each" % args.cpu_sample_pairs,
                                                  n_chunks, n_ref, cores, t_ref)}
                # the per-core anchor: the reference at -t 1 on a bounded slice of the first chunk
                n1 = min(len(chunk_reads[0]), int(os.environ.get("MPIBWA_BENCH_T1_PAIRS", "30000" if not args.quick else "2000")))
                rb1 = abi.SeqBatch(po.libc, chunk_reads[0][:n1])
                ropt1 = ref.opt(flag=abi.MEM_F_PE, n_threads=1)
                t0 = time.perf_counter()
                ref.lib.mem_process_seqs(ropt1, ref.bwt, ref.bns, ref.pac, 0, rb1.n, rb1.arr, None)
                t1 = time.perf_counter() - t0
                rb1.take_sam()
                out["cpu_baseline"]["one_thread"] = {"value": round(rb1.n / t1 / 1e6, 6), "unit": "Mreads/s", "cores": 1,
                                                     "sample": "the first %d reads of chunk 0, reference mem_process_seqs -t 1 (%.1f s)" % (rb1.n, t1)}
                out["parity_on_sample"] = bool(ok)
                if same is not None:
                    out["all_steps_identical_to_checked_sam"] = bool(same)
            else:
                out["cpu_baseline"] = None
        except Exception as e:  # the baseline must never take the bench line down
            out["cpu_baseline"] = {"error": repr(e)}
    # ---- N > 1 (or on request): every rank checks its own chunks against the reference; the line carries the AND over the ranks ----
    if args.check_parity:
        ok_here = 1
        try:
            from oracle import pyoracle as po
            ref = po.RefIndex(idx.prefix)
            C.c_int.in_dll(ref.lib, "bwa_verbose").value = 1
            ropt = ref.opt(flag=abi.MEM_F_PE, n_threads=cores)
            for c in range(n_chunks):
                sample = chunk_reads[c][:min(args.cpu_sample_pairs or len(chunk_reads[c]), len(chunk_reads[c]))]
                rb = abi.SeqBatch(po.libc, sample)
                ref.lib.mem_process_seqs(ropt, ref.bwt, ref.bns, ref.pac, 0, rb.n, rb.arr, None)
                if eng.process(opt, sample) != rb.take_sam():
                    ok_here = 0
        except Exception as e:
            log("rank %d: parity check failed to run: %r" % (rank, e))
            ok_here = 0
        if world > 1:
            t = torch.tensor([ok_here], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            ok_here = int(t.item())
        out["parity_all_ranks"] = bool(ok_here)
    out["index_residency"] = {"rank0": "host upload", "other_ranks": "broadcast from rank 0 (torch.distributed, in place on the index arrays)" if world > 1 else None,
                              "broadcast_s": round(eng.bcast_seconds, 3) if eng.bcast_seconds is not None else None}
    if alt_workloads is not None:
        out["alt_workloads"] = alt_workloads
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    # a throughput figure next to SAM that differs from the reference's is not a result: fail the run
    if out.get("parity_on_sample") is False or out.get("all_steps_identical_to_checked_sam") is False or out.get("parity_all_ranks") is False:
        log("FAILED: SAM differs from the reference (parity_on_sample=%r, all_steps_identical_to_checked_sam=%r)" %
            (out.get("parity_on_sample"), out.get("all_steps_identical_to_checked_sam")))
        sys.exit(3)


if __name__ == "__main__":
    main()
