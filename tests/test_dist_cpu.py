"""World-size-2 check of the multi-GPU plumbing on CPU (gloo): shards partition the work, every rank sees the same
aggregate, and the timing reduction is a MAX — the same code path bench.py takes under torch.distributed.run with nccl."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    from mpibwa_amd import dist as D, simulate
    import numpy as np
    d = D.init("gloo")
    rank, world, _ = D.env_world()
    chunks = list(range(37))
    lo, hi = D.shard_slice(len(chunks), rank, world)
    mine = chunks[lo:hi]
    # per-rank read streams are disjoint and reproducible
    _, seqs = simulate.make_genome(20000, 1, seed=3, n_runs=0)
    reads = simulate.simulate_reads(seqs, 20, 50, paired=True, seed=D.shard_seed(100, rank))
    digest = int(sum(int(r[1].sum()) + int(r[2].sum()) for r in reads))
    d.barrier()
    elapsed = 0.25 * (rank + 1)
    out = {"rank": rank, "world": world, "lo": lo, "hi": hi, "n_sum": D.sum_over_ranks(d, len(mine)),
           "t_max": D.max_over_ranks(d, elapsed), "digest": digest}
    sys.stdout.write(chr(10) + "RESULT " + json.dumps(out) + chr(10))   # one write: the ranks share the pipe
    sys.stdout.flush()
    d.destroy_process_group()
""") % ROOT


def test_two_rank_gloo_sharding(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    import re
    res = sorted((json.loads(m) for m in re.findall(r"RESULT (\{.*?\})", r.stdout)), key=lambda x: x["rank"])
    assert [x["rank"] for x in res] == [0, 1] and all(x["world"] == 2 for x in res)
    assert res[0]["lo"] == 0 and res[0]["hi"] == res[1]["lo"] and res[1]["hi"] == 37      # shards partition the chunk list
    assert all(x["n_sum"] == 37 for x in res)                                             # every rank sees the whole-job total
    assert all(abs(x["t_max"] - 0.5) < 1e-9 for x in res)                                  # timing = MAX over ranks
    assert res[0]["digest"] != res[1]["digest"]                                           # per-rank read streams differ


WORKER2 = textwrap.dedent("""
    import ctypes as C, hashlib, json, os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    from mpibwa_amd import api, dist as D, fastq
    d = D.init("gloo")
    rank, world, _ = D.env_world()
    lib = api.load_library()
    src = fastq.FastqSource(lib, os.environ["FQ1"], os.environ["FQ2"], K=40_000)      # the reference's chunk rule on real files
    lo, hi = D.shard_slice(src.n_chunks, rank, world)
    reads, h = 0, hashlib.md5()
    for c in range(lo, hi):                                                            # this rank's chunks -> bseq1_t, as the hot path gets them
        rec, n = src.chunk(c)[:2]
        reads += n
        for i in range(n):
            h.update(C.string_at(int(rec["name"][i])) + b"|" + C.string_at(int(rec["seq"][i])) + b"|" + C.string_at(int(rec["qual"][i])) + b"\\n")
    out = {"rank": rank, "n_chunks": src.n_chunks, "lo": lo, "hi": hi, "reads": reads, "reads_all": D.sum_over_ranks(d, reads), "md5": h.hexdigest(),
           "starts": [int(x) for x in src.starts]}
    sys.stdout.write(chr(10) + "RESULT " + json.dumps(out) + chr(10))   # one write: the ranks share the pipe
    sys.stdout.flush()
    d.destroy_process_group()
""") % ROOT


def test_two_ranks_shard_the_chunks_of_real_fastq_files(tmp_path, built):
    """Two ranks compute the same chunk table from the same FASTQ pair (mpiBWA's rule, mi355x_fastq_chunks), take disjoint shards
    of it and build their bseq1_t arrays: together they hold every read exactly once, in the order of a single process."""
    import hashlib
    import numpy as np
    sys.path.insert(0, ROOT)
    from mpibwa_amd import simulate
    _, seqs = simulate.make_genome(30000, 1, seed=3, n_runs=0)
    reads = simulate.reads_to_ascii(simulate.simulate_reads(seqs, 1500, 100, paired=True, seed=9, var_len=(60, 140)))
    fq = [str(tmp_path / "R1.fastq"), str(tmp_path / "R2.fastq")]
    with open(fq[0], "wb") as f1, open(fq[1], "wb") as f2:
        for n, a, b in reads:
            f1.write(b"@" + n.encode() + b"\n" + a + b"\n+\n" + b"I" * len(a) + b"\n")
            f2.write(b"@" + n.encode() + b"\n" + b + b"\n+\n" + b"J" * len(b) + b"\n")
    script = tmp_path / "worker2.py"
    script.write_text(WORKER2)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", FQ1=fq[0], FQ2=fq[1])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29733", str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    import re
    res = sorted((json.loads(m) for m in re.findall(r"RESULT (\{[^{}]*\})", r.stdout)), key=lambda x: x["rank"])
    assert len(res) == 2 and res[0]["starts"] == res[1]["starts"] and res[0]["n_chunks"] >= 4
    assert res[0]["lo"] == 0 and res[0]["hi"] == res[1]["lo"] and res[1]["hi"] == res[0]["n_chunks"]
    assert res[0]["reads"] + res[1]["reads"] == 2 * len(reads) == int(res[0]["reads_all"])
    # the two shards, one after the other, are the single-process stream
    whole = hashlib.md5()
    part = [hashlib.md5(), hashlib.md5()]
    starts = res[0]["starts"]
    for c in range(res[0]["n_chunks"]):
        which = 0 if c < res[0]["hi"] else 1
        for i in range(starts[c], starts[c + 1]):
            n, a, b = reads[i]
            for s_, q in ((a, b"I"), (b, b"J")):
                line = n.encode() + b"|" + s_ + b"|" + q * len(s_) + b"\n"
                whole.update(line); part[which].update(line)
    assert [res[0]["md5"], res[1]["md5"]] == [part[0].hexdigest(), part[1].hexdigest()]


def test_shard_slice_covers_everything():
    from mpibwa_amd import dist as D
    for n in (0, 1, 7, 64, 1000):
        for w in (1, 2, 3, 8):
            cuts = [D.shard_slice(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1
