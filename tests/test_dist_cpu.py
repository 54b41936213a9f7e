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
    print("RESULT " + json.dumps(out), flush=True)
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


def test_shard_slice_covers_everything():
    from mpibwa_amd import dist as D
    for n in (0, 1, 7, 64, 1000):
        for w in (1, 2, 3, 8):
            cuts = [D.shard_slice(n, r, w) for r in range(w)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1
