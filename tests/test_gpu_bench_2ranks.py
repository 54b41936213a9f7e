"""bench.py itself under torch.distributed.run with two ranks (the driver's N > 1 launch line) on the one-GPU box: both ranks
share the device and gloo carries the device tensors of the index broadcast (RCCL refuses two ranks on one GPU; with two or
more GPUs the backend is nccl = RCCL and each rank takes its own).  The line must say n_gpus = 2, every rank's chunks must
match the reference, and rank 1 must hold an index it received by broadcast."""
import json
import os
import subprocess
import sys

import pytest

from oracle import pyoracle as po

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not po.ref_available(), reason="oracle/_ref/libbwaref.so not present")]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_shard_reads_and_broadcast_the_index(tmp_path, built):
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MPIBWA_BENCH_BACKEND=backend, MPIBWA_BENCH_DIR=str(tmp_path), MPIBWA_HOST_THREADS="6")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", "29747",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--genome-mbp", "50", "--pairs", "20000", "--chunks", "2",
           "--in-flight", "2", "--check-parity"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900, cwd=ROOT)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-3000:] + "\n" + "\n".join(l for l in r.stderr.splitlines() if not l.startswith("[M::"))[-6000:])
    assert r.returncode == 0
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 2
    assert d["parity_all_ranks"] is True
    assert d["index_residency"]["other_ranks"].startswith("broadcast") and d["index_residency"]["broadcast_s"] is not None
    assert d["value"] > 0 and "roofline" in d and d.get("cpu_baseline") is None or "cpu_baseline" not in d
