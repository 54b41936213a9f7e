"""Index broadcast path (rank 0 uploads, the others receive occ blocks / SA / pac through torch.distributed.broadcast, in
place on the index arrays).  With two or more GPUs the two ranks take one each and the backend is nccl (= RCCL); on the
one-GPU test box both ranks share the device and gloo carries the CUDA tensors (RCCL refuses two ranks on one device; the
broadcast calls and everything around them are identical).  The receiving rank must reproduce the golden SAM.  The C-side
RCCL path (mi355x_init) is tested in test_gpu_boundary.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_index_broadcast_two_ranks(tmp_path, built):
    script = os.path.join(ROOT, "tests", "bcast_worker.py")
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", BCAST_DIR=str(tmp_path), MPIBWA_HOST_THREADS="4", BCAST_BACKEND=backend)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29741", str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    if r.returncode != 0:
        sys.stderr.write(r.stdout[-3000:] + "\n" + "\n".join(l for l in r.stderr.splitlines() if not l.startswith("[M::"))[-6000:])
    assert r.returncode == 0
    assert "rank=0 ok=1" in r.stdout and "rank=1 ok=1" in r.stdout
    # both ranks hashed their device copies and hold the same three numbers
    sums = sorted(l.split(None, 2) for l in r.stdout.splitlines() if l.startswith("CHECKSUMS"))
    assert len(sums) == 2 and sums[0][2] == sums[1][2], r.stdout[-1000:]
