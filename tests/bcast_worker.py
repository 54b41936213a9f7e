"""Worker of tests/test_gpu_bcast.py (run under torch.distributed.run, 2 ranks on one GPU, gloo)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from mpibwa_amd import api, dist as D  # noqa: E402
from golden_util import golden_index, load_reads, load_sam, sam_cases  # noqa: E402

rank, world, local = D.env_world()
backend = os.environ.get("BCAST_BACKEND", "gloo")
dev = local % torch.cuda.device_count() if backend == "nccl" else 0
torch.cuda.set_device(dev)
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
d = os.environ["BCAST_DIR"]
if rank == 0:
    golden_index(d)
dist.barrier()
eng = api.Engine(os.path.join(d, "gold.fa"), device=dev, dist=dist, rank=rank)
assert eng.bcast_seconds is not None
assert len(eng.index_checksums) == 3 and any(eng.index_checksums)   # (compared with rank 0's inside; a difference raises)
print("CHECKSUMS rank=%d %s" % (rank, " ".join("%016x" % x for x in eng.index_checksums)), flush=True)
kw = sam_cases()["pe_default"]
out = b"".join(eng.process(eng.opt(**kw), load_reads("reads_pe150.tsv.gz")))
ok = out == load_sam("pe_default")
print("RESULT rank=%d ok=%d" % (rank, int(ok)), flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 3)
