"""Multi-GPU plumbing shared by bench.py and the tests: one process per GPU, reads sharded by rank, no data-path
collective (chunks are independent units); the only collectives are a barrier and a MAX over the ranks'
elapsed times.  Works with backend "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU."""
import os


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend, device_id=None):
    import torch.distributed as dist
    rank, world, _ = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device_id} if device_id is not None else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist if world > 1 else None


def shard_seed(base_seed, rank):
    """Every rank simulates its own, disjoint stream of reads (weak scaling: fixed work per GPU)."""
    return base_seed + rank


def shard_slice(n_items, rank, world):
    """Contiguous shard of a shared list of chunks (strong scaling / real inputs): [lo, hi)."""
    per, rem = divmod(n_items, world)
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def max_over_ranks(dist, value, device="cpu"):
    import torch
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, value, device="cpu"):
    import torch
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
