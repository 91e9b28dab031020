"""Multi-GPU plumbing of bench.py (torch.distributed only: rendezvous, barrier, max/sum over ranks).

The hot paths shard by independent windows / camera streams ("replicas only"): every rank owns its
own batch, there is NO collective on the data path.  Collectives are used only to agree on the timing
(max over ranks) and to add up throughput.  backend "nccl" (= RCCL on ROCm) on GPUs, "gloo" in the CPU tests."""
import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def init(backend, local_rank=0):
    rank, _, world = env_rank()
    if world <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend)
    return True


def active():
    return dist.is_available() and dist.is_initialized()


def barrier():
    if active():
        dist.barrier()


def _device():
    return "cuda" if (active() and dist.get_backend() == "nccl") else "cpu"


def max_over_ranks(x):
    if not active():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(x):
    if not active():
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def window_seeds(rank, windows_per_gpu, base=1):
    """Disjoint synthetic-window seeds per rank (weak scaling: every rank gets `windows_per_gpu` of its own)."""
    return list(range(base + rank * windows_per_gpu, base + (rank + 1) * windows_per_gpu))


def finish():
    if active():
        dist.barrier()
        dist.destroy_process_group()
