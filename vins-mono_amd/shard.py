"""Landmark shards of one sliding window over ranks (BASELINE.json configs[4]).

Landmarks are conditionally independent given the frames, so every rank takes a CONTIGUOUS share of the landmarks
(balanced by sum (6 n_l)^2, the Schur-complement work of a track of n_l observations) together with their observations;
frames, IMU factors and the prior are replicated.  The ranks run the ordinary C-ABI calls on their share with an
all-reduce hook installed (include/vinsgpu.h, vg_ba_set_allreduce): two in-place sums per trust-region round, the
reduced camera system (~300 KB for 31 frames) and four doubles of step norms -- RCCL over xGMI on GPUs, gloo in the
CPU tests.  No other collective touches the data path."""
import ctypes as C

import numpy as np


def landmark_shards(lm_nobs, world):
    """[(lo, hi)] * world: contiguous landmark ranges with about equal sum (6 n_l)^2."""
    n = np.asarray(lm_nobs, dtype=np.float64)
    w = (6.0 * n) ** 2
    cum = np.concatenate([[0.0], np.cumsum(w)])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(cum, total * r / world, side="left")))
    cuts.append(len(n))
    cuts = [min(max(c, 0), len(n)) for c in cuts]
    for r in range(1, len(cuts)):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def shard_problem(prob, rank, world):
    """The sub-problem of `rank`: same frames / IMU / prior / flags, landmarks [lo, hi) with re-based observation offsets."""
    lo, hi = landmark_shards(prob['lm_nobs'], world)[rank]
    sub = dict(prob)
    obs = np.asarray(prob['obs'], dtype=np.float64).reshape(-1, 7)
    off = np.asarray(prob['obs_off'], dtype=np.int64)
    nob = np.asarray(prob['lm_nobs'], dtype=np.int64)
    o_lo = int(off[lo]) if lo < hi else 0
    o_hi = int(off[hi - 1] + nob[hi - 1]) if lo < hi else 0
    if lo < hi and not np.all(off[lo + 1:hi] == off[lo:hi - 1] + nob[lo:hi - 1]):
        raise ValueError("observation rows of consecutive landmarks must be consecutive")
    sub['inv_depth'] = np.asarray(prob['inv_depth'], dtype=np.float64)[lo:hi].copy()
    sub['lm_start'] = np.asarray(prob['lm_start'])[lo:hi].copy()
    sub['lm_nobs'] = np.asarray(prob['lm_nobs'])[lo:hi].copy()
    sub['obs_off'] = (off[lo:hi] - o_lo).astype(np.int32)
    sub['obs'] = obs[o_lo:o_hi].copy()
    relo = prob.get('relo')
    if relo is not None:
        sub['relo'] = dict(relo, match=[(m[0] - lo, m[1], m[2]) for m in relo['match'] if lo <= m[0] < hi])
    sub['shard'] = (lo, hi)
    return sub


def torch_allreduce_hook(device_buffers=None):
    """An all-reduce hook for Handle.ba_set_allreduce on top of torch.distributed.
    backend nccl (= RCCL): sums the device buffer in place on the library's HIP stream (zero-copy view of the pointer).
    backend gloo with host buffers (CPU tests against the emulated library): sums the host buffer.
    backend gloo with DEVICE buffers (device_buffers=True: two ranks sharing one GPU in a test, where RCCL refuses duplicate
    devices): staged through the host on the library's stream."""
    import torch
    import torch.distributed as dist

    nccl = dist.get_backend() == "nccl"
    on_gpu = nccl if device_buffers is None else bool(device_buffers)

    class _DevBuf:
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}

    def hook(ptr, count, stream):
        if on_gpu:
            t = torch.as_tensor(_DevBuf(ptr, count), device="cuda")
            with torch.cuda.stream(torch.cuda.ExternalStream(stream)):
                if nccl:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)
                else:
                    host = t.cpu()                       # synchronises the library's stream up to here
                    dist.all_reduce(host, op=dist.ReduceOp.SUM)
                    t.copy_(host)
        else:
            a = np.ctypeslib.as_array((C.c_double * count).from_address(ptr))
            t = torch.from_numpy(a)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return hook
