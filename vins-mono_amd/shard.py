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
    # checks on the FULL problem, so that every rank refuses alike before the first collective (host/sharded_estimator.cpp validate_full)
    if len(off) > 1 and not np.all(off[1:] == off[:-1] + nob[:-1]):
        raise ValueError("observation rows of consecutive landmarks must be consecutive")
    if world > 1 and prob.get('relo') is not None and len(prob['relo']['match']) > 0:
        # the relocalisation pose would be a block of the reduced system only on the ranks that hold one of its matched landmarks:
        # reduced systems of different sizes would be summed
        raise ValueError("relocalisation factors are not offered in a window sharded over several ranks")
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


# ---- marginalization of a sharded window --------------------------------------------------------------------------------------
def frame0_share(sub, inv_depth):
    """What a MARGIN_OLD marginalization needs from this rank's landmark share: the tracks anchored at frame 0 with their
    observation rows and their (solved) inverse depths.  The reference hands MarginalizationInfo only the projection factors of
    features with start_frame == 0 (estimator.cpp:853-888) next to the IMU factor of the first interval and the old prior; tracks
    anchored later do not touch the dropped states."""
    start = np.asarray(sub['lm_start'])
    nobs = np.asarray(sub['lm_nobs'])
    off = np.asarray(sub['obs_off'])
    obs = np.asarray(sub['obs'], dtype=np.float64).reshape(-1, 7)
    sel = [l for l in range(len(start)) if start[l] == 0]
    rows = [obs[off[l]:off[l] + nobs[l]] for l in sel]
    return dict(lm_nobs=[int(nobs[l]) for l in sel], inv_depth=[float(np.asarray(inv_depth)[l]) for l in sel],
                obs=(np.concatenate(rows) if rows else np.zeros((0, 7))).tolist())


def marginalize_sharded(handle, sub, state, flag, gather):
    """Marginalization of a window whose landmarks are spread over ranks (Estimator::optimization's second half,
    estimator.cpp:825-1000), after the sharded solve.  MARGIN_OLD touches frame 0 only: the ranks all-gather their frame-0
    shares (`gather(obj)` -> the objects of all ranks in rank order, e.g. torch.distributed.all_gather_object; a few KB), and
    every rank marginalizes the SAME small single-rank problem -- all frames at their solved states, the IMU factors, the old
    prior, the frame-0 tracks of all ranks in landmark order, max_iters = 0 (the factors are evaluated where the solve ended;
    with nothing to iterate the gauge fix is the identity) -- so every rank holds the identical new prior without a broadcast.
    MARGIN_SECOND_NEW involves the prior alone.  `handle`: a vg_handle WITHOUT an all-reduce hook (a second handle on the rank's
    device: the one that runs the sharded solve keeps its hook / RCCL communicator).  Returns the new prior (None if the old one
    stays, estimator.cpp:935-936)."""
    from . import ba
    pieces = gather(frame0_share(sub, state['inv_depth'])) if flag == ba.VG_MARGIN_OLD else []
    red = dict(sub)
    red.pop('shard', None)
    red.update(pose=np.asarray(state['pose'], float), sb=np.asarray(state['sb'], float), ex=np.asarray(state['ex'], float), td=float(state['td']),
               max_iters=0, relo=None)
    nobs = [n for p in pieces for n in p['lm_nobs']]
    red['lm_nobs'] = np.array(nobs, np.int32)
    red['lm_start'] = np.zeros(len(nobs), np.int32)
    red['obs_off'] = np.concatenate([[0], np.cumsum(nobs)[:-1]]).astype(np.int32) if nobs else np.zeros(0, np.int32)
    red['inv_depth'] = np.array([d for p in pieces for d in p['inv_depth']], float)
    red['obs'] = np.array([r for p in pieces for r in p['obs']], float).reshape(-1, 7)
    _, sm, prior = handle.ba_optimize(red, flag)
    if sm['status'] != 0:
        raise RuntimeError(f"marginalization of the sharded window failed with status {sm['status']}")
    return prior
