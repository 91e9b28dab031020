#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native VINS-Mono hot paths.

Metric (BASELINE.json): sliding-window BA solves/sec (+ KLT features/sec as a second object) on the
EuRoC-shaped window (K = 11 frames, ~150 landmarks, IMU + projection + marginalization-prior factors,
8 trust-region iterations, followed by the MARGIN_OLD marginalization: one full Estimator::optimization()).

A "step" = one pass of the hot path over one batch of WINDOWS_PER_GPU independent synthetic windows that are
already resident in HBM (BASELINE.json configs[3]: "batch of 256 independent EuRoC-shape windows").
Multi-GPU is independent-batch (weak scaling, no collective on the data path): every rank owns its own batch.

    python bench.py --gpus N --steps K --warmup W        (N > 1 under torch.distributed.run)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

# ROCm maps HIP streams onto 4 hardware queues by default; the boundary loops below drive 8 handles (16 streams) from 8 host
# threads, and a stream whose next packet waits for a copy holds up the other streams of its queue (measured: 3.87 vs
# 3.25 ms per batch).  Must be set before the HIP runtime initialises; no effect on `value`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

QUICK = False                # --emulated: the contract self-test on the CPU (tests/test_bench_contract.py): tiny loop counts, timings meaningless
WINDOWS_PER_GPU = 256
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = FP64 matrix peak (AMD datasheet)
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md


def make_windows(h, ba, synth, n, seed0):
    """n timed windows: for every seed, window 1 (no prior) is solved + marginalised by the PRODUCT path to
    create the prior, then window 2 (with prior) is assembled — the timed one (SURVEY.md 8(d) configs[2])."""
    seqs = [synth.SyntheticSequence(seed0 + s, n_frames=12 + CHAIN_FRAMES) for s in range(n)]
    first = [q.window(0) for q in seqs]
    h.ba_upload(first, [ba.VG_MARGIN_OLD] * n)
    h.ba_run_async()
    st, sm, pr = h.ba_download()
    assert all(s['status'] == 0 for s in sm), "window-1 solve failed"
    return [q.next_window(st[i], pr[i], 1) for i, q in enumerate(seqs)], seqs


CHAIN_FRAMES = 4             # consecutive frames every sequence advances in the chained boundary loop


def make_chain(h, ba, seqs, packed, flags):
    """Frames 2 .. 1+CHAIN_FRAMES of every sequence: frame 2 is the timed window (prior from the host), each following one is
    built by the sliding-window bookkeeping (SyntheticSequence.next_window = slideWindow + the new frame) from the product's
    own result of the frame before, and its prior is the one that solve's marginalization left ON THE DEVICE."""
    chain = [ba.PackedBatch(packed)]
    for k in range(1, CHAIN_FRAMES):
        h.ba_upload(chain[-1], flags)
        h.ba_run_async()
        st, sm, _ = h.ba_download()
        assert all(s['status'] == 0 for s in sm), "chain solve failed"
        chain.append(ba.PackedBatch([q.next_window(st[i], 'resident', k + 1) for i, q in enumerate(seqs)]))
    return chain


def bench_resident(ba, synth, seqs, nthr=4, rounds=3):
    """Windows that stay on the device (vg_ba_seq_*): every thread walks its handle through the same CHAIN_FRAMES consecutive
    frames of the sequences as the chained loop, but the window never crosses the boundary -- per frame the host sends the new
    frame's observations, state guess and pre-integration (vg_ba_seq_step_async) and reads the states back.  Returns ms per
    256-window frame (all threads together) and the per-step record of the last round."""
    import threading
    K = 11
    src = [synth.FrameSource(q, noise_seed=1000 + i) for i, q in enumerate(seqs)]
    wins, trks = zip(*[synth.sequence_inputs(s.initial_window(K, 0)) for s in src])
    steps = []
    for k in range(1 + CHAIN_FRAMES):                    # global frames K-1 .. K-1+CHAIN_FRAMES
        g = K - 1 + k
        frames = []
        for s, q in zip(src, seqs):
            ids, rows = s.image(g)
            pose, sb = s.guess(g)
            frames.append(dict(pose=pose, sb=sb, imu_new=q.imu[g - 1], imu_merged=None, ids=ids, obs=rows))
        steps.append(ba.Handle.seq_pack_frames(frames))
    hs = [ba.Handle() for _ in range(nthr)]
    errs, info = [], None

    def begin(hh):
        # min_parallax = 0: every frame is a key frame (MARGIN_OLD), like the flags of the chained loop
        hh.seq_begin(list(wins), list(trks), max_features=384, max_new_obs=384, max_landmarks=256, max_factors=1536, min_parallax=0.0)
        hh.seq_step(steps[0])                             # first frame: no prior yet; creates it
        hh.ba_prepare_download()
        hh.sync()

    def worker(hh):
        try:
            for st in steps[1:]:
                hh.seq_step(st)
                rc = hh.ba_download_state_raw()
                if rc != 0:
                    errs.append(rc)
        except Exception as ex:                           # noqa: BLE001
            errs.append(repr(ex))

    total = 0.0
    for r in range(rounds + 1):
        for hh in hs:
            begin(hh)
        ts = [threading.Thread(target=worker, args=(hh,)) for hh in hs]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if r > 0:                                         # round 0 warms up (allocations, graph capture)
            total += time.perf_counter() - t0
    info = hs[0].seq_info()
    for hh in hs:
        hh.seq_end(); hh.close()
    if errs:
        raise RuntimeError(f"resident-sequence loop failed: {errs[:3]}")
    bad = [i for i in info if i['status'] != 0]
    if bad:
        raise RuntimeError(f"resident-sequence loop: capacity exceeded in {len(bad)} windows")
    ms = total / (rounds * CHAIN_FRAMES * nthr) * 1e3
    return ms, {"host_threads": nthr, "frames": CHAIN_FRAMES, "landmarks_mean": float(np.mean([i['n_landmarks'] for i in info])),
                "factors_mean": float(np.mean([i['n_factors'] for i in info])), "tracks_mean": float(np.mean([i['n_features'] for i in info]))}


FE_CAMS = 256                # independent camera streams per GPU in the front-end leg (like the 256 windows of the BA leg; with 64 the LK launch is
                             # dominated by its slowest tracks: 137 us for 9600 tracks against 9.7 ns per additional track, tests/manual/gpu_lk_scaling.py)
FE_BYTES_PER_FEATURE = 8188  # (pyramid 592,200 B + LK 636,000 B) per 752x480 frame / 150 features
FE_GFTT_BYTES_PER_FRAME = 752 * 480 * (70 * 22) // (64 * 16) + 752 * 480 + 8 * 4096   # what the detection really moves: the frame once with the tile halo (70 x 22 per 64 x 16 tile), the mask, ~4K candidate keys (the map stays in LDS; SURVEY 8(d)'s 3.6 MB counted it written and read back)
FE_DISTINCT = 256            # distinct image pairs of the front-end leg (one per stream, seed 1 + b)


def bench_fe(h, synth, steps, warmup, rank, with_cpu):
    """Front-end leg (BASELINE.json configs[1]): per step and per stream, pyramid build of the new 752x480 frame +
    4-level pyramidal LK of 150 corners (frames resident in HBM, two alternating frames per stream)."""
    import numpy as np
    from vins_mono_amd import fe
    W, H, N = 752, 480, 150
    tr = fe.FrontEnd(h, W, H, FE_CAMS, N)
    # every stream its own image pair, seed 1 + b for batch item b (ranks take disjoint items) -- the cost of LK
    # depends on the content (iterations per level), so 256 replicas of 8 pairs are not the stated workload.  (FE_DISTINCT < FE_CAMS
    # only for profiler passes, --quick-fe: the generator takes ~0.1 s per pair.)
    ndist = min(FE_CAMS, FE_DISTINCT)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        base = list(ex.map(lambda c: synth.synth_frame(1 + rank * FE_CAMS + c), range(ndist)))
        nxt = list(ex.map(lambda c: synth.warp_frame(base[c], 100001 + rank * FE_CAMS + c), range(ndist)))
    fa = [base[c % ndist] for c in range(FE_CAMS)]
    fb = [nxt[c % ndist] for c in range(FE_CAMS)]
    tr.push_frames(fa)                      # frame A (and its pyramid)
    tr.detect_upload([N] * FE_CAMS)
    tr.detect_async()
    corners = tr.detect_download()
    tr.upload_frames(fb)                    # frame B into the other frame slot
    tr.track_upload(corners)
    slot = tr.frame_slot()                  # the steps alternate between the two resident frames, starting with B

    def step():
        nonlocal slot
        tr.select_frames(slot)
        tr.build_async(False)
        tr.track_async()
        slot ^= 1

    for _ in range(warmup):
        step()
    h.sync()
    t0 = time.perf_counter()
    h.timer_start()
    for _ in range(steps):
        step()
    ev_ms = h.timer_stop()
    wall = time.perf_counter() - t0
    nfeat = sum(len(c) for c in corners)
    res = tr.track_download()
    tracked = int(sum(int(st.sum()) for (_, st, _) in res))
    last_is_a_to_b = (warmup + steps) % 2 == 1          # the steps alternate A -> B, B -> A with the same start points
    # the same step with the frames arriving from HOST memory every step (the boundary-inclusive figure; never `value`):
    # 256 x 361 KB = 92 MB of H2D per step in front of the pyramid build
    up_frames = [fb, fa]
    tr.upload_frames(fb)
    h.sync()
    t_up = time.perf_counter()
    for k in range(max(4, steps // 2)):
        tr.upload_frames(up_frames[k % 2])
        tr.build_async(False)
        tr.track_async()
    h.sync()
    up_ms = (time.perf_counter() - t_up) / max(4, steps // 2) * 1e3
    # ... and from ONE page-locked buffer holding all streams' frames (vg_host_register; contiguous frames travel as one copy)
    ring = [np.ascontiguousarray(np.stack(fb)), np.ascontiguousarray(np.stack(fa))]
    pinned_ms = None
    try:
        for r in ring:
            h.host_register(r)
        tr.upload_frames(list(ring[0]))
        h.sync()
        t_up = time.perf_counter()
        for k in range(max(4, steps // 2)):
            tr.upload_frames(list(ring[k % 2]))
            tr.build_async(False)
            tr.track_async()
        h.sync()
        pinned_ms = (time.perf_counter() - t_up) / max(4, steps // 2) * 1e3
        for r in ring:
            h.host_unregister(r)
    except RuntimeError:
        pinned_ms = None
    tr.upload_frames(fb if (warmup + steps) % 2 == 0 else fa)      # leave the slots as the alternating steps below expect them
    slot = tr.frame_slot()
    # the same step with CLAHE(3.0, 8x8) on the incoming frame (EQUALIZE = 1 in the EuRoC configuration,
    # feature_tracker.cpp:87-93)
    def step_eq():
        nonlocal slot
        tr.select_frames(slot)
        tr.build_async(True)
        tr.track_async()
        slot ^= 1
    for _ in range(2):
        step_eq()
    h.sync()
    h.timer_start()
    for _ in range(steps):
        step_eq()
    eq_ms = h.timer_stop() / steps
    # GFTT alone
    h.timer_start()
    for _ in range(max(3, steps // 2)):
        tr.detect_async()
    gftt_ms = h.timer_stop() / max(3, steps // 2)
    out = {
        "metric": "KLT features/sec (pyramid build + 4-level 21x21 pyramidal LK, 150 corners per 752x480 frame)",
        "value": nfeat * steps / wall, "unit": "features/s", "streams": FE_CAMS, "features_per_step": nfeat,
        "tracked_last_step": tracked, "ms_per_step": wall / steps * 1e3, "dtype": "u8/int16/int64 + f32",
        "gftt_frames_per_s": FE_CAMS / (gftt_ms * 1e-3), "gftt_ms_per_batch": gftt_ms,
        "with_clahe": {"ms_per_step": eq_ms, "features_per_s": nfeat / (eq_ms * 1e-3),
                       "what": "same step with CLAHE(3.0, 8x8) on the incoming frame (EuRoC equalize: 1), HIP events"},
        "upload_inclusive": {"ms_per_step": up_ms, "features_per_s": nfeat / (up_ms * 1e-3), "h2d_bytes_per_step": FE_CAMS * W * H,
                             "from_registered_memory": None if pinned_ms is None else {
                                 "ms_per_step": pinned_ms, "features_per_s": nfeat / (pinned_ms * 1e-3),
                                 "what": "all frames in one page-locked buffer (vg_host_register): one hipMemcpyAsync per step"},
                             "what": "the same step with every stream's frame uploaded from pageable host memory first (vg_fe_upload_frames: one "
                                     "hipMemcpy2DAsync per stream, then a stream synchronisation), wall clock; NOT the metric"},
        "roofline": {"kernel": "fe_lk_kernel (+ fe_pyrdown_kernel x3)", "bound": "hbm",
                     "achieved": nfeat * FE_BYTES_PER_FEATURE * steps / (ev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "traffic": fe_traffic(), "event_ms_per_step": ev_ms / steps,
                     # LK is VALU-bound, not HBM-bound: the HBM fraction alone says nothing about headroom, the issue fraction does
                     "valu_issue_frac": pmc_field("fe_lk_kernel", "valu_issue_frac"),
                     "waves_parked_frac": pmc_field("fe_lk_kernel", "waves_parked_frac"),
                     "distinct_image_pairs": ndist,
                     "gftt": {"kernel": "fe_mineig_kernel (+ fe_select_kernel)", "bytes_per_frame": FE_GFTT_BYTES_PER_FRAME,
                              "achieved_GBs": FE_CAMS * FE_GFTT_BYTES_PER_FRAME / (gftt_ms * 1e-3) / 1e9,
                              "frac": FE_CAMS * FE_GFTT_BYTES_PER_FRAME / (gftt_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "valu_issue_frac": pmc_field("fe_mineig_kernel", "valu_issue_frac"),
                              "note": "bytes the detection really moves: the frame read once with its tile halo, the mask, the candidate "
                                      "keys (the min-eigenvalue map stays in LDS since round 4); VALU-bound, see valu_issue_frac"}},
    }
    # Which roof LK actually hits: the SQ counters say VALU issue -- valu_issue_frac is the share of a wavefront's
    # cycles in which it issues a VALU instruction, a SIMD holds `waves_per_simd` of them and issues one VALU instruction per cycle, so
    # their product is the occupancy of the SIMD's VALU issue slot (1.0 = saturated).  The HBM figure of SURVEY 8(d) stays as a side value.
    rf = out["roofline"]
    hbm_side = {"achieved": rf["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rf["achieved"] / HBM_PEAK_GBS, "traffic": rf["traffic"],
                "algorithmic_bytes_per_feature": FE_BYTES_PER_FEATURE}
    res_lk = kernel_resources("fe_lk_kernel")
    if rf["valu_issue_frac"] is not None and res_lk:
        slot = rf["valu_issue_frac"] * res_lk["waves_per_simd"]
        rf.update({"bound": "valu", "achieved": slot, "peak": 1.0, "unit": "VALU issue slot occupancy per SIMD (valu_issue_frac x waves_per_simd)",
                   "frac": slot, "waves_per_simd": res_lk["waves_per_simd"], "vgprs": res_lk["vgprs"], "hbm": hbm_side})
    else:
        rf.update({"frac": hbm_side["frac"], "hbm": hbm_side, "waves_per_simd": res_lk["waves_per_simd"] if res_lk else None,
                   "bound_note": "no SQ counters recorded for the device code of this library (profiles/pmc_latest.json is for another build): "
                                 "booked against HBM as SURVEY 8(d) says; LK is VALU-issue bound (DESIGN 2)"})
    res_me = kernel_resources("fe_mineig_kernel")
    if res_me:
        rf["gftt"]["waves_per_simd"] = res_me["waves_per_simd"]
        if rf["gftt"]["valu_issue_frac"] is not None:
            rf["gftt"]["valu_slot_occupancy"] = rf["gftt"]["valu_issue_frac"] * res_me["waves_per_simd"]
    if with_cpu:
        from oracle import fe_cpu
        # parity of the TIMED configuration: the last timed step of EVERY stream against the oracle (38,400 tracks: 1.7 s of one core)
        par_streams = list(range(FE_CAMS))
        npar, nbad, ncmp = len(par_streams), 0, 0
        for c in par_streams:
            pa, pb = (fa[c], fb[c]) if last_is_a_to_b else (fb[c], fa[c])
            r_nxt, r_st, r_err = fe_cpu.lk(pa, pb, corners[c])
            g_nxt, g_st, g_err = res[c]
            ok = (np.array_equal(r_st, g_st) and np.array_equal(r_nxt.view(np.uint32), g_nxt.view(np.uint32))
                  and np.array_equal(r_err.view(np.uint32), g_err.view(np.uint32)))
            nbad += 0 if ok else 1
            ncmp += len(r_st)
        out["parity"] = {"streams_checked": npar, "of": FE_CAMS, "tracks_compared": ncmp, "streams_bit_identical": npar - nbad,
                         "what": "status, positions and err (float bit patterns) of the last timed step vs oracle/fe_cpu.cpp (PARITY UNPINNED "
                                 "against OpenCV itself: tests/golden/make_golden_opencv.py is the kit)"}
        if nbad:
            raise RuntimeError(f"front-end parity failed on {nbad} of {npar} timed streams")
        t = time.perf_counter()
        reps = 0
        while time.perf_counter() - t < (0.2 if QUICK else 5.0):
            fe_cpu.lk(fa[reps % len(fa)], fb[reps % len(fb)], corners[reps % len(corners)])
            reps += 1
        dt = time.perf_counter() - t
        t2 = time.perf_counter()
        g = 0
        while time.perf_counter() - t2 < (0.2 if QUICK else 3.0):
            fe_cpu.gftt(fa[g % len(fa)], N)
            g += 1
        out["cpu_baseline"] = {"value": reps * N / dt, "unit": "features/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} frame pairs x {N} corners, oracle/fe_cpu.cpp (restated single-thread OpenCV-equivalent "
                                         f"calcOpticalFlowPyrLK incl. both pyramids + Scharr; real OpenCV unavailable)",
                               "gftt_frames_per_s": g / (time.perf_counter() - t2)}
        # OpenCV runs the LK point loop under parallel_for_: points of a level spread over threads
        # (8 threads: 150 points per frame do not feed more -- beyond that the per-level fork / join costs more than it buys)
        nthr = max(1, min(os.cpu_count() or 1, 8))
        t3 = time.perf_counter()
        rm = 0
        while time.perf_counter() - t3 < (0.2 if QUICK else 3.0):
            fe_cpu.lk_mt(fa[rm % len(fa)], fb[rm % len(fb)], corners[rm % len(corners)], nthr)
            rm += 1
        out["cpu_baseline"]["all_cores"] = {"value": rm * N / (time.perf_counter() - t3), "unit": "features/s", "cores": nthr,
                                            "what": "same restatement, points of every pyramid level spread over host threads "
                                                    "(one frame pair at a time, as one camera stream would run it)"}
        out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def bench_sharded(args, ba, synth, D, rank, world):
    """BASELINE configs[4]: one enlarged window (K = 31 frames = WINDOW_SIZE 30, 2000 landmarks, ~16K projection factors,
    30 IMU factors, prior on the oldest frame), landmarks sharded over the ranks (vins_mono_amd/shard.py), frames / IMU /
    prior replicated.  A step = one full trust-region solve of the window (8 iterations); per iteration two all-reduces
    (RCCL over xGMI): the reduced camera system [Sp | gp | T] and four doubles of step norms.  STRONG scaling: the window
    is the same for every N."""
    import torch
    from vins_mono_amd import shard
    K, L = (16, 60) if QUICK else (31, 2000)
    cuda_sync = (lambda: None) if QUICK else torch.cuda.synchronize
    seq = synth.SyntheticSequence(5, n_frames=K + 1, K=K, L=L)
    prob = synth.SyntheticSequence.anchor_prior(seq.window(0))        # same seed -> the same window on every rank
    sub = shard.shard_problem(prob, rank, world)
    h = ba.Handle()
    hook_kind = "none (single rank)"
    if world > 1 or args.rccl_hook:
        import torch.distributed as dist
        if (dist.is_initialized() and dist.get_backend() == "nccl" and not args.share_device) or (world == 1 and args.rccl_hook):
            # the C hook inside the library (csrc/vg_rccl.hip): ncclAllReduce on the launch stream, no Python in the loop; the
            # unique id travels over torch.distributed's own rendezvous
            ids = [h.rccl_unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(ids, src=0)
            h.ba_rccl_init(world, rank, ids[0])
            hook_kind = "RCCL in the library (vg_ba_rccl_init)"
        else:
            # (gloo self-test: two ranks on a shared GPU stage through the host; under --emulated the "device" buffers ARE host memory)
            h.ba_set_allreduce(shard.torch_allreduce_hook(device_buffers=not args.emulated))
            hook_kind = "torch.distributed gloo, staged through the host (self-test)"
    packed = ba.PackedProblem(sub)
    h.ba_upload([packed], [ba.VG_MARGIN_NONE])
    info = h.ba_info()
    counts = h.ba_reduce_layout()

    def barrier():
        cuda_sync()
        D.barrier()
    for _ in range(args.warmup):
        h.ba_run_async()
    h.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h.ba_run_async()
    h.sync()
    cuda_sync()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = D.max_over_ranks(elapsed)
    # per-launch HIP events (outside the timed region; every rank runs the pass: the collectives need all of them)
    prof = {}
    runs = 5
    for _ in range(runs):
        for k, (ms, n) in h.ba_run_profiled().items():
            a = prof.setdefault(k, [0.0, 0])
            a[0] += ms; a[1] += n
    st, sm, _ = h.ba_download()
    ok = int(sm[0]['status'] == 0)
    # marginalization of the sharded window (shard.marginalize_sharded: all-gather of the frame-0 tracks, then the same small
    # single-rank problem on every rank, on a handle of its own) -- informational, outside the timed region
    def gather(obj):
        if world == 1:
            return [obj]
        import torch.distributed as dist
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
    marg_info = None
    h2 = ba.Handle()
    tm0 = time.perf_counter()
    new_prior = shard.marginalize_sharded(h2, sub, st[0], ba.VG_MARGIN_OLD, gather)
    marg_first_ms = (time.perf_counter() - tm0) * 1e3
    tm0 = time.perf_counter()
    new_prior = shard.marginalize_sharded(h2, sub, st[0], ba.VG_MARGIN_OLD, gather)
    marg_info = {"ms": (time.perf_counter() - tm0) * 1e3, "first_call_ms": marg_first_ms, "kept_dimension": int(new_prior['n']) if new_prior else 0,
                 "dropped_dimension": int(new_prior['m']) if new_prior else 0,
                 "what": "MARGIN_OLD of the sharded window: all-gather of the tracks anchored at frame 0 (a few KB), then every rank "
                         "marginalizes the same reduced single-rank problem (all frames at the solved states, IMU, old prior, the "
                         "frame-0 tracks; max_iters = 0) on a second handle: host packing + H2D + kernels + D2H, wall clock"}
    h2.close()
    nfac_all = D.sum_over_ranks(float(int(np.sum(np.asarray(sub['lm_nobs']) - 1))))
    flops_all = D.sum_over_ranks(info['flops_by_kernel']["ba_big_schur_kernel"])          # the sharded part of the model
    ok_all = D.sum_over_ranks(float(ok))
    if rank != 0:
        h.close()
        return
    per_kernel = {}
    for k, (ms, n) in prof.items():
        if n == 0:
            continue
        fl = info['flops_by_kernel'][k] / (n / runs)
        avg = ms / n
        per_kernel[k] = {"launches_per_step": n // runs, "ms_per_launch": avg, "ms_per_step": ms / runs, "flops_per_launch": fl,
                         "achieved": fl / (avg * 1e-3) / 1e12 if avg > 0 else 0.0}
        per_kernel[k]["frac"] = per_kernel[k]["achieved"] / FP64_PEAK_TFLOPS
        per_kernel[k]["traffic"] = pmc_traffic(k)
    dom = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"])
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import ba_cpu
        pk = ba.PackedProblem(prob)
        ba_cpu.optimize(prob, margin_flag=ba.VG_MARGIN_NONE, packed=pk)
        ts = []
        st_o = sm_o = None
        for _ in range(2 if QUICK else 30):
            tc = time.perf_counter()
            st_o, sm_o, _ = ba_cpu.optimize(prob, margin_flag=ba.VG_MARGIN_NONE, packed=pk)
            ts.append(time.perf_counter() - tc)
        med = float(np.median(ts))
        cpu = {"value": 1.0 / med, "unit": "solves/s", "cores": 1, "kind": "port", "ms_per_solve": med * 1e3,
               "sample": "30 solves of the same 31-frame x 2000-landmark window (median), oracle/ba_cpu.cpp (restated single-thread "
                         "Ceres DENSE_SCHUR + dogleg path with runtime sizes: the reference fixes WINDOW_SIZE = 10 / NUM_OF_F = 1000 at "
                         "compile time and cannot run this window)",
               "final_cost": float(sm_o['final_cost'])}
        # parity of the timed result (rank 0's frames are every rank's frames)
        cpu["parity"] = {"final_cost_rel": abs(sm[0]['final_cost'] - sm_o['final_cost']) / sm_o['final_cost'],
                         "max_abs_pose_diff": float(np.abs(st[0]['pose'] - st_o['pose']).max()),
                         "same_iteration_flags": bool(list(sm[0]['it_flags'][:8]) == list(sm_o['it_flags'][:8]))}
    value = args.steps / elapsed
    out = {
        "metric": "sliding-window BA solves/sec, enlarged 31-frame x 2000-landmark window, landmark shards over the GPUs "
                  "(BASELINE.json configs[4]; NOT the headline metric, which is --config batch)",
        "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"ONE window of K={K} frames, {L} landmarks, {int(nfac_all)} projection factors, {K - 1} IMU factors, 15-dim prior on "
                               f"the oldest frame, max 8 dogleg iterations, no marginalization; landmarks in {world} contiguous shard(s) balanced by "
                               f"sum (6 n_l)^2; per iteration two all-reduces of {counts[0]} and {counts[1]} doubles (RCCL, in place, on the "
                               f"launch stream); reduced camera system factorised redundantly on every rank",
                   "parallelism": f"landmark shards x{world} + all-reduce of the reduced camera system" if world > 1 else "single rank (no collective)",
                   "allreduce_hook": hook_kind,
                   "valid_solves": int(ok_all)},
        "roofline": {"kernel": dom, "bound": "mfma", "achieved": per_kernel[dom]["achieved"], "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": per_kernel[dom]["frac"], "traffic": per_kernel[dom]["traffic"], "kernels": per_kernel,
                     "note": "rank 0; HIP events after every launch; the all-reduce in front of ba_solve_big_kernel / ba_big_step_kernel is "
                             "counted with that kernel; flops = SURVEY.md 8(d) model split per launch class (Schur-kernel flops of all "
                             f"ranks: {flops_all:.4g})"},
        "cpu_baseline": cpu,
        "marginalization_of_the_sharded_window": marg_info,
    }
    if cpu:
        out["speedup_vs_cpu"] = value / cpu["value"]
    print(json.dumps(out))
    h.close()


def csrc_tag():
    """Identity of the DEVICE code the library carries: sha1 over the .hip_fatbin section of vins-mono_amd/lib/libvinsgpu.so (the
    embedded gfx950 code objects; first 12 hex).  Byte-identical for the same kernel sources built by the in-tree Makefile -- a clean
    rebuild reproduces it -- whatever changes in host-only code, comments or headers that do not reach the generated code; any change
    of a kernel changes it, and so does another OBJDIR (hipcc derives a compilation-unit id from its command line, output path
    included, and that id is part of the code object: checked at the end of round 5).  (Until r03s the tag was a hash of the source text of csrc/, which host-side edits invalidated although the kernels the
    PMC pass had measured were unchanged.)"""
    import hashlib
    import struct
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vins-mono_amd", "lib", "libvinsgpu.so")
    try:
        b = open(lib, "rb").read()
        if b[:4] != b"\x7fELF" or b[4] != 2:
            return "not-elf64"
        shoff = struct.unpack_from("<Q", b, 0x28)[0]
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)

        def section(i):
            name, _, _, _, off, size = struct.unpack_from("<IIQQQQ", b, shoff + i * shentsize)
            return name, off, size
        _, stroff, _ = section(shstrndx)
        h = hashlib.sha1()
        found = False
        for i in range(shnum):
            name, off, size = section(i)
            if b[stroff + name:b.index(b"\0", stroff + name)] == b".hip_fatbin":
                h.update(b[off:off + size])
                found = True
        return h.hexdigest()[:12] if found else "no-fatbin"
    except (OSError, struct.error, ValueError):
        return "no-library"


_PMC = None


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` ('a+b' = sum over the kernels of a class) from the committed PMC summary
    (profiles/pmc_latest.json, written by profiles/run_pmc.sh from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
    this same command).  None when the summary was recorded for OTHER device code than the library holds (its `build` tag is
    csrc_tag() at recording time) or has no entry for the kernel."""
    global _PMC
    if _PMC is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_latest.json")
        try:
            with open(path) as f:
                _PMC = json.load(f)
        except (OSError, ValueError):
            _PMC = {}
        if _PMC.get("build") != csrc_tag():
            _PMC = {"kernels": {}, "stale": True}
    try:
        return sum(_PMC["kernels"][k]["hbm_bytes_per_launch"] for k in kernel.split("+"))
    except KeyError:
        return None


def pmc_field(kernel, field):
    """A derived SQ fraction of `kernel` from the committed PMC summary (profiles/summarize_counters.py --merge-into), or None."""
    pmc_traffic(kernel)                      # (loads the summary, drops it if it belongs to other device code)
    try:
        return _PMC["kernels"][kernel][field]
    except KeyError:
        return None


def sequence_parity(Rm, synth, n_seq, n_solves, lib_dropin):
    """Sequence-level parity (the cheap form of tests/manual/gpu_flip_stats.py): the reference's OWN per-frame loop
    (processIMU / processImage, estimator.cpp:81-215, compiled from /root/reference into oracle/_ref) over `n_seq` synthetic sequences
    of 10 + n_solves frames, once with its own optimization() and once with the product's drop-in body.  A 'flip' = a frame whose solve
    took another number of iterations or another accept / reject sequence than the reference's."""
    n_run = 10 + n_solves
    frames = flips = seq_flipped = 0
    err_no_flip = err_flip = 0.0

    def rel(a, b):
        return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))
    for s in range(n_seq):
        mp = 10.0 / 460.0 if s % 2 == 0 else 0.1                  # every other sequence also takes MARGIN_SECOND_NEW
        a = Rm.run_sequence(synth.SyntheticSequence(2000 + s, n_frames=n_run + 2, K=n_run + 2, L=500), n_run, L=Rm.lib(), min_parallax=mp, collect_priors=False)
        b = Rm.run_sequence(synth.SyntheticSequence(2000 + s, n_frames=n_run + 2, K=n_run + 2, L=500), n_run, L=lib_dropin, min_parallax=mp, collect_priors=False)
        flipped = False
        for r, g in zip(a, b):
            frames += 1
            e = max(rel(g['pose'][:, :3], r['pose'][:, :3]), float(np.abs(g['pose'][:, 3:] - r['pose'][:, 3:]).max()), rel(g['sb'][:, :3], r['sb'][:, :3]),
                    float(np.abs(g['sb'][:, 3:] - r['sb'][:, 3:]).max()))
            same = r['trace'].shape == g['trace'].shape and np.array_equal(r['trace'][:, :2], g['trace'][:, :2])
            if not same:
                flips += 1
                flipped = True
            if flipped:
                err_flip = max(err_flip, e)
            else:
                err_no_flip = max(err_no_flip, e)
        seq_flipped += int(flipped)
    return {"sequences": n_seq, "frames": frames, "flip_frames": flips, "flip_rate": flips / max(1, frames), "sequences_with_a_flip": seq_flipped,
            "max_err_no_flip": err_no_flip, "max_err_from_the_first_flip_on": err_flip,
            "what": f"the reference's processImage loop ({n_seq} synthetic sequences x {n_solves} solves) with its own optimization() (restated "
                    "Ceres inside oracle/_ref: PARITY UNPINNED against real Ceres) vs the same loop with the drop-in on the device; errors = max "
                    "of relative position, quaternion, relative velocity, bias differences over the window; north_star tolerance 1e-4; the "
                    "100-sequence statistics: profiles/r06_flip_stats.json"}


_KRES = None


def kernel_resources(name):
    """Registers / LDS / workgroup size of a kernel from the metadata notes of the code objects embedded in libvinsgpu.so, and the
    wavefronts per SIMD they allow (512 VGPRs per SIMD lane in granules of 8, 160 KB of LDS per CU, 8 wavefronts per SIMD at most).
    None when the ROCm binutils are not there."""
    global _KRES
    if _KRES is None:
        _KRES = {}
        import re
        import shutil
        import tempfile
        tools = "/opt/rocm/lib/llvm/bin"
        lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vins-mono_amd", "lib", "libvinsgpu.so")
        try:
            with tempfile.TemporaryDirectory() as tmp:
                shutil.copy(lib, os.path.join(tmp, "lib.so"))
                subprocess.run([os.path.join(tools, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True, timeout=120)
                for f in sorted(os.listdir(tmp)):
                    if "gfx950" not in f:
                        continue
                    txt = subprocess.run([os.path.join(tools, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True, capture_output=True,
                                         text=True, timeout=120).stdout
                    for blk in txt.split("  - .agpr_count:")[1:]:
                        def field(key):
                            m = re.search(r"\." + key + r":\s+(\S+)", blk)
                            return m.group(1) if m else None
                        nm = field("name")
                        if nm is None or field("vgpr_count") is None:
                            continue
                        vg, lds, wg = int(field("vgpr_count")), int(field("group_segment_fixed_size") or 0), int(field("max_flat_workgroup_size") or 64)
                        _KRES[nm] = {"vgprs": vg, "lds_static_bytes": lds, "workgroup": wg}
        except (OSError, subprocess.SubprocessError, ValueError):
            _KRES = {}
    r = _KRES.get(name)
    if not r:
        return None
    waves_wg = (r["workgroup"] + 63) // 64
    by_regs = min(8, 512 // max(8, (r["vgprs"] + 7) // 8 * 8))
    wgs_by_lds = (160 * 1024) // r["lds_static_bytes"] if r["lds_static_bytes"] else 1 << 20
    by_lds = wgs_by_lds * waves_wg / 4.0
    out = dict(r)
    out["waves_per_simd"] = float(min(by_regs, by_lds, 8))
    return out


def fe_traffic():
    """HBM bytes per FE step (one fe_lk launch + 3 fe_pyrdown; the frame is level 0 of its pyramid, there is no copy) from the
    committed PMC summary (per-launch figures of the PMC pass scale with the number of streams of THAT pass)."""
    t = [pmc_traffic("fe_lk_kernel"), pmc_traffic("fe_pyrdown_kernel")]
    return None if any(v is None for v in t) else t[0] + 3 * t[1]


def device_info(h=None):
    """What the box reports about its GPU (informational).  The boxes of the pool differ: single-window launches take the same time
    everywhere, launches that fill all 256 CUs with the BA kernels' latency-bound workgroups are up to 1.4 x slower on some of them
     -- this object is what a reader can hold such a run against."""
    info = {}
    try:
        import torch
        p = torch.cuda.get_device_properties(0)
        info.update({"name": p.name, "compute_units": p.multi_processor_count, "total_memory_GB": round(p.total_memory / 2**30, 1),
                     "gcn_arch": getattr(p, "gcnArchName", None)})
    except Exception as ex:                                   # noqa: BLE001
        info["torch"] = f"unavailable: {ex!r}"
    try:
        r = subprocess.run(["rocm-smi", "--showcomputepartition", "--showmemorypartition", "--showclocks", "--showperflevel", "--json"],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=20)
        j = json.loads(r.stdout) if r.stdout.strip().startswith("{") else {}
        card = j.get("card0", {})
        info["rocm_smi"] = {k: v for k, v in card.items() if any(t in k.lower() for t in ("partition", "sclk", "mclk", "fclk", "perf"))}
    except Exception as ex:                                   # noqa: BLE001
        info["rocm_smi"] = f"unavailable: {ex!r}"
    if h is not None and not QUICK:                           # (--emulated: the probe's FMA chains would be minutes of emulation)
        # the clock the box REALLY runs at (rocm-smi reports the same figures on boxes that run the BA kernels 1.4 x apart): a dependent
        # FP64 FMA has a fixed latency in core cycles, so ns per FMA ~ 1 / clock -- lone wavefront, and with every CU loaded
        try:
            info["clock_probe"] = h.probe_clocks()
        except Exception as ex:                               # noqa: BLE001
            info["clock_probe"] = f"unavailable: {ex!r}"
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=WINDOWS_PER_GPU)
    ap.add_argument("--in-flight", type=int, default=16, help="independent batches (HIP streams) the steps are spread over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--launch-mode", choices=["graph", "direct"], default=None,
                    help="VG_BA_LAUNCH_MODE for every handle of the run (default: the library's, see vg_ba_set_launch_mode)")
    ap.add_argument("--config", choices=["batch", "sharded"], default="batch",
                    help="batch = BASELINE configs[3] (the headline: independent EuRoC-shape windows, replicas over GPUs); sharded = "
                         "configs[4]: ONE enlarged 31-frame x 2000-landmark window, landmark shards over the GPUs, RCCL all-reduce "
                         "of the reduced camera system")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the "
                    "2-rank self-test on a 1-GPU box together with --share-device)")
    ap.add_argument("--share-device", action="store_true", help="self-test: all ranks use cuda:0")
    ap.add_argument("--rccl-hook", action="store_true", help="--config sharded with one rank: still route the reductions through RCCL")
    ap.add_argument("--profile-loop", action="store_true",
                    help="profiler passes: set-up, W warm-up steps, the K timed steps, then ONE short JSON line and exit -- none of the legs that "
                         "follow the timed region in a normal run (event passes, boundary loops, front end, CPU baseline), so that a rocprofv3 "
                         "kernel trace of the command holds the timed loop's launches only (plus one set-up solve per stream)")
    ap.add_argument("--quick-fe", action="store_true", help="profiler passes: 8 distinct image pairs instead of one per stream (generator time)")
    ap.add_argument("--emulated", action="store_true",
                    help="contract self-test WITHOUT a GPU (tests/test_bench_contract.py): the kernel sources under the CPU fiber emulator of "
                         "tests/simt, tiny loop counts; every number it prints is meaningless except that the line has the right shape")
    args = ap.parse_args()
    # N > 1: one process per GPU.  The driver starts the ranks itself (python -m torch.distributed.run ... bench.py --gpus N); a plain
    # `python bench.py --gpus N` starts them here, so that the command can never silently time ONE GPU and print n_gpus: 1.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # (dmabuf IPC: RCCL across processes needs it on this driver)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if int(os.environ.get("WORLD_SIZE", "1")) != max(1, args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: start one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or let bench.py do it")
    if args.launch_mode:
        os.environ["VG_BA_LAUNCH_MODE"] = args.launch_mode       # read by the library when a handle first launches

    import torch
    import __graft_entry__ as graft
    pkg = graft.load_package()
    from vins_mono_amd import ba, synth, dist_util as D
    rank, local_rank, world = D.env_rank()
    global QUICK, FE_CAMS, FE_DISTINCT
    if args.quick_fe:
        FE_DISTINCT = 8
    if args.emulated:
        # TEST INFRASTRUCTURE, never a measurement: the emulated library of tests/simt stands in for libvinsgpu.so so that the code of
        # this file (legs, JSON assembly) can be exercised where there is no GPU
        import ctypes
        import subprocess
        simt = os.path.join(ROOT, "tests", "simt")
        subprocess.run(["make", "-C", simt, "-j", str(os.cpu_count() or 4)], check=True, stdout=subprocess.DEVNULL)
        pkg._lib, pkg.LIB_PATH = ctypes.CDLL(os.path.join(simt, "_build", "libvinsgpu_simt.so"), mode=ctypes.RTLD_LOCAL), "emulated"
        QUICK, FE_CAMS = True, 2
        if world != 1:
            if args.backend == "nccl":
                raise SystemExit("--emulated with several ranks: --backend gloo (there is no GPU for RCCL)")
            D.init(args.backend, local_rank)         # tests/test_bench_contract.py: the N > 1 plumbing (rank seeds, barrier, max / sum over ranks)
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.share_device:
        local_rank = 0
    if not args.emulated:
        torch.cuda.set_device(local_rank)
        D.init(args.backend, local_rank)    # one process per GPU; RCCL only for barrier / max / sum of the timing
    cuda_sync = (lambda: None) if args.emulated else torch.cuda.synchronize

    if args.config == "sharded":
        bench_sharded(args, ba, synth, D, rank, world)
        D.finish()
        return

    h = ba.Handle()
    nwin = args.windows
    nfl = max(1, args.in_flight) if not QUICK else min(max(1, args.in_flight), 2)      # (--emulated: two streams exercise the rotation)
    # `nfl` independent batches of nwin windows each, one vg_handle (= one HIP stream) per batch: consecutive steps go to
    # different streams.  Since round 5 a solve workgroup is 4 wavefronts / 72 KB of LDS, so the solve kernels of two batches share
    # CUs (two windows per CU); the factor and marginalization kernels still take a CU each, and launches of other batches fill
    # their tails.  Four in flight: equal to two on the fast kind of box (149.5K vs 150.3K), +9 % on the slow kind (131.6K vs
    # 120.4K; profiles/r05k_in_flight_slow_box.json, r05l_ab_solve_two_per_cu.json); eight: another +4 % on the slow kind (136.6K ->
    # 142.3K, 4 / 6 / 8 / 4 / 8 in one call: profiles/r05x_in_flight_4_6_8.txt); sixteen: +4.7 % over eight on the slow kind (141.5K -> 148.2K),
    # +3 % on the fast kind (153.5K -> 158K; 24 and 32 fall off again: profiles/r05x_in_flight_8_to_32.txt) -- the default.
    handles = [h] + [ba.Handle() for _ in range(nfl - 1)]
    seed0 = D.window_seeds(rank, nwin)[0]
    probs, seqs = make_windows(h, ba, synth, nwin, seed0=seed0)
    packed = [ba.PackedProblem(p) for p in probs]
    flags = [ba.VG_MARGIN_OLD] * nwin
    for hh in handles:
        hh.ba_upload(packed, flags)                  # inputs now resident in HBM (every stream owns its copy of the batch)
    info = h.ba_info()

    def barrier():
        cuda_sync()
        D.barrier()

    def sync_all():
        for hh in handles:
            hh.sync()

    # set-up, before the W warm-up steps: every stream runs its batch once.  A handle's first launch loads code objects, sets kernel
    # attributes and allocates its scratch (hipMalloc synchronises the whole device): with more streams than warm-up steps (16 vs the
    # driver's W = 5) those first launches would fall into the timed region -- measured: 77K instead of 153K solves/s at W = 3.
    for hh in handles:
        hh.ba_run_async()
    sync_all()
    for k in range(args.warmup):
        handles[k % nfl].ba_run_async()
    sync_all()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        handles[k % nfl].ba_run_async()
    sync_all()
    cuda_sync()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = D.max_over_ranks(elapsed)
    if args.profile_loop:
        if rank == 0:
            print(json.dumps({"metric": "sliding-window BA solves/sec (profile loop: timed region only)", "value": world * nwin * args.steps / elapsed,
                              "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                              "batches_in_flight": nfl, "windows_per_gpu": nwin}))
        for hh in handles:
            hh.close()
        D.finish()
        return

    # the same loop ten times as long (>= 200 steps): the timed region above is ~50 ms, too short for round-to-round deltas of a
    # few per cent to mean much; reported next to `value`, never instead of it
    n_long = 2 if QUICK else max(200, 10 * args.steps)
    barrier()
    tl0 = time.perf_counter()
    for k in range(n_long):
        handles[k % nfl].ba_run_async()
    sync_all()
    cuda_sync()
    long_elapsed = D.max_over_ranks(time.perf_counter() - tl0)
    barrier()
    launch_stats = h.ba_launch_stats()               # mode in effect + graph launches / captures of the timed handle so far
    # one step at a time (no overlap between steps): the latency of a 256-window step
    ser = []
    for _ in range(max(3, min(args.steps, 10))):
        a, b = h.ba_run_timed()
        ser.append((a, b))
    solve_ms, marg_ms = float(np.mean([a for a, _ in ser])), float(np.mean([b for _, b in ser]))
    # per-kernel durations: a HIP event after every launch on the launch stream (outside the timed region)
    prof = {}
    for _ in range(max(3, min(args.steps, 10))):
        for k, (ms, n) in h.ba_run_profiled().items():
            a = prof.setdefault(k, [0.0, 0, 0])
            a[0] += ms; a[1] += n; a[2] += 1

    # the same event pass with TWO batches' worth of windows in one launch (the same windows twice): a kernel that owns a CU takes
    # twice as long, the solve kernel -- 4 wavefronts, 72 KB of LDS since round 5 -- runs its second 256 workgroups beside the first.
    # Outside the timed region; reported as roofline.two_per_cu (profiles/r05z_two_windows_per_cu.txt).
    prof2 = {}
    if not QUICK and rank == 0:
        try:
            h2 = ba.Handle()
            h2.ba_upload(packed + packed, flags + flags)
            h2.ba_run_profiled()
            for _ in range(3):
                for k, (ms, n) in h2.ba_run_profiled().items():
                    a = prof2.setdefault(k, [0.0, 0, 0])
                    a[0] += ms; a[1] += n; a[2] += 1
            h2.close()
        except Exception as ex:                       # noqa: BLE001  (informational)
            prof2 = {"error": repr(ex)}

    # boundary-inclusive rate (host buffers in, host buffers out: pack + H2D + all launches + D2H), NOT the metric
    up_ms, dn_ms = [], []
    for _ in range(3):
        h.ba_upload(packed, flags)
        up_ms.append(h.last_upload_call_ms)           # vg_ba_batch_upload: pack (host threads) + H2D from pinned staging
        h.ba_run_async()
        h.sync()                                      # so that the download call below does not include kernel time
        h.ba_download()
        dn_ms.append(h.last_download_call_ms)         # vg_ba_batch_download: D2H + unpack
    up_ms, dn_ms = float(np.median(up_ms)), float(np.median(dn_ms))
    # the same, software-pipelined over two handles (two streams, two sets of pinned staging buffers) by ONE host thread:
    # while batch i runs on the GPU the host packs + uploads batch i+1, then collects batch i -- the double-buffered
    # upload / run_async / download the async split of the ABI exists for
    ha = h
    hb = handles[1] if nfl > 1 else ba.Handle()
    if hb not in handles:
        hb.ba_upload(packed, flags)
    ha.ba_prepare_download(); hb.ba_prepare_download()
    nb_pipe = 1 if QUICK else 8

    def pipelined(nb):
        ha.ba_upload(packed, flags); ha.ba_run_async()
        cur, nxt = ha, hb
        for _ in range(nb):
            nxt.ba_upload(packed, flags); nxt.ba_run_async()
            rc = cur.ba_download_raw()
            if rc != 0:
                raise SystemExit(f"vg_ba_batch_download failed in the pipelined boundary loop: {rc}")
            cur, nxt = nxt, cur
        cur.ba_download_raw()
    pipelined(2)
    tb0 = time.perf_counter()
    pipelined(nb_pipe)
    overlapped_ms = (time.perf_counter() - tb0) / (nb_pipe + 1) * 1e3
    if hb not in handles:
        hb.close()
    # the same boundary with one HOST THREAD PER HANDLE (the calls release the GIL; the C-ABI is re-entrant across handles):
    # every thread loops upload -> run_async -> download on its own handle / stream / pinned staging, so the pack + H2D and
    # D2H + unpack of some batches always run beside the kernels of another
    import threading
    nthr = 2 if QUICK else 8
    th_handles = (handles + [ba.Handle() for _ in range(max(0, nthr - len(handles)))])[:nthr]
    for hh in th_handles:
        hh.ba_upload(packed, flags); hh.ba_prepare_download()
    nb_thr = 1 if QUICK else 16
    errs = []

    def boundary_worker(hh, nb):
        try:
            for _ in range(nb):
                hh.ba_upload(packed, flags); hh.ba_run_async()
                rc = hh.ba_download_raw()
                if rc != 0:
                    errs.append(rc)
        except Exception as ex:                       # noqa: BLE001
            errs.append(repr(ex))

    def threaded(nb):
        ts = [threading.Thread(target=boundary_worker, args=(hh, nb)) for hh in th_handles]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    threaded(1)
    tt0 = time.perf_counter()
    threaded(nb_thr)
    threaded_ms = (time.perf_counter() - tt0) / (nb_thr * nthr) * 1e3
    if errs:
        raise SystemExit(f"threaded boundary loop failed: {errs[:3]}")
    # the same threads over CHAINS of consecutive frames (what a caller that tracks sequences does): every thread walks its
    # handle through frames 2 .. 1+CHAIN_FRAMES of its 256 sequences; from the second frame on the prior is the one the
    # previous solve's marginalization left on the device (VG_PRIOR_RESIDENT) and only the states come back
    chain = make_chain(h, ba, seqs, packed, flags)
    dls = {}
    for hh in th_handles:
        for k, cb in enumerate(chain):                # (every handle's slots get the prior of frame k-1 from its own run)
            hh.ba_upload(cb, flags)
            dls[(id(hh), k)] = hh.ba_prepare_download()
            hh.ba_run_async()
        hh.sync()

    def chain_worker(hh, nb):
        try:
            for _ in range(nb):
                for k, cb in enumerate(chain):
                    hh.ba_upload(cb, flags); hh.ba_run_async()
                    rc = hh.ba_download_state_raw(dls[(id(hh), k)])
                    if rc != 0:
                        errs.append(rc)
        except Exception as ex:                       # noqa: BLE001
            errs.append(repr(ex))

    def chained(nb):
        ts = [threading.Thread(target=chain_worker, args=(hh, nb)) for hh in th_handles]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    chained(1)
    tc0 = time.perf_counter()
    nb_ch = 1 if QUICK else 5
    chained(nb_ch)
    chained_ms = (time.perf_counter() - tc0) / (nb_ch * CHAIN_FRAMES * nthr) * 1e3
    if errs:
        raise SystemExit(f"chained boundary loop failed: {errs[:3]}")
    for hh in th_handles:
        if hh not in handles:
            hh.close()
    # windows that stay on the device from frame to frame (vg_ba_seq_*): informational, never fails the bench
    try:
        resident_ms, resident_info = bench_resident(ba, synth, seqs, nthr=2 if QUICK else 4, rounds=1 if QUICK else 3)
    except Exception as ex:                               # noqa: BLE001
        resident_ms, resident_info = None, {"error": repr(ex)}
    h.ba_upload(packed, flags)
    h.ba_run_async()

    # sanity: results of the timed batch are valid
    st, sm, pr = h.ba_download()
    n_ok = sum(1 for s in sm if s['status'] == 0)

    out = None
    if rank == 0:
        total_solves = world * nwin * args.steps
        value = total_solves / elapsed
        # Roofline per kernel class (FP64: vector peak = MFMA peak = 78.6 TF on gfx950): algorithmic flops of SURVEY.md 8(d)
        # split per launch class (vg_ba_batch_flops_by_kernel) / average launch duration from the event pass above.
        # The top-level entry is the DOMINANT kernel: the class with the largest summed duration per step.
        per_kernel = {}
        for k, (ms, n, runs) in prof.items():
            if n == 0:
                continue
            fl = info['flops_by_kernel'][k] / (n / runs)                   # flops per launch (whole batch)
            avg = ms / n
            per_kernel[k] = {"launches_per_step": n // runs, "ms_per_launch": avg, "ms_per_step": ms / runs,
                             "flops_per_launch": fl, "achieved": fl / (avg * 1e-3) / 1e12 if avg > 0 else 0.0}
            per_kernel[k]["frac"] = per_kernel[k]["achieved"] / FP64_PEAK_TFLOPS
            per_kernel[k]["traffic"] = pmc_traffic(k)
        dom = max(per_kernel, key=lambda k: per_kernel[k]["ms_per_step"])
        step_ms = sum(v["ms_per_step"] for v in per_kernel.values())
        roofline = {
            "kernel": dom,
            "bound": "mfma",
            "achieved": per_kernel[dom]["achieved"],
            "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": per_kernel[dom]["frac"],
            "traffic": per_kernel[dom]["traffic"],
            "kernels": per_kernel,
            "whole_step": {"ms_serial": solve_ms + marg_ms, "flops": info['flops'],
                           "achieved_serial": info['flops'] / ((solve_ms + marg_ms) * 1e-3) / 1e12,
                           "achieved_timed_region": info['flops'] * args.steps / elapsed / 1e12,
                           "frac_timed_region": info['flops'] * args.steps / elapsed / 1e12 / FP64_PEAK_TFLOPS},
            "two_per_cu": ({"error": prof2["error"]} if "error" in prof2 else {
                "windows_per_launch": 2 * nwin,
                "ms_per_launch": {k: ms / n for k, (ms, n, runs) in prof2.items() if n},
                "ratio_to_one_batch": {k: (ms / n) / per_kernel[k]["ms_per_launch"] for k, (ms, n, runs) in prof2.items() if n and k in per_kernel},
                "dominant_kernel_achieved": (2 * per_kernel[dom]["flops_per_launch"] / (prof2[dom][0] / prof2[dom][1] * 1e-3) / 1e12) if dom in prof2 and prof2[dom][1] else None,
                "dominant_kernel_frac": (2 * per_kernel[dom]["flops_per_launch"] / (prof2[dom][0] / prof2[dom][1] * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if dom in prof2 and prof2[dom][1] else None,
                "what": "the same per-launch event pass with two batches' worth of windows in ONE launch: kernels that own a CU double, the "
                        "solve kernel's workgroups pair up on the CUs -- the dominant kernel at the operating point the batch runs it at "
                        "(several streams in flight); NOT the headline frac, which stays the stand-alone launch of one batch"} if prof2 else None),
            "reading": "frac = flops of the dominant kernel / its STAND-ALONE launch duration (one stream, launches not overlapped).  Since "
                       "round 5 the solve kernel is built to share a CU with a second window (4 wavefronts, 72 KB of LDS): alone it is slower "
                       "than the 8-wavefront kernel of round 4, in the batch two of them overlap -- the figure that shows what the "
                       "restructuring bought is whole_step.frac_timed_region (flops of a step / the timed region), not this one",
            "note": "ms_per_launch = HIP events after every launch on the launch stream (one stream, steps not overlapped), averaged "
                    "over the passes after the timed region; flops = SURVEY.md 8(d) model split per launch class; traffic = HBM bytes "
                    "per launch from the committed rocprofv3 PMC pass (profiles/pmc_latest.json, FETCH_SIZE x2 per the gfx950 note of "
                    "MI355X_MICROARCH.md + WRITE_SIZE), null when that file has no entry for the kernel",
            "algorithmic_bytes_per_batch": info['bytes_in'] + info['bytes_out'],
            "hbm_GBs_algorithmic": (info['bytes_in'] + info['bytes_out']) * args.steps / elapsed / 1e9,
        }
        # HBM bytes of a whole step from the PMC pass: per-launch traffic x launches per step, summed over the launch
        # classes; null while any class has no entry (summary recorded for other device code)
        # (a class 'a+b' is one launch of each kernel per pass and its traffic the sum of the two: counted once per pass)
        tr = [v["traffic"] * v["launches_per_step"] / len(k.split("+")) if v["traffic"] is not None else None for k, v in per_kernel.items()]
        roofline["traffic_per_step"] = None if any(t is None for t in tr) else float(sum(tr))
        roofline["traffic_per_step_over_algorithmic"] = (roofline["traffic_per_step"] / roofline["algorithmic_bytes_per_batch"]
                                                         if roofline["traffic_per_step"] else None)
        roofline["hbm_GBs_traffic_timed_region"] = (roofline["traffic_per_step"] * args.steps / elapsed / 1e9 if roofline["traffic_per_step"] else None)
        cpu = None
        parity = None
        if not args.no_cpu_baseline and world == 1:          # reported at N = 1 only (bench contract)
            from oracle import ba_cpu
            try:
                os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]})       # taskset -c <first allowed core>
                pinned = True
            except (AttributeError, OSError):
                pinned = False
            ncpu = min(nwin, 64)
            for i in range(3):                                                      # warm-up
                ba_cpu.time_optimize([packed[i % nwin]], [flags[i % nwin]])
            t_full, t_solve = [], []
            tstart = time.perf_counter()
            i = 0
            while (len(t_full) < (3 if QUICK else 50) or time.perf_counter() - tstart < (0.1 if QUICK else 10.0)) and len(t_full) < 400:
                t_full.append(ba_cpu.time_optimize([packed[i % ncpu]], [flags[i % ncpu]]))
                t_solve.append(ba_cpu.time_optimize([packed[i % ncpu]], [ba.VG_MARGIN_NONE]))
                i += 1
            if pinned:
                os.sched_setaffinity(0, set(range(os.cpu_count() or 1)))
            med_full, med_solve = float(np.median(t_full)), float(np.median(t_solve))
            # parity of the TIMED batch (all windows, not a sample): the same oracle, states within 1e-4 relative as north_star asks
            worst_pose = worst_sb = worst_cost = 0.0
            same_trace = 0
            for i in range(nwin):
                st_o, sm_o, _ = ba_cpu.optimize(probs[i], margin_flag=flags[i], packed=packed[i])
                scale_p = max(1.0, float(np.abs(st_o['pose'][:, :3]).max()))
                worst_pose = max(worst_pose, float(np.abs(st[i]['pose'][:, :3] - st_o['pose'][:, :3]).max()) / scale_p,
                                 float(np.abs(st[i]['pose'][:, 3:] - st_o['pose'][:, 3:]).max()))
                worst_sb = max(worst_sb, float(np.abs(st[i]['sb'] - st_o['sb']).max()) / max(1.0, float(np.abs(st_o['sb']).max())))
                worst_cost = max(worst_cost, abs(sm[i]['final_cost'] - sm_o['final_cost']) / max(sm_o['final_cost'], 1e-300))
                n_it = sm_o['num_iterations']
                same_trace += int(sm[i]['num_iterations'] == n_it and list(sm[i]['it_flags'][:n_it]) == list(sm_o['it_flags'][:n_it]))
            parity = {"windows": nwin, "max_rel_pose_error": worst_pose, "max_rel_speedbias_error": worst_sb,
                      "max_rel_final_cost_error": worst_cost, "identical_accept_reject_traces": same_trace,
                      "what": "every window of the timed batch against oracle/ba_cpu.cpp (positions relative to the window's extent, "
                              "quaternion components absolute); tolerance of north_star: 1e-4"}
            try:
                from oracle import ref as Rm
                if Rm.available():
                    parity["sequence"] = sequence_parity(Rm, synth, 2 if QUICK else 16, 2 if QUICK else 8, Rm.lib_simt() if args.emulated else Rm.lib_gpu())
            except Exception as ex:                       # noqa: BLE001  (informational: never fails the bench)
                parity["sequence"] = {"error": repr(ex)}
            cpu = {
                "value": 1.0 / med_full, "unit": "solves/s", "cores": 1, "kind": "port",
                "sample": f"median of {len(t_full)} single solves over {ncpu} of the {nwin} timed windows, pinned to one core "
                          f"(sched_setaffinity = taskset -c): oracle/ba_cpu.cpp (restated single-thread Ceres-equivalent "
                          f"DENSE_SCHUR + DOGLEG + marginalization; real Ceres/Eigen unavailable), host: {os.cpu_count()} cpus",
                "ms_per_solve": med_full * 1e3,
                "ms_solve_only": med_solve * 1e3,
                "ms_marginalization": (med_full - med_solve) * 1e3,
                "marginalization_note": "the reference spreads the marginalization's A = sum J^T J over 4 pthreads "
                                        "(marginalization_factor.h:13); this figure is single-thread",
            }
            # The reference's OWN translation units (estimator.cpp, factor/*.cpp compiled unchanged into oracle/_ref, see oracle/Makefile)
            # timed on the same windows -- for orientation only: they run on the stand-in headers of oracle/ref_stubs (an eager
            # matrix class without expression templates or vectorisation, a restated dense trust-region solver), so this is NOT
            # the speed of the reference on real Eigen + Ceres and is not the baseline.
            try:
                from oracle import ref as R
                if R.available():
                    R.configure_for(probs[0], None)
                    est = R.Estimator()
                    t_ref = []
                    for i in range(6):
                        est.load_window(probs[i % ncpu])
                        tr0 = time.perf_counter()
                        est.optimization(0)
                        t_ref.append(time.perf_counter() - tr0)
                    est.close()
                    cpu["reference_sources_on_stand_in_headers_ms"] = float(np.median(t_ref[1:])) * 1e3
            except Exception as ex:                       # noqa: BLE001  (informational: never fails the bench)
                cpu["reference_sources_on_stand_in_headers_ms"] = f"unavailable: {ex!r}"
        out = {
            "metric": "sliding-window BA solves/sec (Estimator::optimization: 8-iteration dogleg solve + marginalization)",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "value_long_run": world * nwin * n_long / long_elapsed,      # the regression signal: the same loop over >= 200 steps (long_run)
            "long_run": {"steps": n_long, "value": world * nwin * n_long / long_elapsed, "ms_per_step": long_elapsed / n_long * 1e3,
                         "what": "the same timed loop over ten times as many steps (run right after the timed region)"},
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"batch of {nwin} independent EuRoC-shape windows per step and GPU (K=11 frames, ~150 landmarks, "
                                   f"10 IMU factors, ~600 projection factors, 75-dim marginalization prior, max 8 iterations, "
                                   f"MARGIN_OLD marginalization); windows resident in HBM; consecutive steps are issued to "
                                   f"{nfl} HIP streams (independent batches in flight)",
                       "windows_per_gpu": nwin, "batches_in_flight": nfl, "launch": launch_stats,
                       "parallelism": f"independent batches x{world} (no collectives)", "valid_solves": n_ok},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "parity": parity,
            "device": device_info(h),
            "step_latency_ms": {"solve_pipeline": solve_ms, "marginalization": marg_ms, "total": solve_ms + marg_ms,
                                "what": "one 256-window step alone on the GPU (no overlap with other steps), HIP events"},
            "single_window_latency_ms": None,
            "host_boundary_inclusive": {"upload_call_ms": up_ms, "download_call_ms": dn_ms,
                                        "sync_ms_per_batch": up_ms + (solve_ms + marg_ms) + dn_ms,
                                        "sync_solves_per_s": nwin / ((up_ms + solve_ms + marg_ms + dn_ms) * 1e-3),
                                        "overlapped_ms_per_batch": overlapped_ms,
                                        "overlapped_solves_per_s": nwin / (overlapped_ms * 1e-3),
                                        "threaded_ms_per_batch": threaded_ms, "threaded_host_threads": nthr,
                                        "threaded_solves_per_s": nwin / (threaded_ms * 1e-3),
                                        "threaded_over_device_resident": (nwin / (threaded_ms * 1e-3)) / value,
                                        "env": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")},
                                        "chained_ms_per_batch": chained_ms, "chained_frames": CHAIN_FRAMES,
                                        "chained_solves_per_s": nwin / (chained_ms * 1e-3),
                                        "chained_over_device_resident": (nwin / (chained_ms * 1e-3)) / value,
                                        "resident_sequence_ms_per_batch": resident_ms,
                                        "resident_sequence_solves_per_s": (nwin / (resident_ms * 1e-3)) if resident_ms else None,
                                        "resident_sequence_over_device_resident": ((nwin / (resident_ms * 1e-3)) / value) if resident_ms else None,
                                        "resident_sequence": resident_info,
                                        "what": "host buffers in, host buffers out: vg_ba_batch_upload (pack + H2D) + all launches + "
                                                "vg_ba_batch_download (D2H + unpack), per GPU; sync = one batch at a time, nothing "
                                                "overlapped; overlapped = two handles (streams, pinned staging) driven by one host thread, the "
                                                "pack + upload of batch i+1 issued while batch i runs, then batch i downloaded; threaded = one host "
                                                "thread per handle (8 handles), each looping upload -> run -> download, so the host work of some "
                                                "batches always runs beside the kernels of another: the rate a caller with host buffers gets from one "
                                                "GPU; chained = the same 8 threads, each walking its handle through CHAIN_FRAMES consecutive frames "
                                                "of its 256 sequences: frame 2 uploads its prior, the following frames use the prior the previous "
                                                "solve's marginalization left in HBM (VG_PRIOR_RESIDENT) and download the states only; "
                                                "resident_sequence = windows that stay on the device (vg_ba_seq_*): 4 threads, per frame only the "
                                                "new observations / state guess / pre-integration go in and the states come back, the track "
                                                "bookkeeping (addFeatureCheckParallax, triangulate, problem tables, slideWindow) runs on the device; "
                                                "NOT the metric (bench contract: `value` is device-resident)"},
        }
    fe_out = bench_fe(h, synth, 2 if QUICK else max(args.steps, 10), args.warmup, rank, rank == 0 and world == 1 and not args.no_cpu_baseline)
    fe_out["value_all_gpus"] = D.sum_over_ranks(fe_out["value"])
    # single-window latency (configs[2]) on rank 0
    if rank == 0:
        out["fe"] = fe_out
        h.ba_upload([packed[0]], [ba.VG_MARGIN_OLD])
        for _ in range(3):
            h.ba_run_async()
        h.sync()
        lat = [h.ba_run_timed() for _ in range(2 if QUICK else 20)]
        out["single_window_latency_ms"] = float(np.median([a + b for a, b in lat]))
        out["single_window"] = {
            "solve_pipeline_ms": float(np.median([a for a, _ in lat])), "marginalization_ms": float(np.median([b for _, b in lat])),
            "what": "one window alone on the GPU, device-resident, HIP events; the marginalization result is only needed "
                    "by the NEXT frame's optimization()"}
        # the boundary-inclusive single call a drop-in Estimator::optimization() makes: vg_ba_optimize (pack + H2D + launches + D2H)
        calls = []
        for _ in range(2 if QUICK else 20):
            tc = time.perf_counter()
            h.ba_optimize(packed[0], ba.VG_MARGIN_OLD)
            calls.append((time.perf_counter() - tc) * 1e3)
        out["single_window"]["vg_ba_optimize_call_ms"] = float(np.median(calls))
        # the same in two parts (vg_ba_optimize_begin / _prior): time until the STATES are back on the host -- what
        # Estimator::optimization() waits for; the marginalization runs behind it and is collected by the next frame
        ready = [h.ba_optimize_split(packed[0], ba.VG_MARGIN_OLD)[3] for _ in range(2 if QUICK else 20)]
        out["single_window"]["states_on_host_ms"] = float(np.median(ready))
        if out["cpu_baseline"]:
            out["single_window_speedup_vs_cpu"] = out["cpu_baseline"]["ms_per_solve"] / out["single_window_latency_ms"]
            out["single_window"]["call_speedup_vs_cpu"] = out["cpu_baseline"]["ms_per_solve"] / out["single_window"]["vg_ba_optimize_call_ms"]
            out["single_window"]["states_speedup_vs_cpu_solve_only"] = out["cpu_baseline"]["ms_solve_only"] / out["single_window"]["states_on_host_ms"]
            out["batch_speedup_vs_cpu_per_gpu"] = out["value"] / world / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    for hh in handles[1:]:
        hh.close()
    h.close()
    D.finish()


if __name__ == "__main__":
    main()
