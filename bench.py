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
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WINDOWS_PER_GPU = 256
FP64_PEAK_TFLOPS = 78.6      # MI355X FP64 vector = FP64 matrix peak (AMD datasheet)
HBM_PEAK_GBS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md


def make_windows(h, ba, synth, n, seed0):
    """n timed windows: for every seed, window 1 (no prior) is solved + marginalised by the PRODUCT path to
    create the prior, then window 2 (with prior) is assembled — the timed one (SURVEY.md 8(d) configs[2])."""
    seqs = [synth.SyntheticSequence(seed0 + s) for s in range(n)]
    first = [q.window(0) for q in seqs]
    h.ba_upload(first, [ba.VG_MARGIN_OLD] * n)
    h.ba_run_async()
    st, sm, pr = h.ba_download()
    assert all(s['status'] == 0 for s in sm), "window-1 solve failed"
    return [q.next_window(st[i], pr[i], 1) for i, q in enumerate(seqs)]


FE_CAMS = 64                 # independent camera streams per GPU in the front-end leg
FE_BYTES_PER_FEATURE = 8188  # (pyramid 592,200 B + LK 636,000 B) per 752x480 frame / 150 features
FE_GFTT_BYTES_PER_FRAME = 3609600


def bench_fe(h, synth, steps, warmup, rank, with_cpu):
    """Front-end leg (BASELINE.json configs[1]): per step and per stream, pyramid build of the new 752x480 frame +
    4-level pyramidal LK of 150 corners (frames resident in HBM, two alternating frames per stream)."""
    import numpy as np
    from vins_mono_amd import fe
    W, H, N = 752, 480, 150
    tr = fe.FrontEnd(h, W, H, FE_CAMS, N)
    base = [synth.synth_frame(1000 + rank * FE_CAMS + c) for c in range(min(FE_CAMS, 8))]
    nxt = [synth.warp_frame(b, 2000 + c) for c, b in enumerate(base)]
    fa = [base[c % len(base)] for c in range(FE_CAMS)]
    fb = [nxt[c % len(nxt)] for c in range(FE_CAMS)]
    tr.push_frames(fa)                      # slot 0 = frame A (and pyramid)
    tr.detect_upload([N] * FE_CAMS)
    tr.detect_async()
    corners = tr.detect_download()
    tr.upload_frames(fb)                    # slot 1 = frame B
    tr.track_upload(corners)
    slot = 1

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
        "roofline": {"kernel": "fe_lk_kernel (+ fe_pyrdown_kernel x3 + fe_copy_kernel)", "bound": "hbm",
                     "achieved": nfeat * FE_BYTES_PER_FEATURE * steps / (ev_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "traffic": fe_traffic(), "event_ms_per_step": ev_ms / steps,
                     "gftt_GBs": FE_CAMS * FE_GFTT_BYTES_PER_FRAME / (gftt_ms * 1e-3) / 1e9},
    }
    out["roofline"]["frac"] = out["roofline"]["achieved"] / HBM_PEAK_GBS
    if with_cpu:
        from oracle import fe_cpu
        t = time.perf_counter()
        reps = 0
        while time.perf_counter() - t < 5.0:
            fe_cpu.lk(fa[reps % len(fa)], fb[reps % len(fb)], corners[reps % len(corners)])
            reps += 1
        dt = time.perf_counter() - t
        t2 = time.perf_counter()
        g = 0
        while time.perf_counter() - t2 < 3.0:
            fe_cpu.gftt(fa[g % len(fa)], N)
            g += 1
        out["cpu_baseline"] = {"value": reps * N / dt, "unit": "features/s", "cores": 1, "kind": "port",
                               "sample": f"{reps} frame pairs x {N} corners, oracle/fe_cpu.cpp (restated single-thread OpenCV-equivalent "
                                         f"calcOpticalFlowPyrLK incl. both pyramids + Scharr; real OpenCV unavailable)",
                               "gftt_frames_per_s": g / (time.perf_counter() - t2)}
        out["speedup_vs_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary (profiles/pmc_latest.json, written by
    profiles/summarize_pmc.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_latest.json")
    try:
        with open(path) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        return None


def fe_traffic():
    """HBM bytes per FE step (one fe_lk launch + 3 fe_pyrdown + 1 fe_copy) from the committed PMC summary."""
    t = [pmc_traffic("fe_lk_kernel"), pmc_traffic("fe_pyrdown_kernel"), pmc_traffic("fe_copy_kernel")]
    return None if any(v is None for v in t) else t[0] + 3 * t[1] + t[2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--windows", type=int, default=WINDOWS_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the "
                    "2-rank self-test on a 1-GPU box together with --share-device)")
    ap.add_argument("--share-device", action="store_true", help="self-test: all ranks use cuda:0")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as graft
    graft.load_package()
    from vins_mono_amd import ba, synth, dist_util as D
    rank, local_rank, world = D.env_rank()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    D.init(args.backend, local_rank)    # one process per GPU; RCCL only for barrier / max / sum of the timing

    h = ba.Handle()
    nwin = args.windows
    probs = make_windows(h, ba, synth, nwin, seed0=D.window_seeds(rank, nwin)[0])
    packed = [ba.PackedProblem(p) for p in probs]
    flags = [ba.VG_MARGIN_OLD] * nwin
    h.ba_upload(packed, flags)                       # inputs now resident in HBM
    info = h.ba_info()

    def barrier():
        torch.cuda.synchronize()
        D.barrier()

    for _ in range(args.warmup):
        h.ba_run_async()
    h.sync()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        h.ba_run_async()
    h.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = D.max_over_ranks(elapsed)

    # per-kernel durations from HIP events on the launch stream (not part of the timed region)
    ks, km = [], []
    for _ in range(max(3, min(args.steps, 10))):
        a, b = h.ba_run_timed()
        ks.append(a)
        km.append(b)
    solve_ms, marg_ms = float(np.mean(ks)), float(np.mean(km))

    # boundary-inclusive rate (host buffers in, host buffers out: pack + H2D + both launches + D2H), NOT the metric
    up_ms, dn_ms = [], []
    for _ in range(3):
        h.ba_upload(packed, flags)
        up_ms.append(h.last_upload_call_ms)           # vg_ba_batch_upload: pack (host threads) + H2D from pinned staging
        h.ba_run_async()
        h.sync()                                      # so that the download call below does not include kernel time
        h.ba_download()
        dn_ms.append(h.last_download_call_ms)         # vg_ba_batch_download: D2H + unpack
    up_ms, dn_ms = float(np.median(up_ms)), float(np.median(dn_ms))

    # sanity: results of the timed batch are valid
    st, sm, pr = h.ba_download()
    n_ok = sum(1 for s in sm if s['status'] == 0)

    out = None
    if rank == 0:
        total_solves = world * nwin * args.steps
        value = total_solves / elapsed
        flops_per_launch = info['flops']                 # algorithmic FLOP model of SURVEY.md 8(d), whole batch
        # One "step" = two launches; the roofline is booked per kernel (FP64: vector peak = MFMA peak = 78.6 TF on gfx950)
        # with the algorithmic flop model of SURVEY.md 8(d) split per launch; the top-level entry is the DOMINANT kernel
        # (longest average launch), the other kernel and the pair are listed beside it.
        per_kernel = {
            "ba_solve_kernel": {"ms": solve_ms, "flops": info['flops_solve']},
            "ba_marg_kernel": {"ms": marg_ms, "flops": info['flops_marg']},
        }
        for k, v in per_kernel.items():
            v["achieved"] = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
            v["frac"] = v["achieved"] / FP64_PEAK_TFLOPS
            v["traffic"] = pmc_traffic(k)
        dom = max(per_kernel, key=lambda k: per_kernel[k]["ms"])
        roofline = {
            "kernel": dom,
            "bound": "mfma",
            "achieved": per_kernel[dom]["achieved"],
            "peak": FP64_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": per_kernel[dom]["frac"],
            "traffic": per_kernel[dom]["traffic"],
            "kernels": per_kernel,
            "pair": {"ms": solve_ms + marg_ms, "flops": info['flops'],
                     "achieved": info['flops'] / ((solve_ms + marg_ms) * 1e-3) / 1e12,
                     "frac": info['flops'] / ((solve_ms + marg_ms) * 1e-3) / 1e12 / FP64_PEAK_TFLOPS},
            "note": "durations = HIP events on the launch stream, averaged over the launches after the timed region; "
                    "traffic = HBM bytes per launch from the committed rocprofv3 PMC pass (profiles/, FETCH_SIZE x2 per the "
                    "gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE), null when no PMC file is present",
            "algorithmic_bytes_per_batch": info['bytes_in'] + info['bytes_out'],
            "hbm_GBs_algorithmic": (info['bytes_in'] + info['bytes_out']) / ((solve_ms + marg_ms) * 1e-3) / 1e9,
        }
        cpu = None
        if not args.no_cpu_baseline and world == 1:          # reported at N = 1 only (bench contract)
            from oracle import ba_cpu
            ncpu = min(nwin, 64)
            reps = 1
            t_cpu = ba_cpu.time_optimize(packed[:ncpu], flags[:ncpu], repeats=1)
            while t_cpu * (reps + 1) < 10.0 and reps < 20:
                reps += 1
            if reps > 1:
                t_cpu = ba_cpu.time_optimize(packed[:ncpu], flags[:ncpu], repeats=reps) / reps
            cpu = {
                "value": ncpu / t_cpu, "unit": "solves/s", "cores": 1, "kind": "port",
                "sample": f"{ncpu} of the {nwin} timed windows x {reps} passes, oracle/ba_cpu.cpp (restated single-thread "
                          f"Ceres-equivalent DENSE_SCHUR+DOGLEG + marginalization; real Ceres/Eigen unavailable), "
                          f"host: {os.cpu_count()} cpus",
                "ms_per_solve": t_cpu / ncpu * 1e3,
            }
        out = {
            "metric": "sliding-window BA solves/sec (Estimator::optimization: 8-iteration dogleg solve + marginalization)",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"batch of {nwin} independent EuRoC-shape windows per GPU (K=11 frames, ~150 landmarks, "
                                   f"10 IMU factors, ~600 projection factors, 75-dim marginalization prior, max 8 iterations, "
                                   f"MARGIN_OLD marginalization); windows resident in HBM",
                       "windows_per_gpu": nwin, "parallelism": f"independent batches x{world} (no collectives)",
                       "valid_solves": n_ok},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "single_window_latency_ms": None,
            "host_boundary_inclusive": {"upload_call_ms": up_ms, "download_call_ms": dn_ms,
                                        "sync_ms_per_batch": up_ms + (solve_ms + marg_ms) + dn_ms,
                                        "sync_solves_per_s": nwin / ((up_ms + solve_ms + marg_ms + dn_ms) * 1e-3),
                                        "what": "host buffers in, host buffers out, nothing overlapped: vg_ba_batch_upload (pack + "
                                                "H2D) + both kernels + vg_ba_batch_download (D2H + unpack), per GPU; NOT the metric"},
        }
    fe_out = bench_fe(h, synth, max(args.steps, 10), args.warmup, rank, rank == 0 and world == 1 and not args.no_cpu_baseline)
    fe_out["value_all_gpus"] = D.sum_over_ranks(fe_out["value"])
    # single-window latency (configs[2]) on rank 0
    if rank == 0:
        out["fe"] = fe_out
        h.ba_upload([packed[0]], [ba.VG_MARGIN_OLD])
        for _ in range(3):
            h.ba_run_async()
        h.sync()
        lat = [sum(h.ba_run_timed()) for _ in range(10)]
        out["single_window_latency_ms"] = float(np.median(lat))
        if out["cpu_baseline"]:
            out["single_window_speedup_vs_cpu"] = out["cpu_baseline"]["ms_per_solve"] / out["single_window_latency_ms"]
            out["batch_speedup_vs_cpu_per_gpu"] = out["value"] / world / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    h.close()
    D.finish()


if __name__ == "__main__":
    main()
