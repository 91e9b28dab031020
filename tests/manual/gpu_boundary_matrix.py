"""Where the threaded host boundary loses time: the same 4-thread loop with parts of it switched off."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as graft
graft.load_package()
from vins_mono_amd import ba, synth
import bench
n = 256
h0 = ba.Handle()
probs, seqs = bench.make_windows(h0, ba, synth, n, seed0=1000)
flags = [ba.VG_MARGIN_OLD] * n
pb = ba.PackedBatch(probs)
h0.ba_upload(pb, flags); h0.ba_run_async(); st, sm, pr = h0.ba_download()
nxt = ba.PackedBatch([q.next_window(st[i], 'resident', 2) for i, q in enumerate(seqs)])


def run(nthr, mode, nb=int(os.environ.get('NB', '8'))):
    hs = [ba.Handle() for _ in range(nthr)]
    dls = []
    for hh in hs:
        hh.ba_upload(pb, flags); hh.ba_run_async(); hh.sync()
        if mode.startswith("res"):
            d0 = hh.ba_prepare_download()
            hh.ba_upload(nxt, flags); hh.ba_run_async(); hh.sync()
            dls.append((d0, hh.ba_prepare_download()))
        else:
            dls.append(hh.ba_prepare_download())

    def worker(hh, dl, k):
        for _ in range(k):
            if mode == "run":
                hh.ba_run_async(); hh.sync()
            elif mode == "up+run":
                hh.ba_upload(pb, flags); hh.ba_run_async(); hh.sync()
            elif mode == "run+down":
                hh.ba_run_async(); hh.ba_download_raw()
            elif mode == "run+state":
                hh.ba_run_async(); hh.ba_download_state_raw(dl)
            elif mode == "full":
                hh.ba_upload(pb, flags); hh.ba_run_async(); hh.ba_download_raw()
            elif mode == "full-state":
                hh.ba_upload(pb, flags); hh.ba_run_async(); hh.ba_download_state_raw(dl)
            elif mode == "res":          # alternate: frame 2 (host prior) then frame 3 (resident prior), states only
                hh.ba_upload(pb, flags); hh.ba_run_async(); hh.ba_download_state_raw(dl[0])
                hh.ba_upload(nxt, flags); hh.ba_run_async(); hh.ba_download_state_raw(dl[1])
            elif mode == "upload-only":
                hh.ba_upload(pb, flags)

    def go(k):
        ts = [threading.Thread(target=worker, args=(hh, dl, k)) for hh, dl in zip(hs, dls)]
        for t in ts: t.start()
        for t in ts: t.join()
    go(1)
    import gc
    if os.environ.get('NOGC'): gc.disable()
    t0 = time.perf_counter(); go(nb); dt = time.perf_counter() - t0
    gc.enable()
    per = dt / (nb * nthr * (2 if mode == "res" else 1)) * 1e3
    print(f"threads {nthr}  {mode:11s}  {per:6.3f} ms/batch  {n / per * 1e-3 * 1e3:9.0f} solves/s", flush=True)
    for hh in hs:
        hh.close()


modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ("run", "upload-only", "up+run", "run+state", "run+down", "full-state", "full", "res")
thr = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (1, 2, 4, 8)
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES"), "VG_PACK_THREADS", os.environ.get("VG_PACK_THREADS"))
for mode in modes:
    for nthr in thr:
        run(nthr, mode)
