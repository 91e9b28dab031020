"""Per-window statistics of ba_marg_kernel over the frames of bench.py's chain (profile build, VG_DEBUG_MARG lines parsed)."""
import os, sys, re, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "child":
    os.environ["VG_DEBUG_MARG"] = "1"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import numpy as np
    import __graft_entry__ as g
    pkg = g.load_package()
    pkg.LIB_PATH = os.path.join(os.path.dirname(pkg.LIB_PATH), "libvinsgpu_prof.so")
    from vins_mono_amd import ba, synth
    import bench
    n = 256
    h = ba.Handle()
    probs, seqs = bench.make_windows(h, ba, synth, n, seed0=1000)
    flags = [ba.VG_MARGIN_OLD] * n
    cur = probs
    for k in range(4):
        print("FRAME", k, "m", " ".join(str(15 + int((np.asarray(p['lm_start']) == 0).sum())) for p in cur), file=sys.stderr, flush=True)
        h.ba_upload(cur, flags); h.ba_run_async()
        st, sm, pr = h.ba_download()
        print("ENDFRAME", k, "status", sum(1 for s in sm if s['status'] != 0), "prior none", sum(1 for p in pr if p is None), file=sys.stderr, flush=True)
        cur = [q.next_window(st[i], pr[i], k + 2) for i, q in enumerate(seqs)]
    sys.exit(0)
out = subprocess.run([sys.executable, __file__, "child"], stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
frame, ms, rows = None, [], []
for line in out.splitlines():
    if line.startswith("FRAME"):
        t = line.split(); frame = int(t[1]); ms = [int(v) for v in t[3:]]; rows = []
    elif line.startswith("[marg]") and frame is not None:
        v = [int(x) for x in re.findall(r"-?\d+", line)]
        rows.append(v)
    elif line.startswith("ENDFRAME"):
        import numpy as np
        r = np.array(rows)
        # columns: eig1 sweeps, attempts, eig2 sweeps, attempts, kcyc, setup, prior, imu, proj, eig1, schur, eig2, out   (digits of 'eig1'/'eig2' names come first)
        print(line, "| windows reported", len(rows))
        names = re.findall(r"[a-z0-9]+(?==|\s-?\d)", "")
        kc = r[:, -9]
        order = np.argsort(kc)
        print("   kernel kcyc: median %d, p90 %d, max %d" % (np.median(kc), np.percentile(kc, 90), kc.max()))
        for i in list(order[-3:]) + [order[len(order) // 2]]:
            print("   window", i, "m", ms[i] if i < len(ms) else "?", "row", rows[i])
        frame = None
print(out[-1500:] if "Traceback" in out else "")
