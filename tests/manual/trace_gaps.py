"""Reads a rocprofv3 kernel trace (+ memory copy trace) CSV pair and prints how busy the GPU was: union of kernel intervals,
the largest gaps and what ran around them.  usage: trace_gaps.py <dir>"""
import csv, glob, sys, os
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
mt = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
ev = []
for f in kt:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:40], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
for f in mt:
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "M", r.get("Direction", "?")[:40], "-", r.get("Stream_Id", "?")))
ev.sort()
ks = [e for e in ev if e[2] == "K"]
print("kernels", len(ks), "copies", len(ev) - len(ks))
# analyse the last 60 % of the kernel timeline (steady state)
t0, t1 = ks[0][0], max(e[1] for e in ks)
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
lo = t0 + int((t1 - t0) * frac)
hi = t1 - int((t1 - t0) * 0.03)
ks2 = [e for e in ks if e[0] >= lo and e[1] <= hi]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, _, name, q, st in ks2:
    if cur_s is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e, s, name))
        cur_s, cur_e = s, e
busy += cur_e - cur_s
span = ks2[-1][1] - ks2[0][0]
print("span %.2f ms, kernels busy (union) %.2f ms = %.1f %%" % (span * 1e-6, busy * 1e-6, 100.0 * busy / span))
gaps.sort(reverse=True)
print("idle time in gaps > 20 us: %.2f ms in %d gaps; > 5 us: %.2f ms" % (sum(g[0] for g in gaps if g[0] > 20000) * 1e-6, sum(1 for g in gaps if g[0] > 20000),
      sum(g[0] for g in gaps if g[0] > 5000) * 1e-6))
import collections
nxt = collections.Counter()
for g in gaps:
    if g[0] > 5000:
        nxt[g[3]] += g[0]
print("idle time by the kernel that ended the gap:")
for k, v in nxt.most_common(8):
    print("   %-40s %.2f ms" % (k, v * 1e-6))
# per kernel name: total time
tot = collections.Counter(); cnt = collections.Counter()
for s, e, _, name, q, st in ks2:
    tot[name] += e - s; cnt[name] += 1
print("kernel time (sum over streams, may overlap):")
for k, v in tot.most_common(8):
    print("   %-40s %.2f ms  n=%d  avg %.1f us" % (k, v * 1e-6, cnt[k], v / cnt[k] * 1e-3))
print("queues used:", sorted(set(e[4] for e in ks2)))

# the largest gaps with the copies that overlap them
cp = [e for e in ev if e[2] == "M"]
print("largest gaps (us) and the copies in flight during them:")
for g in gaps[:6]:
    inside = [(c[3], (c[1] - c[0]) // 1000) for c in cp if c[0] < g[2] and c[1] > g[1]]
    print("   %.0f us, ended by %s; copies: %s" % (g[0] * 1e-3, g[3], inside[:6]))
