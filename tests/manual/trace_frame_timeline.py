"""MANUAL: per-frame timeline of a single front-end stream from a rocprofv3 kernel + memory-copy trace (CSV output): splits the
event list into frames at every H2D copy of >= 300 KB (the image) and prints, for the last few published (long) and other frames,
each kernel / copy with its start relative to the frame's first event, its duration and the idle gap before it.
usage: trace_frame_timeline.py <dir> [n_frames_to_print]"""
import csv, glob, os, sys
d = sys.argv[1]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:32], 0))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        nb = 0
        for k in ("Size", "Bytes", "size"):
            if k in r and r[k]:
                nb = int(r[k]); break
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?").replace("MEMORY_COPY_", "")[:18], nb))
ev.sort()
# frame starts: a gap of > 150 us before an event (the host's python loop between frames) -- robust against the copy size column
frames, cur = [], []
for i, e in enumerate(ev):
    if cur and e[0] - max(x[1] for x in cur) > 150000 and len(cur) >= 3:
        frames.append(cur); cur = []
    cur.append(e)
if cur: frames.append(cur)
frames = frames[len(frames) // 2:]
long_f = [f for f in frames if any(e[2].startswith("fe_mineig") for e in f)]
short_f = [f for f in frames if not any(e[2].startswith("fe_mineig") for e in f) and any(e[2].startswith("fe_lk") for e in f)]
def span(f): return (max(e[1] for e in f) - f[0][0]) * 1e-3
import statistics
for name, fs in (("published", long_f), ("other", short_f)):
    if not fs: continue
    print("%s frames: %d, device span first event -> last end: median %.1f us; busy (sum of durations) median %.1f us; events median %d" % (
        name, len(fs), statistics.median(span(f) for f in fs), statistics.median(sum(e[1] - e[0] for e in f) * 1e-3 for f in fs),
        statistics.median(len(f) for f in fs)))
    for f in fs[-nshow:]:
        print("  ---- frame: %d events, span %.1f us" % (len(f), span(f)))
        last = f[0][0]
        for s, e, n, nb in f:
            print("   +%8.1f us  gap %7.1f  dur %7.1f  %s%s" % ((s - f[0][0]) * 1e-3, (s - last) * 1e-3, (e - s) * 1e-3, n, "  %d B" % nb if nb else ""))
            last = max(last, e)
