"""One frame's launches from a rocprofv3 kernel trace: per stream (queue), start / end relative to the frame's first launch,
and for every traversal launch what ran beside it.  usage: python3 tools/kernel_timeline.py kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    name = r["Kernel_Name"]
    short = "T" if "wf_travq" in name else "F" if "wf_advance" in name and "true>" in name.replace(" ", "") and name.replace(" ", "").endswith("true>(rtk::Scene,rtk::Frame,rtk::WfState)") else "A" if "wf_advance" in name else None
    if "wf_advance" in name:
        short = "F" if name.replace(" ", "").split("wf_advance<")[1].split(">")[0].endswith("true") else "A"
    if short is None:
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", r.get("Stream_Id", "?"))))
ev.sort()
# frames: a frame starts with launches F on both queues; split on gaps > 20 us without any running launch
frames, cur, last_end = [], [], None
for e in ev:
    if last_end is not None and e[0] - last_end > 20000 and cur:
        frames.append(cur); cur = []
    cur.append(e); last_end = max(last_end or 0, e[1])
if cur: frames.append(cur)
frames = [f for f in frames if len(f) >= 20]
print("frames found:", len(frames), "launches per frame:", collections.Counter(len(f) for f in frames))
f = frames[len(frames) // 2]
t0 = f[0][0]
queues = sorted(set(e[3] for e in f))
print("frame span us: %.1f" % ((max(e[1] for e in f) - t0) / 1e3))
for q in queues:
    print("queue", q)
    print("  " + " ".join("%s[%.0f-%.0f]" % (e[2], (e[0] - t0) / 1e3, (e[1] - t0) / 1e3) for e in f if e[3] == q))
# busy accounting: time with 0 / 1 / 2 launches running, and by pair of kinds
pts = sorted(set([e[0] for e in f] + [e[1] for e in f]))
acc = collections.Counter()
for a, b in zip(pts, pts[1:]):
    run = sorted(e[2] for e in f if e[0] <= a and e[1] >= b)
    acc["+".join(run) or "idle"] += (b - a) / 1e3
for k, v in sorted(acc.items(), key=lambda x: -x[1]):
    print("  %-8s %.1f us" % (k, v))
# over all frames
tot = collections.Counter()
for f in frames:
    pts = sorted(set([e[0] for e in f] + [e[1] for e in f]))
    for a, b in zip(pts, pts[1:]):
        run = sorted(e[2] for e in f if e[0] <= a and e[1] >= b)
        tot["+".join(run) or "idle"] += (b - a) / 1e3 / len(frames)
print("mean over frames:", {k: round(v, 1) for k, v in sorted(tot.items(), key=lambda x: -x[1])})

# the whole trace between its 20 % and 80 % points in time (frames in flight overlap: there are no gaps to split frames at)
lo = ev[0][0] + (ev[-1][1] - ev[0][0]) // 5
hi = ev[0][0] + (ev[-1][1] - ev[0][0]) * 4 // 5
win = [e for e in ev if e[1] > lo and e[0] < hi]
pts = sorted(set([max(e[0], lo) for e in win] + [min(e[1], hi) for e in win]))
acc = collections.Counter()
for a, b in zip(pts, pts[1:]):
    run = sorted(e[2] for e in win if e[0] <= a and e[1] >= b)
    acc["+".join(run) or "idle"] += (b - a)
tot = float(hi - lo)
print("steady window (%.1f ms): share of time by what runs: %s" % (tot / 1e6, {k: round(v / tot, 3) for k, v in sorted(acc.items(), key=lambda x: -x[1])}))
nF = sum(1 for e in win if e[2] == "F")
print("frames in the window (F launches / 2): %.1f -> %.4f ms per frame" % (nF / 2, tot / 1e6 / max(nF / 2, 1)))
