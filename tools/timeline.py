"""Timeline analysis of a rocprofv3 --kernel-trace CSV: how busy the GPU is, how much of the time two or three streams really overlap,
which kernels sit alone on the critical path and where the idle gaps are.  usage: timeline.py DIR [skip_frac=0.3]

The window analysed is the trace minus its first `skip_frac` (warm-up, plan building) and its last 2 %.
"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo, hi = t0 + (t1 - t0) * skip, t1 - (t1 - t0) * 0.02
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
span = rows[-1][1] - rows[0][0]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


# sweep: concurrency levels, exclusive time per kernel name
ev = []
for i, (s, e, n, q, st) in enumerate(rows):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
live = set()
level_time = defaultdict(int)
excl = defaultdict(int)
tot = defaultdict(int)
prev = ev[0][0]
gaps = []
for t, kind, i in ev:
    dt = t - prev
    if dt > 0:
        level_time[len(live)] += dt
        if len(live) == 1:
            excl[short(rows[next(iter(live))][2])] += dt
        if len(live) == 0 and dt > 3000:
            gaps.append((dt, t))
    prev = t
    if kind == 1:
        live.add(i)
    else:
        live.discard(i)
for s, e, n, q, st in rows:
    tot[short(n)] += e - s
queues = defaultdict(int)
for s, e, n, q, st in rows:
    queues[(q, st)] += e - s

print(f"# {f.split('/')[-1]}: window {span/1e6:.1f} ms, {len(rows)} dispatches")
print("concurrency  share of wall time")
for k in sorted(level_time):
    print(f"  {k} kernels   {100*level_time[k]/span:6.2f} %   {level_time[k]/1e6:8.2f} ms")
print("queue/stream busy (sum of kernel durations / window)")
for k, v in sorted(queues.items(), key=lambda kv: -kv[1]):
    print(f"  queue {k[0]} stream {k[1]}: {100*v/span:6.2f} %")
print(f"{'alone %wall':>11} {'total %wall':>11} {'alone/total':>11}  kernel")
for n, v in sorted(tot.items(), key=lambda kv: -excl.get(kv[0], 0))[:28]:
    print(f"{100*excl.get(n,0)/span:11.2f} {100*v/span:11.2f} {excl.get(n,0)/v:11.2f}  {n}")
gaps.sort(reverse=True)
print(f"idle gaps > 3 us: {len(gaps)}, total {sum(g[0] for g in gaps)/1e6:.2f} ms ({100*sum(g[0] for g in gaps)/span:.2f} % of window); largest (us):",
      [round(g[0]/1e3, 1) for g in gaps[:12]])
# which kernel follows an idle gap most often
after = defaultdict(lambda: [0, 0])
starts = {r[0]: short(r[2]) for r in rows}
for dt, t in gaps:
    n = starts.get(t)
    if n:
        after[n][0] += 1; after[n][1] += dt
print("kernels that start after an idle gap (count, total us):")
for n, (c, tt) in sorted(after.items(), key=lambda kv: -kv[1][1])[:10]:
    print(f"  {c:5d} {tt/1e3:9.1f}  {n}")
