"""One training step of a rocprofv3 --kernel-trace, kernel by kernel: start (us from the step's first kernel), duration, stream, gap to the
previous kernel of the same stream, and how many kernels of OTHER streams overlap it.  usage: step_trace.py DIR [step_from_end=2]

Steps are delimited by the generator's optimiser launch (the last adam_apply_kernel of a step on the main stream)."""
import csv
import glob
import sys

d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append([int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "?"))])
rows.sort()


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]


# the main stream = the stream with the most conv_patch time
from collections import defaultdict
busy = defaultdict(int)
for s, e, n, st in rows:
    busy[st] += e - s
main = max(busy, key=busy.get)
marks = [i for i, r in enumerate(rows) if r[3] == main and "adam_apply" in r[2]]
a, b = marks[-back - 1] + 1, marks[-back] + 1
step = rows[a:b]
t0 = step[0][0]
print(f"# step of {len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us; main stream = {main}; streams: {sorted(set(r[3] for r in step))}")
last_end = {}
for i, (s, e, n, st) in enumerate(step):
    ov = sum(1 for (s2, e2, n2, st2) in step[max(0, i - 40):i + 40] if st2 != st and s2 < e and e2 > s)
    gap = (s - last_end[st]) / 1e3 if st in last_end else 0.0
    last_end[st] = e
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} us  st{st:>2} gap{gap:7.1f} ov{ov}  {'' if st == main else '        '}{short(n)}")
