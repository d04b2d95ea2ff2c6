"""Summarises a rocprofv3 --kernel-trace --stats CSV directory into a small text table (kept under profiles/)."""
import csv
import glob
import sys

d, steps = sys.argv[1], int(sys.argv[2])
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary ({f.split('/')[-1]}); {steps} steps profiled")
print(f"total kernel time {tot/1e6:.2f} ms = {tot/1e6/steps:.2f} ms/step")
print(f"{'ms/step':>9} {'%':>6} {'calls/step':>10} {'avg_us':>9}  kernel")
for r in rows[:40]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:110]
    print(f"{float(r['TotalDurationNs'])/1e6/steps:9.3f} {float(r['Percentage']):6.2f} {int(r['Calls'])/steps:10.1f} {float(r['AverageNs'])/1e3:9.1f}  {n}")
