"""Averages rocprofv3 --pmc counter_collection CSVs per kernel: pmc_avg.py <dir> [kernel substring]"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in acc.items():
    if sub in k:
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:32s} {sum(v) / len(v):16.0f}  (n={len(v)})")
