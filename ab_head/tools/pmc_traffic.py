"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv) into per-kernel HBM bytes per launch.

    python tools/pmc_traffic.py <dir_fetch> <dir_write> [out.json]

Units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM): both counters are in KiB; FETCH_SIZE reports
half of the bytes of wide coalesced reads on gfx950 and is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import csv
import glob
import json
import sys
from collections import defaultdict


def per_kernel(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        k = k.split("(")[0]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return acc


fe, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(fe, key=lambda k: -fe[k][1]):
    n = fe[k][0]
    fb = 2.0 * fe[k][1] * 1024 / n
    wb = wr[k][1] * 1024 / wr[k][0] if k in wr and wr[k][0] else 0.0
    out[k] = {"launches": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
json.dump(out, open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout, indent=1)
for k, v in list(out.items())[:14]:
    print(f"{v['hbm_bytes_per_launch']/1e6:10.2f} MB/launch (fetch {v['fetch_bytes_per_launch']/1e6:8.2f} write {v['write_bytes_per_launch']/1e6:8.2f}) x{v['launches']:5d}  {k[:70]}", file=sys.stderr)
