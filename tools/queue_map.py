"""Which hardware queue each HIP stream of a profiled run landed on (rocprofv3 --kernel-trace csv dir): stream -> queue, kernel time share."""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
t = defaultdict(int)
for r in csv.DictReader(open(f)):
    t[(r.get("Stream_Id", "?"), r.get("Queue_Id", "?"))] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(t.values())
print(" ".join(f"stream{s}->q{q}:{100*v/tot:.0f}%" for (s, q), v in sorted(t.items(), key=lambda kv: -kv[1])))
