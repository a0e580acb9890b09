#!/usr/bin/env python3
"""GPU-side time of the last pass in a rocprofv3 kernel-trace CSV: tools/trace_span.py TRACE.csv DISPATCHES_PER_PASS
Prints the span from the first dispatch's start to the last one's end, the sum of the kernel durations, the idle time between
dispatches, and the durations summed by kernel name (a launch-bound host loop shows as idle time, not as kernel time)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2])
last = rows[-n:]
span = (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e6
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / 1e6
by = collections.defaultdict(lambda: [0, 0.0])
for r in last:
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void relax::", "").replace("relax::", "")
    by[name][0] += 1
    by[name][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print(f"{n} dispatches: span {span:.2f} ms, kernels {busy:.2f} ms, idle between dispatches {span - busy:.2f} ms")
for name, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1]):
    print(f"  {t:8.3f} ms  {c:5d} x  {name}")
