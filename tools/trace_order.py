#!/usr/bin/env python3
"""Lists the dispatches of the LAST pass in a rocprofv3 kernel-trace CSV in launch order: tools/trace_order.py TRACE.csv FIRST_KERNEL
(a pass starts at the last dispatch whose name contains FIRST_KERNEL).  Runs anywhere."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = max(i for i, r in enumerate(rows) if sys.argv[2] in r["Kernel_Name"])
total = 0.0
print(f"{'#':>3} {'us':>9}  kernel")
for i, r in enumerate(rows[first:]):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total += us
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void relax::", "").replace("relax::", "")
    grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
    print(f"{i:>3} {us:9.1f}  {name}  grid {grid} wg {wg}")
print(f"total {total / 1e3:.3f} ms over {len(rows) - first} dispatches")
