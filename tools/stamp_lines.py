#!/usr/bin/env python3
"""Averages the per-launch lines a stamped library prints (tools/abl/gemm_h2_stamps.hip, gemm_x6_stamps.hip) per launch shape: stdin -> table."""
import re
import sys
from collections import defaultdict

agg = defaultdict(list)
for line in sys.stdin:
    m = re.match(r"(\w+ \d+x\d+x\d+ .*?): cycles per tile: prologue (\d+), K loop (\d+) \((\d+) steps, (\d+) per step.*?epilogue (\d+)", line)
    if m:
        agg[m.group(1)].append([int(m.group(i)) for i in (2, 3, 4, 5, 6)])
for k, v in sorted(agg.items()):
    n = len(v)
    a = [sum(x[i] for x in v) / n for i in range(5)]
    print(f"{k}: {n} launches; per tile: prologue {a[0]:.0f}  K loop {a[1]:.0f} ({a[2]:.0f} steps, {a[3]:.0f} per 16 k)  epilogue {a[4]:.0f}  "
          f"= {a[0] + a[1] + a[4]:.0f}; loop share {a[1] / (a[0] + a[1] + a[4]):.2f}")
