#!/usr/bin/env python3
"""Reads a RELAX_X6_STAMP_DUMP file (tools/build_ablations.sh x6stamps) and reports, per launch shape, what a CU does between the
tiles it runs: prologue / K loop / epilogue of a tile (s_memtime ticks) and the gap from the end of one tile's epilogue to the start
of the next tile on the same CU (workgroup dispatch), one workgroup per CU launches only:  tools/x6_gaps.py DUMP"""
import struct
import sys
from collections import defaultdict

import numpy as np

data = open(sys.argv[1], "rb").read()
off = 0
agg = defaultdict(list)
while off < len(data):
    M, N, K, BM, BN, units, full, _ = struct.unpack_from("8i", data, off)
    off += 32
    rec = np.frombuffer(data, dtype=np.uint64, count=units * 8, offset=off).reshape(units, 8)
    off += units * 64
    if full < 512 or BM * BN != 65536:
        continue
    t0, t1, t2, t3, hw, xcc = (rec[:full, i].astype(np.int64) for i in range(6))
    cu = ((xcc & 0xf) << 16) | (((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5))
    gaps = []
    for c in np.unique(cu):
        idx = np.where(cu == c)[0]
        idx = idx[np.argsort(t0[idx])]
        gaps += list(t0[idx[1:]] - t3[idx[:-1]])
    gaps = np.array(gaps)
    agg[(M, N, K)].append((np.mean(t1 - t0), np.mean(t2 - t1), np.mean(t3 - t2), np.median(gaps), np.mean(gaps),
                           t3.max() - t0.min(), full / len(np.unique(cu))))
for (M, N, K), v in sorted(agg.items()):
    a = np.mean(np.array(v), axis=0)
    tile = a[0] + a[1] + a[2] + a[4]
    print(f"{M}x{N}x{K}: {len(v)} launches; per tile: prologue {a[0]:.0f}  K loop {a[1]:.0f}  epilogue {a[2]:.0f}  gap to the next tile "
          f"(median {a[3]:.0f}, mean {a[4]:.0f}) = {100 * a[4] / tile:.1f} % of a tile slot; launch {a[5]:.0f} ticks for {a[6]:.1f} tiles per CU")
