#!/usr/bin/env python3
"""Reads a RELAX_X6_STAMP_DUMP file (stamps build of gemm_x6.hip) and reports, per launch, how the two workgroups that share
a CU overlap: fraction of K-loop time spent while the co-resident workgroup is also in its K loop."""
import struct
import sys
from collections import defaultdict

import numpy as np

data = open(sys.argv[1], "rb").read()
off = 0
n = 0
while off < len(data):
    M, N, K, BM, BN, units, full, stagger = struct.unpack_from("8i", data, off)
    off += 32
    rec = np.frombuffer(data, dtype=np.uint64, count=units * 8, offset=off).reshape(units, 8)
    off += units * 64
    n += 1
    t0, t1, t2, t3, hw, xcc = (rec[:full, i].astype(np.int64) for i in range(6))
    base = t0.min()
    cu = ((xcc & 0xf) << 16) | (((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5))   # XCC | SE | SH | CU
    slot = hw & 0xf
    groups = defaultdict(list)
    for i in range(full):
        groups[int(cu[i])].append(i)
    both = alone = epi_both = 0
    for c, idx in groups.items():
        ev = []
        for i in idx:
            ev.append((t1[i], +1, 0)); ev.append((t2[i], -1, 0))     # K loop
            ev.append((t2[i], +1, 1)); ev.append((t3[i], -1, 1))     # epilogue
        ev.sort()
        nk = ne = 0
        last = ev[0][0]
        for t, d, kind in ev:
            dt = t - last
            if nk >= 2: both += dt * 2
            elif nk == 1: alone += dt
            if nk == 0 and ne >= 1: epi_both += dt
            last = t
            if kind == 0: nk += d
            else: ne += d
    first = np.argsort(t0)[:min(full, 600)]
    print(f"launch {n}: {M}x{N}x{K} tile {BM}x{BN} units {units} stagger {stagger}: CUs seen {len(groups)}, WG per CU "
          f"{full / max(len(groups), 1):.1f}; K-loop WG-time with a co-resident K loop {both / (both + alone + 1e-9):.2f}; "
          f"time with no K loop running (epilogues only) {epi_both / (t3.max() - base):.2f} of the launch x CUs {len(groups)}; "
          f"slots of the first 512 WGs: {np.bincount(slot[first].astype(int), minlength=4)[:4]}; "
          f"start spread of first 512: {np.sort(t0)[min(full, 512) - 1] - base} cycles; launch length {t3.max() - base}")
