#!/usr/bin/env python3
"""Summarises the rocprofv3 PMC passes of bench.py into profiles/:
   python tools/summarize_pmc.py hbm  FETCH_DIR WRITE_DIR TAG [CLIPS_PER_STEP]   -> profiles/r01_hbm_traffic_TAG.txt + r01_hbm_traffic.json
   python tools/summarize_pmc.py sq   SQ_DIR TAG                                 -> profiles/r01_pmc_sq_bench_TAG.txt
   python tools/summarize_pmc.py hbm2 FETCH_DIR WRITE_DIR TAG WORKLOAD CLIPS PRECISION KERNEL_PREFIX [JSON]   (round 2+: RELAX_ROUND=r02)
   python tools/summarize_pmc.py tcc  TCC_DIR TAG DESCRIPTION                    -> profiles/rNN_l2_hit_rate_TAG.txt
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE is doubled in the corrected figure (gfx950 reports half of a 16-B
per lane streaming read, MI355X_MICROARCH.md, HBM / rocprofv3 section).  Passes are collected separately (one --pmc each)."""
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    m = re.match(r"(?:void )?(relax::\w+(?:<[^>]*>)?)", name)
    return m.group(1) if m else None


def newest(dirname):
    """gpurun merges every call's output into the same directory: take the most recent pass, never an arbitrary one."""
    return max(glob.glob(os.path.join(dirname, "*", "*counter_collection.csv")), key=os.path.getmtime)


def per_kernel(dirname, counter):
    path = newest(dirname)
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if k:
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return tot, cnt


def hbm2(fetch_dir, write_dir, tag, workload="config3", clips=8, precision="bf16x6", main_prefix="relax::gemm_x6", out_json=None):
    """Round 2: same method, the dominant kernel is named by prefix; writes profiles/r02_hbm_traffic_TAG.txt (+ the json bench.py reads)."""
    f, n = per_kernel(fetch_dir, "FETCH_SIZE")
    w, _ = per_kernel(write_dir, "WRITE_SIZE")
    lines = [f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --workload {workload} --steps 2 "
             f"--warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d` ({clips} clips per step, precision {precision})",
             "KiB summed over the dispatches as reported; FETCH_SIZE is doubled in the corrected figure (gfx950 reports half of a "
             "16-B/lane streaming read, MI355X_MICROARCH.md)"]
    main_bytes, main_n = 0.0, 0
    for k in sorted(f, key=lambda k: -f[k]):
        lines.append(f"{k:70s} dispatches {n[k]:4d}  fetch_KiB(raw) {f[k]:12.0f}  write_KiB {w.get(k, 0):12.0f}  corrected MB/dispatch "
                     f"{(2 * f[k] + w.get(k, 0)) * 1024 / n[k] / 1e6:10.1f}")
        if any(k.startswith(pfx) for pfx in main_prefix.split("|")):     # several kernels: prefixes separated by |
            main_bytes += (2 * f[k] + w.get(k, 0)) * 1024
            main_n += n[k]
    per = main_bytes / max(main_n, 1)
    lines.append(f"{main_prefix}*: {main_n} dispatches, corrected HBM bytes per dispatch = (2*FETCH + WRITE) = {per / 1e6:.1f} MB")
    rnd = os.environ.get("RELAX_ROUND", "r02")
    out = os.path.join(ROOT, "profiles", f"{rnd}_hbm_traffic_{tag}.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    if out_json:
        json.dump({"workload": workload, "clips_per_step": clips, "precision": precision, "hbm_bytes_per_launch": per,
                   "kernel": main_prefix + "*",
                   "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes on `bench.py --steps 2 "
                             "--warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d`; bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024 "
                             "(gfx950 FETCH_SIZE reports half of 16-B/lane streaming reads)",
                   "source": f"profiles/{rnd}_hbm_traffic_{tag}.txt"}, open(os.path.join(ROOT, "profiles", out_json), "w"), indent=1)
    print("\n".join(lines))


def hbm(fetch_dir, write_dir, tag, clips=8):
    f, n = per_kernel(fetch_dir, "FETCH_SIZE")
    w, _ = per_kernel(write_dir, "WRITE_SIZE")
    lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `bench.py --steps 2 --warmup 1 --no-cpu-baseline "
             f"--no-fast-mode --no-h2d` ({clips} clips per step)",
             "KiB summed over the dispatches as reported; FETCH_SIZE is doubled in the corrected figure (gfx950 reports half of a "
             "16-B/lane streaming read, MI355X_MICROARCH.md)"]
    gemm_bytes, gemm_n = 0.0, 0
    for k in sorted(f, key=lambda k: -f[k]):
        lines.append(f"{k:70s} dispatches {n[k]:4d}  fetch_KiB(raw) {f[k]:12.0f}  write_KiB {w.get(k, 0):12.0f}")
        if k.startswith("relax::conv_gemm_f32"):
            gemm_bytes += (2 * f[k] + w.get(k, 0)) * 1024
            gemm_n += n[k]
    per = gemm_bytes / max(gemm_n, 1)
    lines.append(f"contraction kernels: {gemm_n} dispatches, corrected HBM bytes per dispatch = (2*FETCH + WRITE) = {per / 1e6:.1f} MB")
    ps = [k for k in f if k.startswith("relax::patch_score")]
    if ps:
        k = ps[0]
        lines.append(f"patch_score: corrected HBM bytes per dispatch = {(2 * f[k] + w.get(k, 0)) * 1024 / n[k] / 1e6:.1f} MB "
                     "(algorithmic 2*1080*1920*3*32 + scores = 398.4 MB per 32-pair clip)")
    out = os.path.join(ROOT, "profiles", f"r01_hbm_traffic_{tag}.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    json.dump({"workload": "config3", "clips_per_step": clips, "hbm_bytes_per_launch": per,
               "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes on `bench.py --steps 2 "
                         "--warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d`; conv_gemm_f32 dispatches only; bytes = (2*FETCH_SIZE + "
                         "WRITE_SIZE) KiB * 1024 (gfx950 FETCH_SIZE reports half of 16-B/lane streaming reads)",
               "source": f"profiles/r01_hbm_traffic_{tag}.txt"}, open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


def sq(sq_dir, tag):
    path = newest(sq_dir)
    acc = collections.defaultdict(collections.Counter)
    dur = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if not k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    lines = ["rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY "
             "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE (own pass) on `bench.py --steps 2 --warmup 1 --no-cpu-baseline "
             "--no-fast-mode --no-h2d`; totals per kernel over the run",
             "cycles = GRBM_GUI_ACTIVE / 8 XCDs; clock = cycles / time; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles * 1024 SIMDs); "
             "waves/SIMD = 4 * SQ_WAVE_CYCLES / (cycles * 1024); wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES; "
             "lds_conflict = SQ_LDS_BANK_CONFLICT / (cycles * 256 CUs)"]
    for k in sorted(acc, key=lambda k: -dur[k])[:12]:
        c = acc[k]
        cyc = (c["GRBM_GUI_ACTIVE"] or 1) / 8
        lines.append(f"{k:66s} time {dur[k] / 1e6:8.2f} ms  clock {cyc / dur[k]:.2f} GHz  MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}  "
                     f"waves/SIMD {4 * c['SQ_WAVE_CYCLES'] / (cyc * 1024):.2f}  wait_any {c['SQ_WAIT_ANY'] / (c['SQ_WAVE_CYCLES'] or 1):.2f}  "
                     f"lds_conflict_cycles/CU/elapsed {c['SQ_LDS_BANK_CONFLICT'] / (cyc * 256):.3f}")
    out = os.path.join(ROOT, "profiles", f"{os.environ.get('RELAX_ROUND', 'r01')}_pmc_sq_bench_{tag}.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


def tcc(tcc_dir, tag, what):
    hit, n = per_kernel(tcc_dir, "TCC_HIT_sum")
    miss, _ = per_kernel(tcc_dir, "TCC_MISS_sum")
    lines = [f"rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum (own pass) on `bench.py --steps 2 --warmup 1 --no-cpu-baseline "
             f"--no-fast-mode --no-h2d` ({what}): L2 hit rate per kernel"]
    for k in sorted(hit, key=lambda k: -(hit[k] + miss[k]))[:10]:
        lines.append(f"{k:62s} hits {hit[k]:.3e} misses {miss[k]:.3e} hit rate {hit[k] / max(hit[k] + miss[k], 1):.3f}")
    out = os.path.join(ROOT, "profiles", f"{os.environ.get('RELAX_ROUND', 'r01')}_l2_hit_rate_{tag}.txt")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    if sys.argv[1] == "hbm":
        hbm(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 8)
    elif sys.argv[1] == "hbm2":   # hbm2 FETCH_DIR WRITE_DIR TAG WORKLOAD CLIPS PRECISION MAIN_PREFIX [OUT_JSON]
        hbm2(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), sys.argv[7], sys.argv[8],
             sys.argv[9] if len(sys.argv) > 9 else None)
    elif sys.argv[1] == "tcc":    # tcc TCC_DIR TAG DESCRIPTION
        tcc(sys.argv[2], sys.argv[3], sys.argv[4])
    else:
        sq(sys.argv[2], sys.argv[3])
