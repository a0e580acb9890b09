#!/bin/bash
# round 4: PMC passes (each alone, kernel-trace only) over the Farneback stage on one 2160p clip: HBM traffic per kernel and SQ ratios
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="python3 $R/tools/flow_step.py 2160 3840 32 1 2"
rm -rf /tmp/fpmc
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fpmc/fetch -- $T > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/fpmc/write -- $T > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d /tmp/fpmc/sq -- $T > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d /tmp/fpmc/sq2 -- $T > /tmp/fpmc_sq2.log 2>&1 || tail -3 /tmp/fpmc_sq2.log
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/fpmc/tcc -- $T > /dev/null 2>&1
python3 - <<PY | tee $O/flow_pmc.txt
import csv, glob, collections, re
def load(d):
    f = glob.glob("/tmp/fpmc/%s/*/*counter_collection.csv" % d)
    tot = collections.defaultdict(lambda: collections.Counter()); n = collections.defaultdict(set)
    if not f: return tot, n
    for r in csv.DictReader(open(f[0])):
        m = re.match(r"(?:void )?(relax::\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if not m: continue
        tot[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); n[m.group(1)].add(r["Dispatch_Id"])
    return tot, n
def times(d):
    f = glob.glob("/tmp/fpmc/%s/*/*kernel_trace.csv" % d)
    t = collections.Counter(); c = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        m = re.match(r"(?:void )?(relax::\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if m: t[m.group(1)] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])); c[m.group(1)] += 1
    return t, c
fe, n = load("fetch"); wr, _ = load("write"); sq, _ = load("sq"); sq2, _ = load("sq2"); tc, _ = load("tcc")
tm, cnt = times("fetch")
print("Farneback stage, one 32-pair 2160p clip x 2 passes; rocprofv3 --pmc passes, each alone (FETCH_SIZE doubled per the gfx950 guide)")
print("%-34s %5s %9s %10s %9s %7s %7s %7s %7s %7s" % ("kernel", "disp", "avg us", "MB/disp", "TB/s", "L2hit", "active", "waitI", "wait", "ldsconf"))
for k in sorted(tm, key=lambda k: -tm[k]):
    d = max(cnt[k], 1)
    mb = (2 * fe[k]["FETCH_SIZE"] + wr[k]["WRITE_SIZE"]) * 1024 / d / 1e6
    us = tm[k] / d / 1e3
    s = sq[k]; wc = max(s["SQ_WAVE_CYCLES"], 1)
    hit = tc[k]["TCC_HIT_sum"] / max(tc[k]["TCC_HIT_sum"] + tc[k]["TCC_MISS_sum"], 1)
    print("%-34s %5d %9.1f %10.1f %9.2f %7.2f %7.2f %7.2f %7.2f %7.3f" % (k[:34], d, us, mb, mb / us, hit, s["SQ_ACTIVE_INST_ANY"] / wc, s["SQ_WAIT_INST_ANY"] / wc, s["SQ_WAIT_ANY"] / wc, s["SQ_LDS_BANK_CONFLICT"] / max(s["SQ_BUSY_CYCLES"], 1)))
    if sq2[k]:
        print("    extra:", {kk: round(v / d) for kk, v in sq2[k].items()})
PY
