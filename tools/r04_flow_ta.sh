#!/bin/bash
# round 4: is the load unit what the Farneback kernels wait for?  SQ-side counters of the texture-address FIFOs (one pass, kernel-trace only;
# the TA_* / TCP_* block counters abort rocprofv3 on this image - signal 6 - and are not collected)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
T="python3 $R/tools/flow_step.py 2160 3840 4 1 1"
rm -rf /tmp/fta
timeout 150 rocprofv3 --kernel-trace --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/fta/b -- $T > /tmp/fta_b.log 2>&1 || { echo "pass b failed / timed out"; tail -3 /tmp/fta_b.log; }
python3 - <<PY
import csv, glob, collections, re
tot = collections.defaultdict(collections.Counter); disp = collections.defaultdict(set)
for d in "abc":
    for f in glob.glob("/tmp/fta/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            m = re.match(r"(?:void )?(relax::\w+(?:<[^>]*>)?)", r["Kernel_Name"])
            if m: tot[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"]); disp[m.group(1)].add(r["Dispatch_Id"])
print("Farneback kernels, one 4-pair 2160p clip: counter sums over all dispatches of a kernel (TA / TCP counters are summed over the units)")
for k in sorted(tot, key=lambda k: -tot[k].get("GRBM_GUI_ACTIVE", 0)):
    c = tot[k]
    if c.get("GRBM_GUI_ACTIVE", 0) < 1e5: continue
    g = c["GRBM_GUI_ACTIVE"]
    print("%-36s" % k[:36], "TA addr FIFO full / active VMEM %.2f" % (c["SQ_VMEM_TA_ADDR_FIFO_FULL"] / max(c["SQ_ACTIVE_INST_VMEM"], 1)), " cmd FIFO full %.2f" % (c["SQ_VMEM_TA_CMD_FIFO_FULL"] / max(c["SQ_ACTIVE_INST_VMEM"], 1)),
          " VMEM rd %d wr %d" % (c["SQ_INSTS_VMEM_RD"], c["SQ_INSTS_VMEM_WR"]),)
PY
