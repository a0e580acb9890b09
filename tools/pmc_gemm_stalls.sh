#!/bin/bash
# (run through gpurun: GRAFT_REPO_ROOT is the snapshot of the repo on the GPU box; default: this script's repo)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
export GRAFT_REPO_ROOT
# Where does the K loop of gemm_x6 wait?  Three PMC passes (each alone, kernel-trace only) over one ViT-B pass of 1024 fragments,
# summed over the gemm_x6 dispatches:   tools/pmc_gemm_stalls.sh TAG   -> gpurun_out/pmc_stalls_TAG.txt
R=$GRAFT_REPO_ROOT
TAG=$1
O=$R/gpurun_out/pmc_stalls_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --pmc $2 --output-format csv -d $O/$1 -- python3 $R/tools/vit_step.py ${PREC:-f16x2} 1024 1 > $O/$1.log 2>&1; }
run sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE"
run ta "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
run tcp "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
run tcc "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
python3 - $O > $R/gpurun_out/pmc_stalls_$TAG.txt <<'PY'
import collections, csv, glob, os, re, sys
root = sys.argv[1]
for p in ("sq", "ta", "tcp", "tcc"):
    files = glob.glob(os.path.join(root, p, "*", "*counter_collection.csv"))
    if not files:
        print(p, "no output"); continue
    acc = collections.defaultdict(collections.Counter); n = collections.Counter(); seen = set()
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        m = re.search(r"relax::(\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if not m: continue
        k = m.group(1)
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen:
            seen.add((k, r["Dispatch_Id"])); n[k] += 1
    print(f"== pass {p}")
    for k in sorted(acc, key=lambda k: -acc[k]["GRBM_GUI_ACTIVE"])[:4]:
        c = acc[k]; cyc = c["GRBM_GUI_ACTIVE"] / 8 or 1
        print(f"{k}  dispatches {n[k]}  cycles {cyc:.3e}")
        for name, v in sorted(c.items()):
            if name != "GRBM_GUI_ACTIVE":
                print(f"    {name:44s} {v:.4e}   per elapsed cycle {v / cyc:10.3f}   per CU-cycle {v / cyc / 256:8.4f}")
PY
cat $R/gpurun_out/pmc_stalls_$TAG.txt
