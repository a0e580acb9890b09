#!/bin/bash
# (run through gpurun: GRAFT_REPO_ROOT is the snapshot of the repo on the GPU box; default: this script's repo)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
export GRAFT_REPO_ROOT
# PMC passes over one ViT pass (tools/vit_step.py): per-kernel sums of the given counters.  tools/pmc_vit.sh <tag> <precision> <counters...>
R=$GRAFT_REPO_ROOT
TAG=$1; PREC=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$TAG -- python3 $R/tools/vit_step.py $PREC 512 1 > $R/gpurun_out/pmc_$TAG.log 2>&1
python3 - <<PY
import csv, glob, collections, re
path = glob.glob("$R/gpurun_out/pmc_$TAG/*/*counter_collection.csv")[0]
tot = collections.defaultdict(lambda: collections.Counter()); cnt = collections.Counter()
for r in csv.DictReader(open(path)):
    k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))[:60]
    tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k in sorted(tot, key=lambda k: -sum(tot[k].values()))[:8]:
    print(k, {c: f"{v:.4g} ({cnt[(k,c)]} disp)" for c, v in tot[k].items()})
PY
