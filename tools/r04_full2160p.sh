#!/bin/bash
# round 4: the config-5 shape (2160p, 32 pairs, full ReLaX with flow): bench line + rocprofv3 kernel stats
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python bench.py --workload full2160p --clips-per-step 8 --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-other-workloads --no-measure-traffic > $O/bench_full2160p_$TAG.json 2> $O/bench_full2160p_$TAG.err
python - <<PY
import json
r=json.load(open("$O/bench_full2160p_$TAG.json"))
print("full2160p", round(r["value"],2), "clips/s", round(r["ms_per_step"],1), "ms/step", "flow stage", r.get("roofline_flow_stage",{}).get("kernel_time_share_of_step"))
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fp2 -- python3 $R/bench.py --workload full2160p --clips-per-step 2 --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-other-workloads --no-measure-traffic > /dev/null 2>&1
f=$(find /tmp/fp2 -name "*kernel_stats.csv" | head -1)
cp $f $O/full2160p_kernel_stats_$TAG.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/full2160p_kernel_stats_$TAG.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
flow=sum(float(r["TotalDurationNs"]) for r in rows if any(k in r["Name"] for k in ("flow_","poly_expansion","pyramid_fused","gauss","update_matrices","box_solve","resize_linear","mag_minmax")))
print("GPU time total %.1f ms, flow kernels %.1f ms = %.1f %%" % (tot/1e6, flow/1e6, 100*flow/tot))
for r in rows[:14]:
    print(r["Name"][:72], r["Calls"], round(float(r["TotalDurationNs"])/1e6,2), "ms", r["Percentage"])
PY
