#!/bin/bash
# round 4: the whole GPU suite, the default bench line (as the driver runs it, timed) and the config-5 shape with its flow-stage roofline
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
TAG=${1:-a}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/suite_pytest_$TAG.txt
tail -4 $O/suite_pytest_$TAG.txt
SECONDS=0
python bench.py > $O/bench_default_$TAG.json 2> $O/bench_default_$TAG.err
echo "default bench wall: $SECONDS s"
python bench.py --workload full2160p --clips-per-step 8 --steps 3 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d --no-other-workloads > $O/bench_full2160p_$TAG.json 2> $O/bench_full2160p_$TAG.err
python - <<PY
import json
r=json.load(open("$O/bench_default_$TAG.json"))
print("config3", round(r["value"],2), "frac", round(r["roofline"]["frac"],3), "h2d", r.get("with_pinned_host_to_device_copy",{}).get("value"))
for k,v in r.get("other_workloads",{}).items():
    print(" ", k, round(v["value"],2), {kk: (round(vv,3) if isinstance(vv,float) else vv) for kk,vv in v.get("host_fed",{}).items() if kk in ("value","frac_of_device_resident","h2d_hidden_frac")}, v.get("roofline_flow_stage",{}).get("frac"))
r=json.load(open("$O/bench_full2160p_$TAG.json"))
fs=r["roofline_flow_stage"]
print("full2160p", round(r["value"],2), "flow stage frac", round(fs["frac"],3), "share", round(fs["time_share_of_step"],3), "traffic/alg", fs["traffic_over_algorithmic"], "dominant", round(fs["dominant_kernel"]["frac"],3))
PY
