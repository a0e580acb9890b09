#!/bin/bash
# (run through gpurun: GRAFT_REPO_ROOT is the snapshot of the repo on the GPU box; default: this script's repo)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
export GRAFT_REPO_ROOT
# The non-headline BASELINE configurations on one GPU (config 2, config-4 shape, full ReLaX at 1080p / 2160p): bench lines
# and a kernel profile of the 2160p full pipeline.  Outputs under gpurun_out/.
R=$GRAFT_REPO_ROOT
python $R/bench.py --workload config2 --clips-per-step 8 --no-cpu-baseline --no-fast-mode --no-h2d > $R/gpurun_out/w_config2.json 2>/dev/null
python $R/bench.py --workload config4 --clips-per-step 8 --no-cpu-baseline --no-fast-mode --no-h2d > $R/gpurun_out/w_config4.json 2>/dev/null
python $R/bench.py --workload full1080p --clips-per-step 4 --steps 4 --warmup 1 --no-cpu-baseline --no-fast-mode > $R/gpurun_out/w_full1080p.json 2>/dev/null
python $R/bench.py --workload full2160p --clips-per-step 2 --steps 4 --warmup 1 --no-cpu-baseline --no-fast-mode > $R/gpurun_out/w_full2160p.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_full2160p -- python3 $R/bench.py --workload full2160p --clips-per-step 2 --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode > $R/gpurun_out/prof_full2160p.log 2>&1
for f in config2 config4 full1080p full2160p; do python3 -c "
import json,sys; d=json.load(open('$R/gpurun_out/w_$f.json')); print('$f', round(d['value'],2), 'clips/s', round(d['ms_per_step'],1), 'ms/step', d['config']['clips_per_step_per_gpu'], 'clips/step')"; done
