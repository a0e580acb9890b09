#!/bin/bash
# (run through gpurun: GRAFT_REPO_ROOT is the snapshot of the repo on the GPU box; default: this script's repo)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
export GRAFT_REPO_ROOT
# rocprofv3 kernel stats of one bench configuration: tools/prof_x6.sh <tag> <bench args...>
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fast-mode --no-h2d "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
f=$(ls $R/gpurun_out/prof_$TAG/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/prof_${TAG}_kernel_stats.csv
head -25 $f | cut -c1-200
