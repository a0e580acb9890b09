#!/bin/bash
# (run through gpurun: GRAFT_REPO_ROOT is the snapshot of the repo on the GPU box; default: this script's repo)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
export GRAFT_REPO_ROOT
# per-dispatch durations of ONE ResNet-50 pass (rocprofv3 --kernel-trace), in launch order:
#   tools/resnet_layers.sh <tag> [N]      -> gpurun_out/resnet_layers_<tag>.txt
R=$GRAFT_REPO_ROOT
TAG=$1; N=${2:-1024}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rl_$TAG -- python3 $R/tools/resnet_step.py $N 2 both > $R/gpurun_out/rl_$TAG.log 2>&1
f=$(ls $R/gpurun_out/rl_$TAG/*/*kernel_trace.csv | head -1)
python3 $R/tools/trace_order.py $f conv1_x6 > $R/gpurun_out/resnet_layers_$TAG.txt
tail -75 $R/gpurun_out/resnet_layers_$TAG.txt
