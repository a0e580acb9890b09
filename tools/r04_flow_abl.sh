#!/bin/bash
# timing ablations of the fused Farneback iteration (WRONG results by construction): which part of its work bounds it
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
{
python tools/flow_time.py 2160 3840 32 product
for f in tools/abl/librelax_flowabl*.so; do RELAX_HIP_LIB=$R/$f python tools/flow_time.py 2160 3840 32; done
} 2>/dev/null | tee $O/flow_ablations_${1:-a}.txt
