#!/bin/bash
# round 6, review item 4: what a conv2 -> conv3 fusion of ResNet-50's layer3 / layer4 could return at most (tools/abl_r06/librelax_h3b2b.so: the
# conv3 launches never fetch their A operand, the 3x3 launches never store their planes - WRONG results, timing only)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
run() {
  if [ -n "$2" ]; then export RELAX_HIP_LIB=$R/tools/abl_r06/librelax_$2.so; else unset RELAX_HIP_LIB; fi
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abl_$1 -- python3 $R/tools/resnet_step.py 1024 2 both > $R/gpurun_out/abl_$1.log 2>&1
  f=$(ls -t $R/gpurun_out/abl_$1/*/*kernel_trace.csv | head -1)
  python3 $R/tools/trace_order.py $f conv1_x6 > $R/gpurun_out/h3b2b_$1.txt
  echo "== $1: gemm_h3 launches of the pass: $(grep gemm_h3 $R/gpurun_out/h3b2b_$1.txt | awk '{s+=$2} END {printf "%.2f ms in %d launches", s/1e3, NR}'); $(tail -1 $R/gpurun_out/h3b2b_$1.txt)"
  cd $R
}
run product ""
run bound h3b2b
run product_again ""
