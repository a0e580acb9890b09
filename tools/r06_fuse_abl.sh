#!/bin/bash
# round 6: variants of the back-to-back kernel's tail (tools/abl_r06/librelax_<tag>.so, built by hand from gemm_x6.hip with -DX6_B2B_PREFETCH_B=<n>)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
cd $R
run() {  # run <tag> <lib or "">
  if [ -n "$2" ]; then export RELAX_HIP_LIB=$R/tools/abl_r06/librelax_$2.so; else unset RELAX_HIP_LIB; fi
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abl_$1 -- python3 $R/tools/resnet_step.py 1024 2 both > $R/gpurun_out/abl_$1.log 2>&1
  f=$(ls -t $R/gpurun_out/abl_$1/*/*kernel_trace.csv | head -1)
  echo "== $1: $(python3 $R/tools/trace_order.py $f conv1_x6 | grep 'gemm_x6<256, [0-9]*, 4, [12], true' | awk '{printf "%s ", $2}')"
  cd $R
}
for t in "$@"; do run $t $t; done
for t in "$@"; do run ${t}_again $t; done
