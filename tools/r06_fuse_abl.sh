#!/bin/bash
# round 6: variants of the f16x2 3x3 / back-to-back kernels (tools/abl_r06/*.so; "nophase2" gives WRONG results: timing only)
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
python -m pytest tests/test_gpu_h2.py tests/test_gpu_backbones.py tests/test_gpu_x6.py -m gpu -q -x -k "resnet50 or back_to_back or conv" 2>&1 | tail -3
run product ""
run stg2 stg2
run nophase2 nophase2
run product2 ""
