#!/bin/bash
# round 6: do the convolution launches of layer3 / layer4 gain when the CUs do not reach their epilogues together?  Diagnostic builds
# tools/build_ablations.sh h3stagger:<cycles> (first-round workgroups start (blockIdx & 3) x cycles apart); one ResNet-50 pass each on the same box
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
run() {  # run <tag> <lib or "">
  if [ -n "$2" ]; then export RELAX_HIP_LIB=$R/tools/abl_r06/librelax_$2.so; else unset RELAX_HIP_LIB; fi
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/stg_$1 -- python3 $R/tools/resnet_step.py 1024 2 both > $R/gpurun_out/stg_$1.log 2>&1
  f=$(ls -t $R/gpurun_out/stg_$1/*/*kernel_trace.csv | head -1)
  python3 $R/tools/trace_order.py $f conv1_x6 > $R/gpurun_out/stg_$1.txt
  echo "== $1: 1x1 form $(grep 'gemm_h3<false, false, true>' $R/gpurun_out/stg_$1.txt | awk '{s+=$2} END{printf "%.0f us in %d", s, NR}'); 3x3 form $(grep 'gemm_h3<false, true, true>' $R/gpurun_out/stg_$1.txt | awk '{s+=$2} END{printf "%.0f us", s}'); $(tail -1 $R/gpurun_out/stg_$1.txt)"
  cd $R
}
run product ""
for t in "$@"; do run $t $t; done
run product_again ""
