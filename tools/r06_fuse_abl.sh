#!/bin/bash
# round 6: timing-only variants of the back-to-back kernel (tools/abl_r06/*.so: WRONG results except nopf) - what each part of its tail costs
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
R=$GRAFT_REPO_ROOT
cd $R
run() {  # run <tag> <lib or ""> 
  if [ -n "$2" ]; then export RELAX_HIP_LIB=$R/tools/abl_r06/librelax_$2.so; else unset RELAX_HIP_LIB; fi
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/abl_$1 -- python3 $R/tools/resnet_step.py 1024 2 both > $R/gpurun_out/abl_$1.log 2>&1
  f=$(ls $R/gpurun_out/abl_$1/*/*kernel_trace.csv | head -1)
  echo "== $1: $(python3 $R/tools/trace_order.py $f conv1_x6 | grep 'true, true>' | awk '{printf "%s ", $2}')  $(tail -1 $R/gpurun_out/abl_$1.log)"
  cd $R
}
run product ""
run nopf nopf
run nost nost
run nores nores
run nost_nores nost_nores
run nophase2 nophase2
