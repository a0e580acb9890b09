#!/bin/bash
# (run through gpurun: GRAFT_REPO_ROOT is the snapshot of the repo on the GPU box; default: this script's repo)
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
export GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for g in 2 4 8 16 32; do
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/grp_f_$g -- python3 $R/tools/gemm_bench.py --clips 8 --only vit_ --iters 2 --group-m $g > $R/gpurun_out/grp_f_$g.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/grp_w_$g -- python3 $R/tools/gemm_bench.py --clips 8 --only vit_ --iters 2 --group-m $g > $R/gpurun_out/grp_w_$g.log 2>&1
  grep TFLOP $R/gpurun_out/grp_f_$g.log | head -5
done
